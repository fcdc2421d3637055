"""HARNESS.  Multi-GPU layout of the scoring path: one process per GPU, candidates sharded by contiguous row ranges, ONE tiny
exchange per nomination -- b7_score_finish_global / b7_eval_nominate in libbot7hip.so (csrc/comm.hip: ncclAllReduce over
xGMI) -- and the stable deletion of the nominee on the union of the shards (b7_nominate_commit).  This module is the
host-side arithmetic around it (shard ranges, the winner rule restated for the gloo rehearsal) and the two candidate
stores the harness bot (harness/bots) runs the reference's trial loop on: ShardedScorer (this rank's shard, one process
per GPU) and GroupCandidates (one process, several GPUs: b7_group_*).

Candidates are independent given the fitted GP (bots/bayesopt.lua:56-99 scores them elementwise and takes one
max), so rank r of G owns rows [lo, hi) of the global grid, generates them itself (Sobol is closed-form per
index; the counter-based random grid likewise), refits the N x N model redundantly (2.9 GFLOP, deterministic,
cheaper than broadcasting 32 MiB of L) and scores only its shard.  The only data-path communication is the
arg-max exchange: every rank contributes (value, global 1-based index); RCCL has no MAXLOC, so the pairs are
summed into a zero-initialised [G, 2] slot buffer (an all-gather in all-reduce clothing, 16 B per rank) and every
rank picks the winner with TH's max semantics: the first NaN wins, otherwise the largest value, ties to the
lowest global index -- exactly what score:max(1) (bots/bayesopt.lua:96) returns on the unsharded vector."""
import numpy as np


def shard_range(M, rank, world):
    """Contiguous, near-equal split of M rows: the first M % world ranks get one extra row."""
    base, extra = divmod(int(M), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pick_winner(pairs):
    """pairs: iterable of (value, global_idx1); idx1 <= 0 marks an empty shard.  TH max semantics."""
    best = None
    for v, i in pairs:
        i = int(i)
        if i <= 0:
            continue
        if best is None:
            best = (v, i)
            continue
        bv, bi = best
        vn, bn = v != v, bv != bv
        if vn or bn:
            if vn and (not bn or i < bi):
                best = (v, i)
        elif v > bv or (v == bv and i < bi):
            best = (v, i)
    if best is None:
        raise ValueError("every shard is empty")
    return best


def exchange_best(local_val, local_idx1, lo, device=None, group=None):
    """All ranks learn the global (value, 1-based global index).  local_idx1 is 1-based within the shard that
    starts at global row `lo` (0-based); pass local_idx1 = 0 for an empty shard.  Indices ride in the f64 slot
    (exact below 2^53)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if local_idx1 <= 0:
            raise ValueError("every shard is empty")
        return local_val, int(lo + local_idx1)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    slots = torch.zeros(world, 2, dtype=torch.float64, device=device)
    # NaN scores must survive a SUM with zeros from the other ranks: they do (0 + NaN = NaN), and only this
    # rank writes its own slot.
    slots[rank, 0] = float(local_val) if local_idx1 > 0 else 0.0
    slots[rank, 1] = float(lo + local_idx1) if local_idx1 > 0 else 0.0
    dist.all_reduce(slots, op=dist.ReduceOp.SUM, group=group)
    s = slots.cpu().numpy()
    return pick_winner((float(s[r, 0]), int(s[r, 1])) for r in range(world))


class ShardedScorer(object):
    """Rank-local view of a sharded nomination: owns rows [lo, hi) of a global Sobol / counter-random grid on
    this rank's Context, and turns local (value, index) results into global ones."""

    def __init__(self, ctx, M_global, rank, world):
        self.ctx, self.M_global, self.rank, self.world = ctx, int(M_global), int(rank), int(world)
        self.lo, self.hi = shard_range(M_global, rank, world)

    # ---- what harness/bots needs of a candidate set that is not a host tensor ----------------------------------------
    @property
    def shape(self):
        """(rows of the UNION, dims): bots/bayesopt.lua:91 draws the initial picks against candidates:size(1)."""
        return (self.M_global, self.ctx.grid_shape()[1])

    def stage_data(self, X_obs, Y_obs):
        self.ctx.gp_set_data(X_obs, Y_obs)

    def commit(self, idx1_global, device=None, group=None):
        """bots/abstract.lua:118 steal(pending, candidates, idx) on the sharded set: the nominee's coordinates on every
        rank, its stable deletion on the owner, the offset shift behind it.  Product path: ONE C-ABI call,
        b7_nominate_commit (the row rides in the exchange record of the nomination; the random initial picks cost one
        more all-reduce).  gloo rehearsal: the library's own bookkeeping rule (b7_shard_commit_rule, host-only) + a
        torch.distributed sum that broadcasts the owner's row."""
        if self.ctx.comm_info()[1] == self.world:
            row, self.lo = self.ctx.nominate_commit(idx1_global, self.lo)
        else:
            import torch
            import torch.distributed as dist
            from bot7_amd import _lib
            M_local = self.hi - self.lo
            loc, new_lo = _lib.shard_commit_rule(idx1_global, self.lo, M_local)
            d = self.ctx.grid_shape()[1]
            row = torch.zeros(d + 1, dtype=torch.float64)
            if loc > 0:
                row[:d] = torch.from_numpy(np.asarray(self.ctx.grid_remove(loc), dtype=np.float64))
                row[d] = 1.0
            dist.all_reduce(row, op=dist.ReduceOp.SUM, group=group)
            if row[d].item() != 1.0:
                raise ValueError("index %d lies in %d shards" % (idx1_global, int(row[d].item())))
            row = row[:d].numpy().copy()
            self.lo = new_lo
        self.M_global -= 1
        self.hi = self.lo + self.ctx.grid_shape()[0]
        return row

    def make_sobol(self, dims, skip=1, mins=None, maxes=None, download=False):
        return self.ctx.grid_sobol(self.hi - self.lo, dims, skip + self.lo, mins, maxes, download=download)

    def make_random(self, dims, seed=0, mins=None, maxes=None, download=False):
        return self.ctx.grid_random(self.hi - self.lo, dims, seed, self.lo, mins, maxes, download=download)

    def eval_nominate(self, hyps, spec, device=None, group=None):
        """bayesopt:eval + nominate over the sharded grid (bots/bayesopt.lua:56-99): S fits + posteriors + score:adds
        and the global arg-max.  spec: the keyword arguments of Context.eval_nominate (score, fmin, tradeoff, ...).
        With a communicator on the context (or a world of one) this is ONE C-ABI call, b7_eval_nominate; the gloo
        rehearsal runs the separate entry points and exchanges through torch.distributed."""
        if self.ctx.comm_info()[1] == self.world:
            return self.ctx.eval_nominate(hyps, global_row_offset=self.lo, **spec)
        if self.hi > self.lo:
            for s, h in enumerate(hyps):
                self.ctx.gp_predict_hyp(h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
                if s == 0:
                    self.ctx.score_reset()
                if spec["score"] == "ei":
                    self.ctx.score_ei(spec["fmin"], spec.get("tradeoff") or 0.0)
                else:
                    self.ctx.score_cb(spec.get("tradeoff", 1.0), spec.get("upper", False), spec.get("sign", -1.0))
        return self.nominate(float(len(hyps)), device=device, group=group)

    def nominate(self, divisor=1.0, device=None, group=None):
        """score:div + global score:max(1).  Returns (value, global 1-based index).

        With a communicator on the context (Context.comm_init) the whole thing is ONE C-ABI call,
        b7_score_finish_global: the product path, RCCL inside the library.  Without one, the local result is
        exchanged through torch.distributed (the gloo rehearsal of tests/test_dist_gloo.py; same winner rule)."""
        if self.ctx.comm_info()[1] == self.world:   # a context without a communicator reports a world of one
            return self.ctx.score_finish_global(divisor, self.lo)
        if self.hi > self.lo:
            v, i, _ = self.ctx.score_finish(divisor, download=False)
        else:
            v, i = 0.0, 0
        return exchange_best(v, i, self.lo, device=device, group=group)


class GroupCandidates(object):
    """The candidate set of a single-process group (bot7_amd.Group, b7_group_*): what harness/bots calls on it."""

    def __init__(self, group):
        self.group = group

    @property
    def shape(self):
        M, d, _ = self.group.grid_shape()
        return (M, d)

    def stage_data(self, X_obs, Y_obs):
        self.group.gp_set_data(X_obs, Y_obs)

    def eval_nominate(self, hyps, spec, device=None, group=None):
        return self.group.eval_nominate(hyps, **spec)

    def commit(self, idx1_global, device=None, group=None):
        return self.group.nominate_commit(idx1_global)
