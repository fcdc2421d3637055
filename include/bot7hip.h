/*
 * bot7hip.h -- C ABI of libbot7hip.so: the MI355X (gfx950) implementation of bot7's GP-posterior +
 * acquisition-scoring hot path.
 *
 * The reference (montyhall/bot7, Lua/Torch7) has no FFI on this path: the "interface" a replacement
 * must honour is the Lua class protocol of bot7.grids / bot7.models / bot7.scores.  Each entry point
 * below names the reference method(s) whose arithmetic it replaces (paths relative to the reference
 * root).  The LuaJIT `ffi.cdef` of exactly these declarations and the three shim classes that call
 * them are in lua/ and INTEGRATION.md; tests and bench drive the same symbols through ctypes.
 *
 * Conventions
 *   - every function returns int: B7_OK (0) or a negative B7_ERR_*; nothing throws or aborts across the
 *     boundary; b7_last_error(ctx) gives the message of the last failure on that context;
 *   - all pointers are HOST pointers to contiguous row-major binary64 (int64_t for indices), owned by the
 *     caller and only touched during the call; "nullable" outputs may be NULL;
 *   - device memory is owned by the opaque context; one context = one GPU = one host thread at a time
 *     (the multi-GPU layout is one process per GPU, each with its own context);
 *   - calls are synchronous unless stated; row/candidate indices that cross the boundary are 1-BASED
 *     (Torch convention, bots/bayesopt.lua:96).
 */
#ifndef BOT7HIP_H
#define BOT7HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B7_ABI_VERSION 1

#define B7_OK 0
#define B7_ERR_INVALID (-1)     /* bad argument (shape, range, NULL) */
#define B7_ERR_HIP (-2)         /* HIP runtime failure; message holds hipGetErrorString */
#define B7_ERR_NOMEM (-3)       /* device allocation failed */
#define B7_ERR_STATE (-4)       /* call order: e.g. predict before fit, score before predict */
#define B7_ERR_UNSUPPORTED (-5) /* valid in the reference, not built yet (see DESIGN.md "out of scope") */
#define B7_ERR_RANGE (-6)       /* Sobol: dims >= 40 or index beyond 2^30-2 (grids/sobol.lua:36,317-324) */
#define B7_ERR_COMM (-7)        /* RCCL: library not loadable, or a communicator call failed */

typedef struct b7_ctx b7_ctx;

/* ---- context ---------------------------------------------------------------------------------- */
int b7_abi_version(void);
int b7_create(b7_ctx **out, int device_id);
void b7_destroy(b7_ctx *ctx);
const char *b7_last_error(const b7_ctx *ctx);
/* name_out: >= 64 bytes, nullable. */
int b7_device_info(b7_ctx *ctx, char *name_out, int *compute_units, int64_t *hbm_bytes);
int b7_sync(b7_ctx *ctx);

/* Bytes of the K(X*,X) chunk workspace (default 4 GiB); candidates are processed in chunks of
 * workspace / (8 * Npad) rows.  Must be set before the first predict. */
int b7_set_workspace(b7_ctx *ctx, int64_t bytes);

/* ---- grids: bot7.grids.sobol / bot7.grids.random ---------------------------------------------- */

/* grids/sobol.lua:58-90 generate + :216-335 i4_sobol.  Row j (1-based) is Sobol point number
 * j + skip - 1 (Gray-code order, 30 bits), then x*(maxes-mins)+mins as two rounded ops (:79-81);
 * mins/maxes both NULL = no affine map; mins only: x + (mins + column minimum) (:82-83, as the reference writes it);
 * maxes only: x * (maxes / column maximum) (:84-85) -- the column extremes are those of the WHOLE grid: a context with a
 * communicator combines them across ranks (one all-reduce of d doubles; collective).  The grid stays resident on the
 * device as THE candidate set; out_host (nullable, size x dims) receives a copy.  dims must be < 40 (:36).  A rank that
 * owns rows [lo, hi) of a global grid calls this with size = hi-lo and skip = global_skip + lo. */
int b7_grid_sobol(b7_ctx *ctx, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes,
                  double *out_host);

/* Host-only: the scaled direction numbers V[i][b] = m_(b+1) * 2^(29-b) (i < dims <= 39, b < 30) that
 * grids/sobol.lua:243-289 builds in self.bank; out has dims*30 entries.  Needs no context and no GPU. */
int b7_sobol_direction_numbers(int dims, uint32_t *out);

/* grids/random.lua:23-35 with torch.rand replaced by a counter-based generator (Torch's MT19937 stream
 * is not part of the reference tree): u(row, col) = top 53 bits of splitmix64(seed, row_offset+row, col)
 * scaled by 2^-53, then the same affine map.  Any dims. */
int b7_grid_random(b7_ctx *ctx, int64_t size, int dims, uint64_t seed, int64_t row_offset, const double *mins,
                   const double *maxes, double *out_host);

/* grids/random.lua:23-35 with torch.rand's OWN stream, for replaying a reference run point for point: Torch7's CPU generator
 * is MT19937 seeded by torch.manualSeed(seed) (TH/THRandom.c; not part of the reference tree, restated from its published
 * algorithm: parity unpinned).  resolution 32: a double is random() * 2^-32 (Torch7 of bot7's era); 53: the later
 * (random64() & (2^53 - 1)) * 2^-53.  The stream is sequential, so the uniforms are made on the host and uploaded; the
 * affine map (both / one / none of mins, maxes) runs on the device as for b7_grid_random.  No row_offset: a rank that owns
 * rows [lo, hi) of a grid made this way generates the whole stream and uploads its slice (b7_torch_rand + b7_grid_upload). */
int b7_grid_random_torch(b7_ctx *ctx, int64_t size, int dims, uint64_t seed, int resolution, const double *mins,
                         const double *maxes, double *out_host);
/* Host-only: n doubles of that stream (what torch.manualSeed(seed); torch.rand(n) returns). */
int b7_torch_rand(uint64_t seed, int64_t n, int resolution, double *out);

/* The pieces of the one-sided maps, for callers that combine shards themselves: grid:min(1) / grid:max(1) of the resident
 * grid (both nullable, d entries), and the map itself given the extremes of the whole grid (exactly one of mins / maxes). */
int b7_grid_colrange(b7_ctx *ctx, double *col_min, double *col_max);
int b7_grid_apply_onesided(b7_ctx *ctx, const double *mins, const double *maxes, const double *col_ext);

/* A caller-made grid (cache.candidates, bots/abstract.lua:30). */
int b7_grid_upload(b7_ctx *ctx, const double *X_hid, int64_t M, int d);
int b7_grid_download(b7_ctx *ctx, int64_t row0 /*0-based*/, int64_t rows, double *out_host);
int b7_grid_shape(b7_ctx *ctx, int64_t *M, int *d);

/* utils/tensor.lua:158-170 remove, as used by steal (:175-193) from bots/abstract.lua:118: stable
 * deletion of candidate row idx1 (1-based); later rows shift up by one.  row_out (nullable, d) gets the
 * removed row (what steal appends to `pending`). */
int b7_grid_remove(b7_ctx *ctx, int64_t idx1, double *row_out);
/* The same for an index list, as utils.tensor.remove / steal accept (utils/tensor.lua:158-193: `idx` is a
 * LongTensor): the rows named by idx1[0..n) (1-based, against the grid BEFORE the call; duplicates count once,
 * keep:indexFill is idempotent) are deleted in one stable pass.  rows_out (nullable, n x d) = src:index(1, idx),
 * in the order given. */
int b7_grid_remove_rows(b7_ctx *ctx, const int64_t *idx1, int64_t n, double *rows_out);

/* ---- model: gp_regressor + ardse + GaussianNoise_iso + constant mean (bots/bayesopt.lua:40-43) - */

/* Replaces the table model:parse_hypers returns (bots/bayesopt.lua:75).  lenscale_sq[k] is what
 * utils.math.pdist receives as `lenscale` (it divides squared differences, utils/math.lua:72). */
typedef struct {
  const double *lenscale_sq; /* d entries, > 0 */
  double amp;                /* signal variance sigma_f^2 */
  double noise;              /* sigma_n^2 added to diag K(X,X) */
  double mean;               /* constant mean m */
} b7_hyp;

/* Choices the reference leaves to the absent `gp` package, made explicit (defaults in brackets). */
typedef struct {
  double jitter_eps;    /* [1e-8]  utils/math.lua:175 */
  double jitter_growth; /* [1.1]   utils/math.lua:176 */
  int var_with_noise;   /* [0] add `noise` to the predictive variance */
  int var_clamp;        /* [0] clamp variance below at var_min (TH clamp: NaN passes) */
  double var_min;       /* [0.0] */
} b7_gp_opts;
int b7_gp_default_opts(b7_gp_opts *out);
int b7_gp_set_opts(b7_ctx *ctx, const b7_gp_opts *opts);

/* The arithmetic of model:predict's first half (call sites scores/expected_improvement.lua:63,
 * scores/confidence_bound.lua:63): K = amp*exp(-pdist(X,X,lenscale_sq)/2) + noise*I with the distance of
 * utils/math.lua:65-111; L = chol(K) with the jitter schedule of utils/math.lua:159-218 (eps <- eps*growth
 * added to the ORIGINAL diagonal until success, or chol(I) once eps > ||K||_F); alpha = K^-1 (Y - mean).
 * X_obs N x d (d <= 96), Y_obs N x ycols (ycols <= 256: fantasy columns share K, L and differ only in alpha).
 * Outputs (all nullable): nll_out[ycols] negative log marginal likelihood,
 * jitter_used (0 = none, -1 = fell back to chol(I)), info = 1-based first failing pivot (not > 0, NaN, or below
 * the smallest normal double) of the FIRST
 * attempt (0 = positive definite). */
int b7_gp_fit(b7_ctx *ctx, const double *X_obs, const double *Y_obs, int N, int d, int ycols, const b7_hyp *hyp,
              double *nll_out, double *jitter_used, int *info);

/* The same in two steps, for the regime Bayesian optimisation lives in: model:sample_hypers (bots/bayesopt.lua:68,
 * 73-75) drives samplers/slice.lua:92-168, whose every density evaluation refits the SAME (X_obs, Y_obs) under new
 * hypers.  b7_gp_set_data puts the data on the device once; b7_gp_fit_hyp then uploads d + 3 numbers per fit (the
 * residual Y - mean is formed on the device).  b7_gp_fit == b7_gp_set_data + b7_gp_fit_hyp, bit for bit. */
int b7_gp_set_data(b7_ctx *ctx, const double *X_obs, const double *Y_obs, int N, int d, int ycols);
int b7_gp_fit_hyp(b7_ctx *ctx, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info);

/* model:predict(X_obs, Y_obs, X_hid, hyp, {mean, var}) in ONE call for resident data and a resident grid
 * (scores/expected_improvement.lua:63 fits and predicts in one breath): b7_gp_fit_hyp + b7_gp_predict with the posterior
 * enqueued right behind the fit, so the host waits for the pivot report only and never idles the GPU between the two
 * (a failed plain attempt redoes the prediction after the jitter schedule).  Results as b7_gp_fit_hyp / b7_gp_predict;
 * every output is nullable. */
int b7_gp_predict_hyp(b7_ctx *ctx, const b7_hyp *hyp, double *mean_host, double *var_host, double *nll_out,
                      double *jitter_used, int *info);

/* B likelihood evaluations of the resident data at once: the slice sampler's step-out / step-in probes
 * (samplers/slice.lua:118-164) and multi-chain samplers ask for the density at several hyper vectors per update.
 * lenscale_sq is B x d row-major; amp / noise / mean have B entries; nll_out[B] as b7_gp_fit_hyp's (same jitter schedule
 * per fit: jitter_out[B] / info_out[B], nullable).  N <= 128 observations with d <= 32 -- the reference's own regime
 * (budget 100, bots/abstract.lua:64), where a trial is hundreds of sequential density evaluations against one nomination --
 * take ONE workgroup of ONE launch per evaluation (observations in, two numbers out through device-mapped pinned memory: no
 * copy calls; B = 1 carries its hypers in the kernel arguments and waits on a completion word instead of the stream:
 * 24 us per call up to N = 64, 43 us up to 128; a batch of up to 256 costs ~7 us more).  Larger problems: the B
 * factorisations run concurrently in ONE persistent launch (each a chain of workgroups; 28 fit on the chip at N <= 256, 7 at
 * N <= 512, larger ones go one after the other), with L z = r solved alongside: no inverse, no alpha.  The context's current fit (and its predictions) is left untouched -- except
 * when a fit's hand-offs time out twice (the GPU is shared with another persistent kernel): that likelihood is then
 * evaluated through the launch schedule in the context's fit slot, and the next predict asks for a new fit (B7_ERR_STATE).
 * One response column, N <= 4096. */
int b7_gp_nll_batch(b7_ctx *ctx, int B, const double *lenscale_sq, const double *amp, const double *noise,
                    const double *mean, double *nll_out, double *jitter_out, int *info_out);

/* utils.math.chol(src, 'L') (utils/math.lua:159-218) on a caller-provided symmetric n x n matrix: lower factor
 * with the same jitter schedule as b7_gp_fit.  res_host n x n (upper triangle zero).  Replaces the current fit
 * on this context.  jitter_used / info as in b7_gp_fit (nullable). */
int b7_chol(b7_ctx *ctx, const double *src_host, int n, double *res_host, double *jitter_used, int *info);

/* Second half of model:predict over the resident candidate grid: mean = m + K(X*,X) alpha (M x ycols),
 * var = amp - colsumsq(L^-1 K(X*,X)') (M).  Results stay on the device for the score calls; host copies
 * are optional. */
int b7_gp_predict(b7_ctx *ctx, double *mean_host, double *var_host);

/* model:predict for arbitrary X1 (M1 x d) that is not the resident grid; does not disturb the grid or the
 * score accumulator. */
int b7_gp_predict_at(b7_ctx *ctx, const double *X1, int64_t M1, double *mean_host, double *var_host);

/* model:fantasize(nFantasies, X_obs, Y_obs, X_pend, hyp) (scores/expected_improvement.lua:57; the class is in the
 * absent `gp` package): nFantasies joint draws from the posterior of the CURRENT fit at the P pending points,
 * y = mu_P + chol(Sigma_P) z with Sigma_P = K(Xp,Xp) - K(Xp,X) K^-1 K(X,Xp) (+ noise when var_with_noise) factored
 * with the utils.math.chol jitter schedule.  torch.randn is replaced by a counter-based generator:
 * z(k, s) = Box-Muller of splitmix64(seed, 2(k n + s) + 1 | + 2).  Y_out P x nFantasies; mean_out[P] and
 * cov_out[P x P] (nullable) expose mu_P and Sigma_P.  P <= 64; the fit must have ycols == 1. */
int b7_gp_fantasize(b7_ctx *ctx, const double *X_pend, int P, int nFantasies, uint64_t seed, double *Y_out,
                    double *mean_out, double *cov_out);

/* Incremental refit (bots/abstract.lua:137-144 appends one observation per trial): extends the CURRENT fit by one
 * observation (x_new[d], y_new[ycols]) under the same hypers in O(N^2): l = L^-1 k, lambda^2 = kappa - l'l, new rows
 * of L and L^-1, alpha recomputed.  Returns B7_ERR_STATE when the padded factor is full (N = 64, or a multiple of 128 above) or
 * lambda^2 does not stand clear of the rounding-error bound of its own evaluation,
 * lambda^2 <= (N+2) u (2 |l|'(|L^-1||k|) + l'l), u = 2^-53 (this includes lambda^2 <= 0) -- refit with b7_gp_fit then
 * (which also applies the jitter schedule). */
int b7_gp_append(b7_ctx *ctx, const double *x_new, const double *y_new);

/* Inspection (tests): lower Cholesky factor N x N, alpha N x ycols, explicit inverse factor N x N. */
int b7_gp_download(b7_ctx *ctx, double *L_host, double *alpha_host, double *Linv_host);

/* ---- model: bot7.models.dngo -- basis network + Bayesian linear head (models/dngo.lua:108-175) -------------- */

/* The trained feature extractor as plain arrays: layer l is nn.Linear(dims[l], dims[l+1]) with W[l] row-major
 * dims[l+1] x dims[l] (nn.Linear's weight layout) and bias b[l], each followed by the same activation
 * (0 identity, 1 Tanh, 2 ReLU, 3 Sigmoid); the output of the last listed layer is the basis (self.basis.output,
 * models/dngo.lua:101-105).  1..4 layers, widths <= 256.  Training the network (nnTools.trainer) is out of scope. */
typedef struct {
  int n_layers;
  const int *dims;          /* n_layers + 1 entries */
  const double *const *W;
  const double *const *b;
  int activation;
} b7_mlp;

/* models/dngo.lua:155-171: features by a forward pass.  X == NULL: the resident candidate grid, features stay on
 * the device for b7_blr_predict (Z_host nullable, M ignored).  X != NULL: M x dims[0] host rows, Z_host M x z. */
int b7_blr_basis(b7_ctx *ctx, const b7_mlp *net, const double *X, int64_t M, double *Z_host);
/* Caller-made features for the candidates (M x z) instead of a grid + network. */
int b7_blr_features(b7_ctx *ctx, const double *Z1, int64_t M, int z);
/* gp.models.bayes_linear (absent `gp` package) restated as standard Bayesian linear regression (Bishop 3.3;
 * Snoek et al. 2015): prior precision alpha_prec, noise precision beta, constant mean:
 *   K = beta Z0'Z0 + alpha_prec I,  m = beta K^-1 Z0'(Y0 - mean).  Replaces the GP fit on this context.
 * nll_out (nullable): negative log evidence, for sampling (alpha_prec, beta) on the host ('marginalize'). */
int b7_blr_fit(b7_ctx *ctx, const double *Z0, const double *Y0, int N, int z, double alpha_prec, double beta,
               double mean, double *nll_out);
/* The same with Z0 = basis(X0) computed and kept on the device (models/dngo.lua:155-162 + :174 in one call). */
int b7_blr_fit_x(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec,
                 double beta, double mean, double *nll_out);
/* mean = m0 + Z1 m, var = 1/beta + z' K^-1 z over the resident features; results stay on the device for the
 * b7_score_* calls (same accumulator and arg-max as the GP path). */
int b7_blr_predict(b7_ctx *ctx, double *mean_host, double *var_host);

/* ---- scores: bot7.scores.expected_improvement / confidence_bound + bayesopt marginalisation ---- */

/* bots/bayesopt.lua:69: score = zeros(M). */
int b7_score_reset(b7_ctx *ctx);
/* scores/expected_improvement.lua:69-88 on the last predict, added into the accumulator
 * (bots/bayesopt.lua:76 score:add).  fmin[ycols] = Y_obs:min(1) (:64); tradeoff = xi (:30). */
int b7_score_ei(b7_ctx *ctx, const double *fmin, double tradeoff);
/* scores/confidence_bound.lua:70-106; upper != 0 selects UCB (:96-100) else LCB (:102-106); the result is
 * val if sign > 0 else -val (:89-93); added into the accumulator. */
int b7_score_cb(b7_ctx *ctx, double tradeoff, int upper, double sign);
/* bots/bayesopt.lua:79 score:div(nSamples) then :96 score:max(1): best_val and the 1-based index of the
 * first maximum (first NaN wins, as TH's max).  divisor = 1 skips nothing: x/1 is exact.
 * scores_host nullable (M). */
int b7_score_finish(b7_ctx *ctx, double divisor, double *best_val, int64_t *best_idx1, double *scores_host);

/* ---- multi-GPU: the one exchange of a candidate-sharded nomination (RCCL over xGMI) ------------ *
 * Layout: one process per GPU, each with its own context; rank r owns candidate rows
 * [offset_r, offset_r + M_r) of the global grid (it generates them itself: b7_grid_sobol with skip + offset_r),
 * refits the GP redundantly, accumulates scores locally.  bots/bayesopt.lua:95-96's score:max(1) over ALL
 * candidates is then ONE ncclAllReduce inside b7_score_finish_global / b7_eval_nominate, over a table with one
 * fixed-width record of 64-bit words per rank: (value bits, global 1-based index, failure flag, rows of the shard,
 * the winner's grid row) -- so the same exchange also delivers what bots/abstract.lua:118-121 needs next, the
 * nominee's coordinates, to every rank (b7_nominate_commit), and a rank that could not score its shard makes every
 * rank return an error instead of leaving the others inside the collective.  800 B per rank, latency-bound.
 * librccl is dlopen'ed at the first b7_comm_* call; a context without a communicator is a world of one.
 * (One process driving several GPUs: the b7_group_* calls further down.) */
#define B7_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 makes the id (no context needed) and hands the 128 bytes to the other ranks by whatever
 * channel the host has (a file, an environment variable, a socket). */
int b7_comm_unique_id(void *id_out);
/* ncclCommInitRank on this context's device; collective over all `world` ranks; the exchange runs on the context's
 * stream.  One communicator per context. */
int b7_comm_init(b7_ctx *ctx, int rank, int world, const void *id);
int b7_comm_info(b7_ctx *ctx, int *rank, int *world);
int b7_comm_destroy(b7_ctx *ctx); /* also done by b7_destroy */
/* Control-plane reduction of a few host doubles (n <= 128) over the communicator, in place; also a barrier
 * (every rank returns only after all have entered and the context's stream has drained). */
#define B7_COMM_SUM 0
#define B7_COMM_MAX 1
#define B7_COMM_MIN 2
int b7_comm_allreduce_f64(b7_ctx *ctx, double *inout, int n, int op);
/* Host-only, needs no context and no GPU: the rule every rank applies to the gathered records, here as a [world, 2]
 * table of 64-bit words (value bits, global 1-based index; index 0 = that rank's shard was empty) -- TH's max over the union of the shards:
 * the first NaN wins, else the largest value, ties to the lowest global index.  B7_ERR_STATE when every shard is empty.
 * Exposed so that the rule can be checked for any world size without that many GPUs. */
int b7_comm_pick_winner(const uint64_t *table, int world, double *best_val, int64_t *best_idx1);
/* b7_score_finish across ranks: score:div(divisor) on this rank's shard, local first-maximum on the device, the
 * (value, global 1-based index = global_row_offset + local index) pairs of all ranks exchanged by one all-reduce,
 * and on every rank the same winner by TH's rule (first NaN, else max, ties to the lowest global index): exactly
 * what score:max(1) returns on the unsharded vector.  A rank whose shard is empty (no grid rows) still calls it. */
int b7_score_finish_global(b7_ctx *ctx, double divisor, int64_t global_row_offset, double *best_val,
                           int64_t *best_idx1);

/* bots/abstract.lua:118 `pending, candidates = steal(pending, candidates, idx)` on a candidate set sharded across
 * ranks: idx1_global is the nominee's 1-based index in the union of the shards (what b7_eval_nominate /
 * b7_score_finish_global returned, or the random initial pick of bots/bayesopt.lua:90-91 drawn against the GLOBAL row
 * count from a seed all ranks share).  *global_row_offset is this rank's offset (rows before its shard), in and out.
 * Every rank calls it with the same idx1_global:
 *   - the rank whose shard holds the row deletes it on the device, stably (utils/tensor.lua:158-170: later rows move up);
 *   - ranks behind the owner get *global_row_offset - 1 back: the union has shrunk in front of them;
 *   - row_out (nullable, d) receives the nominee's coordinates on EVERY rank: from the record of the last exchange when
 *     idx1_global is its winner (no communication at all), otherwise by one more all-reduce in which the owner
 *     contributes the row (the random initial trials).
 * The removal is enqueued, not waited for: the next nomination runs behind it on the context's stream.
 * B7_ERR_INVALID when the index lies in no shard, or the offsets overlap. */
int b7_nominate_commit(b7_ctx *ctx, int64_t idx1_global, int64_t *global_row_offset, double *row_out);
/* Host-only, needs no context and no GPU: the bookkeeping rule b7_nominate_commit applies, for a shard that holds global
 * rows (offset, offset + M_local] (1-based): local_idx1 = the row to delete in this shard (0: another shard's),
 * new_offset = offset - 1 when the deleted row lies before this shard, else offset. */
int b7_shard_commit_rule(int64_t idx1_global, int64_t offset, int64_t M_local, int64_t *local_idx1, int64_t *new_offset);
/* What the last exchange told this rank (all outputs nullable): world size, candidate rows held by every rank
 * (world entries: their sum is the global row count, their prefix sums the offsets of contiguous shards), the winner's
 * global index, owning rank and grid row (d entries).  B7_ERR_STATE when the grid has changed since. */
int b7_exchange_info(b7_ctx *ctx, int *world, int64_t *rows_per_rank, int64_t *winner_idx1, int *winner_rank,
                     double *winner_row);

/* ---- bayesopt:eval + nominate as ONE call (bots/bayesopt.lua:56-99) --------------------------- *
 * score = (1/S) sum_s acq(model, hyp_s, X_obs, Y_obs, X_hid) over the resident data (b7_gp_set_data) and the
 * resident grid, then score:max(1) -- across ranks when the context has a communicator, exactly as
 * b7_score_finish_global.  Same results as the loop  { b7_gp_predict_hyp(hyp_s); b7_score_ei|cb } x S  +
 * b7_score_finish_global(S, offset), bit for bit, but all S fits, posteriors and score:adds are enqueued back to
 * back and the host waits once (without a communicator: on a word the arg-max kernel raises behind its record in mapped
 * host memory, so the call may return a moment before the stream is formally idle -- later calls queue behind it as always,
 * b7_sync waits for the stream): each fit's pivot report is checked afterwards, and a failed pivot (or a
 * hand-off time-out) redoes the nomination through the per-sample path with utils/math.lua:159-218's jitter
 * schedule.  With S > 1 and one response column the S fits run SIDE BY SIDE in one persistent launch (one critical
 * workgroup each): a fit's dependent chain leaves most of the chip idle, so ten fits cost little more than one;
 * and when K(X*,X) of all S samples fits the workspace, K*, posterior and score:add of all samples are one launch each.
 * jitter_out / info_out: nullable, S entries (as b7_gp_fit's).  The accumulator holds score / S afterwards
 * (b7_score_finish with divisor 1 downloads it); the context's own fit slot and mean / variance vectors hold none of
 * the S samples (fit again before b7_gp_predict / b7_score_*). */
#define B7_SCORE_EI 1 /* scores/expected_improvement.lua: needs fmin[ycols]; tradeoff = xi */
#define B7_SCORE_CB 2 /* scores/confidence_bound.lua: tradeoff = kappa, upper, sign as b7_score_cb */
typedef struct {
  int kind;
  double tradeoff;
  int upper;
  double sign;
  const double *fmin;
} b7_score_spec;
int b7_eval_nominate(b7_ctx *ctx, int S, const b7_hyp *hyps, const b7_score_spec *spec, int64_t global_row_offset,
                     double *best_val, int64_t *best_idx1, double *jitter_out, int *info_out);

/* bayesopt:eval's DNGO branch + nominate as ONE call (bots/bayesopt.lua:65-66, :96 over models/dngo.lua:155-175):
 * b7_blr_fit_x(net, X0, Y0, ...) + b7_blr_basis(net, resident grid) + b7_blr_predict + the acquisition of `spec` (written,
 * not accumulated: ONE point (alpha_prec, beta, mean); b7_blr_eval_nominate_marg below marginalises over S of them) +
 * b7_score_finish_global, enqueued back to back with ONE host
 * synchronisation; the candidates' features are recomputed on every call, as the reference does.  Results as the separate
 * calls (the posterior mean comes out of the feature kernel itself: its last bits may differ from b7_blr_predict's).
 * best_idx1 / global_row_offset / communicator as b7_eval_nominate.  jitter_used (nullable): 0, or < 0 when the head's
 * Cholesky needed utils.math.chol's jitter schedule and the fit was redone through b7_blr_fit_x. */
int b7_blr_eval_nominate(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec,
                         double beta, double mean, const b7_score_spec *spec, int64_t global_row_offset, double *best_val,
                         int64_t *best_idx1, double *jitter_used);

/* The same with the head's hypers MARGINALISED -- models/dngo.lua:109 defaults hyp to 'marginalize' and hands it to the predictor
 * (:174).  gp.models.bayes_linear's own treatment lives in the absent `gp` package (parity unpinned); what is built is the
 * analogue of bayesopt:eval's GP loop (bots/bayesopt.lua:69-79): S samples (alpha_prec[s], beta[s], mean[s]) -- drawn by the host,
 * e.g. with bot7.samplers.slice over the evidence nll_out returns --, S heads fitted over the SAME features (one basis pass over
 * the observations, one over the candidates; for z <= 64 features the S heads are S workgroups of one launch), the acquisition
 * of every head added in sample order, score:div(S), score:max(1).  nll_out (nullable, S entries): the negative log evidence of
 * every head, as b7_blr_fit's.  jitter_used (nullable): 0, or < 0 when a head's Cholesky needed utils.math.chol's jitter schedule
 * and the nomination was redone head by head through b7_blr_fit_x.  S = 1 is b7_blr_eval_nominate. */
int b7_blr_eval_nominate_marg(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, int S,
                              const double *alpha_prec, const double *beta, const double *mean, const b7_score_spec *spec,
                              int64_t global_row_offset, double *best_val, int64_t *best_idx1, double *nll_out,
                              double *jitter_used);

/* ---- one process, several GPUs: the reference's single-process trial loop over a sharded grid ----------------- *
 * The reference is ONE LuaJIT process (bots/abstract.lua:155-169); a group lets that one process drive n GPUs, so the
 * trial loop, Torch's RNG stream and the single evaluation of the user's objective per trial (bots/abstract.lua:124)
 * stay as they are.  Member r is a b7_ctx on device_ids[r] holding candidate rows (offset_r, offset_r + M_r] of the grid
 * (contiguous, near-equal shards: the first M % n members hold one row more) and a full copy of the observations.
 * b7_group_eval_nominate enqueues fit + K* + posterior + score on every member's stream without waiting in between,
 * combines the members' exchange records -- ONE grouped ncclAllReduce over xGMI when the devices are distinct
 * (ncclCommInitAll), a merge on the host when members share a device (RCCL refuses that; it is how the sharding is
 * exercised on a one-GPU machine) or B7_GROUP_EXCHANGE=host is set -- and waits once per member.  Results are those of
 * the unsharded calls, bit for bit.  A member's grid must only be changed through the group. */
typedef struct b7_group b7_group;
int b7_group_create(b7_group **out, int n, const int *device_ids);   /* device ids may repeat (virtual ranks) */
void b7_group_destroy(b7_group *g);
const char *b7_group_last_error(const b7_group *g);
int b7_group_info(b7_group *g, int *n, int *uses_rccl);
/* Member r, e.g. for model:sample_hypers' density evaluations (b7_gp_fit_hyp / b7_gp_nll_batch on member 0) or
 * b7_profile_*; owned by the group. */
b7_ctx *b7_group_ctx(b7_group *g, int rank);
int b7_group_set_workspace(b7_group *g, int64_t bytes);
int b7_group_gp_set_opts(b7_group *g, const b7_gp_opts *opts);
/* b7_grid_sobol / _random / _upload for the whole grid: every member generates (receives) its shard; one-sided maps use
 * the column extremes of the union. */
int b7_group_grid_sobol(b7_group *g, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes);
int b7_group_grid_random(b7_group *g, int64_t size, int dims, uint64_t seed, const double *mins, const double *maxes);
int b7_group_grid_onesided(b7_group *g, const double *mins, const double *maxes);
int b7_group_grid_upload(b7_group *g, const double *X_hid, int64_t M, int d);
/* Rows of the union, its dims, and the members' offsets (n + 1 entries: offsets[n] = M_global); all nullable. */
int b7_group_grid_shape(b7_group *g, int64_t *M_global, int *d, int64_t *offsets);
int b7_group_grid_download(b7_group *g, int64_t row0 /*0-based, in the union*/, int64_t rows, double *out_host);
/* utils.tensor.remove / steal with an index tensor on the union (indices 1-based against the union before the call). */
int b7_group_grid_remove_rows(b7_group *g, const int64_t *idx1, int64_t n, double *rows_out);
/* b7_gp_set_data on every member. */
int b7_group_gp_set_data(b7_group *g, const double *X_obs, const double *Y_obs, int N, int d, int ycols);
/* b7_eval_nominate over the union: best_idx1 is 1-based in the union. */
int b7_group_eval_nominate(b7_group *g, int S, const b7_hyp *hyps, const b7_score_spec *spec, double *best_val,
                           int64_t *best_idx1, double *jitter_out, int *info_out);
/* b7_nominate_commit over the union: the nominee's coordinates (row_out, nullable, d) and its stable deletion on the
 * member that holds it.  When idx1_global is the winner of the last b7_group_eval_nominate the row comes from the
 * exchange record and nothing is copied or waited for. */
int b7_group_nominate_commit(b7_group *g, int64_t idx1_global, double *row_out);

/* EI.compute / conf_bound.compute / max on caller-provided host vectors (M x c mean, M var). */
int b7_ei_compute(b7_ctx *ctx, const double *mean, const double *var, const double *fmin, double tradeoff,
                  int64_t M, int c, double *out);
int b7_cb_compute(b7_ctx *ctx, const double *mean, const double *var, double tradeoff, int upper, double sign,
                  int64_t M, int c, double *out);
int b7_argmax(b7_ctx *ctx, const double *scores, int64_t M, double *best_val, int64_t *best_idx1);

/* ---- measurement ------------------------------------------------------------------------------ */

/* HIP-event timers on the context's stream (the stream every kernel of this library is launched on). */
#define B7_MAX_TIMERS 16
int b7_timer_start(b7_ctx *ctx, int slot);
int b7_timer_stop(b7_ctx *ctx, int slot);
int b7_timer_ms(b7_ctx *ctx, int slot, float *ms_out);

/* Per-kernel-phase event timing inside fit/predict/score (off by default: it serialises phases).
 * Phases: "prep" "kxx" "potrf" "trtri" "alpha" "ksx" "post" "score" "argmax" "exchange" "sobol" "remove".
 * b7_profile_get returns the summed milliseconds and launch count since the last reset. */
int b7_profile_enable(b7_ctx *ctx, int on);
int b7_profile_reset(b7_ctx *ctx);
int b7_profile_get(b7_ctx *ctx, const char *phase, double *ms_total, int64_t *launches);

/* How many persistent launches of this context (the one-launch Cholesky of N > 128 observations) ran into a hand-off time-out
 * -- the GPU was shared with somebody else's persistent kernel -- and were redone by the per-panel launch schedule (same bits).
 * A host that keeps its own "the context's fit is current" flag compares the count around b7_gp_nll_batch: when it moved, the
 * fallback used the context's fit slot (see there) and the next predict needs a new fit.  -1 for a null context. */
int b7_persist_fallbacks(b7_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* BOT7HIP_H */
