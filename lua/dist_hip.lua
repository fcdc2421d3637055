--[[
Multi-GPU nomination from LuaJIT: one process per GPU (LOCAL_RANK picks the device in bot7hip_ffi.lua), candidates
sharded by contiguous row ranges, ONE exchange per nomination inside the library (b7_score_finish_global:
ncclAllReduce over xGMI, csrc/comm.hip).  Mirrors bot7_amd/dist.py + the communicator set-up of bench.py.

    local D = require('bot7hip.dist_hip')
    D.init()                                   -- RANK / WORLD_SIZE from the environment, id through B7_COMM_ID_FILE
    local lo, hi = D.shard_range(M_global)     -- this rank's rows [lo, hi) (0-based, half open); D.M_global = M_global
    grid = bot7.grids.sobol_hip{size = hi - lo, dims = d, skip = 1 + lo, mins = mins, maxes = maxes}()
    bot  = bot7.bots.bayesopt_hip(objective, hypers, config, {candidates = grid})   -- every rank, in lock step
    bot:run_experiment()

Lock step means: every rank runs the same driver on the same observations with Torch's RNG seeded identically
(torch.manualSeed(s) on all ranks before the bot is built), so the hyper samples and the random initial picks agree, and
every rank evaluates the objective on the nominee (bots/abstract.lua:124) -- world times per trial.  When that is not
acceptable (an expensive black box), use ONE process with a group of GPUs instead: bot7hip_ffi.use_group (INTEGRATION.md 4).

Index convention: bot:nominate returns the nominee's 1-based index in the UNION of the shards; D.commit(idx) is the sharded
form of bots/abstract.lua:118's steal and is the only consumer of that index (bots_bayesopt_hip.lua:run_trial).
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')
local D   = {rank = 0, world = 1, lo = 0, M_global = nil}

-- rank 0 makes the 128-byte id (ncclGetUniqueId) and publishes it through a file every rank can read (a shared
-- /dev/shm path on one node); the others poll for it.  Any other channel (MPI, a socket) does as well.
function D.init(rank, world, id_file)
  D.rank  = rank  or tonumber(os.getenv('RANK') or '0')
  D.world = world or tonumber(os.getenv('WORLD_SIZE') or '1')
  if D.world == 1 then return end
  local path = id_file or os.getenv('B7_COMM_ID_FILE') or '/dev/shm/bot7hip_comm_id'
  local id   = ffi.new('char[?]', hip.COMM_ID_BYTES)
  if D.rank == 0 then
    hip.check(hip.C.b7_comm_unique_id(id))
    local f = assert(io.open(path .. '.tmp', 'wb')); f:write(ffi.string(id, hip.COMM_ID_BYTES)); f:close()
    os.rename(path .. '.tmp', path)
  else
    local s
    repeat
      local f = io.open(path, 'rb')
      if f then s = f:read(hip.COMM_ID_BYTES); f:close() end
      if not s or #s < hip.COMM_ID_BYTES then s = nil; os.execute('sleep 0.05') end
    until s
    ffi.copy(id, s, hip.COMM_ID_BYTES)
  end
  hip.check(hip.C.b7_comm_init(hip.ctx, D.rank, D.world, id))       -- collective: ncclCommInitRank
  local z = ffi.new('double[1]', 0)
  hip.check(hip.C.b7_comm_allreduce_f64(hip.ctx, z, 1, hip.COMM_SUM))  -- barrier: everyone is up
  if D.rank == 0 then os.remove(path) end
end

-- contiguous, near-equal split of M rows: the first M % world ranks get one extra row (bot7_amd/dist.py)
function D.shard_range(M, rank, world)
  local rank, world = rank or D.rank, world or D.world
  local base, extra = math.floor(M / world), M % world
  local lo = rank * base + math.min(rank, extra)
  D.lo, D.M_global = lo, M
  return lo, lo + base + ((rank < extra) and 1 or 0)
end

-- bots/abstract.lua:118 `pending, candidates = steal(pending, candidates, idx)` with the candidates sharded over ranks:
-- idx_global is 1-based in the union.  Returns the nominee's coordinates (a d-vector, the same on every rank) and the
-- 1-based LOCAL index of the deleted row on the rank that held it (0 elsewhere).  The library deletes the row on the
-- owner's device and moves D.lo on the ranks behind it (b7_nominate_commit); D.M_global shrinks by one everywhere.
function D.commit(idx_global, d)
  local Mloc, dd = ffi.new('int64_t[1]'), ffi.new('int[1]')
  hip.check(hip.C.b7_grid_shape(hip.ctx, Mloc, dd))
  local loc, off = ffi.new('int64_t[1]'), ffi.new('int64_t[1]', D.lo)
  hip.check(hip.C.b7_shard_commit_rule(idx_global, D.lo, Mloc[0], loc, nil))
  local row = torch.DoubleTensor(d or dd[0])
  hip.check(hip.C.b7_nominate_commit(hip.ctx, idx_global, off, torch.data(row)))
  D.lo = tonumber(off[0])
  if D.M_global then D.M_global = D.M_global - 1 end
  return row, tonumber(loc[0])
end

-- score:div(divisor) on this rank's accumulator, then the global first maximum: (value, 1-based GLOBAL index)
function D.nominate(divisor, lo)
  local v, i = ffi.new('double[1]'), ffi.new('int64_t[1]')
  hip.check(hip.C.b7_score_finish_global(hip.ctx, divisor or 1.0, lo or D.lo, v, i))
  return v[0], tonumber(i[0])
end

function D.finalize() hip.check(hip.C.b7_comm_destroy(hip.ctx)) end

return D
