--[[
Multi-GPU nomination from LuaJIT: one process per GPU (LOCAL_RANK picks the device in bot7hip_ffi.lua), candidates
sharded by contiguous row ranges, ONE exchange per nomination inside the library (b7_score_finish_global:
ncclAllReduce over xGMI, csrc/comm.hip).  Mirrors bot7_amd/dist.py + the communicator set-up of bench.py.

    local D = require('bot7hip.dist_hip')
    D.init()                                   -- RANK / WORLD_SIZE from the environment, id through B7_COMM_ID_FILE
    local lo, hi = D.shard_range(M_global)     -- this rank's rows [lo, hi) (0-based, half open)
    grid = bot7.grids.sobol_hip{size = hi - lo, dims = d, skip = 1 + lo, mins = mins, maxes = maxes}()
    ... per hyper sample: model:predict_device(...), b7_score_ei / _cb (local accumulator) ...
    local value, idx = D.nominate(nSamples)    -- bots/bayesopt.lua:79 score:div + :96 score:max(1), over ALL ranks
--]]
local ffi = require('ffi')
local hip = require('bot7hip.bot7hip_ffi')
local D   = {rank = 0, world = 1, lo = 0}

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
  D.lo = lo
  return lo, lo + base + ((rank < extra) and 1 or 0)
end

-- score:div(divisor) on this rank's accumulator, then the global first maximum: (value, 1-based GLOBAL index)
function D.nominate(divisor, lo)
  local v, i = ffi.new('double[1]'), ffi.new('int64_t[1]')
  hip.check(hip.C.b7_score_finish_global(hip.ctx, divisor or 1.0, lo or D.lo, v, i))
  return v[0], tonumber(i[0])
end

function D.finalize() hip.check(hip.C.b7_comm_destroy(hip.ctx)) end

return D
