--[[
LuaJIT FFI binding of libbot7hip.so (include/bot7hip.h) for Torch7.

Delivered as source: this pipeline has no LuaJIT/Torch7 runtime, so the same C ABI is exercised by the
Python ctypes harness (bot7_amd/_lib.py, tests/).  The cdef block between the BEGIN/END markers is
include/bot7hip.h with comments and preprocessor lines stripped, written by tools/gen_lua_cdef.py;
tests/test_lua_shims.py fails when the two differ, and checks every hip.C.b7_* call in lua/*.lua against it
(name and argument count).
--]]
local ffi = require('ffi')

ffi.cdef[[
/* BEGIN generated from include/bot7hip.h (tools/gen_lua_cdef.py) */
typedef struct b7_ctx b7_ctx;
int b7_abi_version(void);
int b7_create(b7_ctx **out, int device_id);
void b7_destroy(b7_ctx *ctx);
const char *b7_last_error(const b7_ctx *ctx);
int b7_device_info(b7_ctx *ctx, char *name_out, int *compute_units, int64_t *hbm_bytes);
int b7_sync(b7_ctx *ctx);
int b7_set_workspace(b7_ctx *ctx, int64_t bytes);
int b7_grid_sobol(b7_ctx *ctx, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes, double *out_host);
int b7_sobol_direction_numbers(int dims, uint32_t *out);
int b7_grid_random(b7_ctx *ctx, int64_t size, int dims, uint64_t seed, int64_t row_offset, const double *mins, const double *maxes, double *out_host);
int b7_grid_random_torch(b7_ctx *ctx, int64_t size, int dims, uint64_t seed, int resolution, const double *mins, const double *maxes, double *out_host);
int b7_torch_rand(uint64_t seed, int64_t n, int resolution, double *out);
int b7_grid_colrange(b7_ctx *ctx, double *col_min, double *col_max);
int b7_grid_apply_onesided(b7_ctx *ctx, const double *mins, const double *maxes, const double *col_ext);
int b7_grid_upload(b7_ctx *ctx, const double *X_hid, int64_t M, int d);
int b7_grid_download(b7_ctx *ctx, int64_t row0 , int64_t rows, double *out_host);
int b7_grid_shape(b7_ctx *ctx, int64_t *M, int *d);
int b7_grid_remove(b7_ctx *ctx, int64_t idx1, double *row_out);
int b7_grid_remove_rows(b7_ctx *ctx, const int64_t *idx1, int64_t n, double *rows_out);
typedef struct { const double *lenscale_sq; double amp; double noise; double mean; } b7_hyp;
typedef struct { double jitter_eps; double jitter_growth; int var_with_noise; int var_clamp; double var_min; } b7_gp_opts;
int b7_gp_default_opts(b7_gp_opts *out);
int b7_gp_set_opts(b7_ctx *ctx, const b7_gp_opts *opts);
int b7_gp_fit(b7_ctx *ctx, const double *X_obs, const double *Y_obs, int N, int d, int ycols, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info);
int b7_gp_set_data(b7_ctx *ctx, const double *X_obs, const double *Y_obs, int N, int d, int ycols);
int b7_gp_fit_hyp(b7_ctx *ctx, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info);
int b7_gp_predict_hyp(b7_ctx *ctx, const b7_hyp *hyp, double *mean_host, double *var_host, double *nll_out, double *jitter_used, int *info);
int b7_gp_nll_batch(b7_ctx *ctx, int B, const double *lenscale_sq, const double *amp, const double *noise, const double *mean, double *nll_out, double *jitter_out, int *info_out);
int b7_chol(b7_ctx *ctx, const double *src_host, int n, double *res_host, double *jitter_used, int *info);
int b7_gp_predict(b7_ctx *ctx, double *mean_host, double *var_host);
int b7_gp_predict_at(b7_ctx *ctx, const double *X1, int64_t M1, double *mean_host, double *var_host);
int b7_gp_fantasize(b7_ctx *ctx, const double *X_pend, int P, int nFantasies, uint64_t seed, double *Y_out, double *mean_out, double *cov_out);
int b7_gp_append(b7_ctx *ctx, const double *x_new, const double *y_new);
int b7_gp_download(b7_ctx *ctx, double *L_host, double *alpha_host, double *Linv_host);
typedef struct { int n_layers; const int *dims; const double *const *W; const double *const *b; int activation; } b7_mlp;
int b7_blr_basis(b7_ctx *ctx, const b7_mlp *net, const double *X, int64_t M, double *Z_host);
int b7_blr_features(b7_ctx *ctx, const double *Z1, int64_t M, int z);
int b7_blr_fit(b7_ctx *ctx, const double *Z0, const double *Y0, int N, int z, double alpha_prec, double beta, double mean, double *nll_out);
int b7_blr_fit_x(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec, double beta, double mean, double *nll_out);
int b7_blr_predict(b7_ctx *ctx, double *mean_host, double *var_host);
int b7_score_reset(b7_ctx *ctx);
int b7_score_ei(b7_ctx *ctx, const double *fmin, double tradeoff);
int b7_score_cb(b7_ctx *ctx, double tradeoff, int upper, double sign);
int b7_score_finish(b7_ctx *ctx, double divisor, double *best_val, int64_t *best_idx1, double *scores_host);
int b7_comm_unique_id(void *id_out);
int b7_comm_init(b7_ctx *ctx, int rank, int world, const void *id);
int b7_comm_info(b7_ctx *ctx, int *rank, int *world);
int b7_comm_destroy(b7_ctx *ctx);
int b7_comm_allreduce_f64(b7_ctx *ctx, double *inout, int n, int op);
int b7_comm_pick_winner(const uint64_t *table, int world, double *best_val, int64_t *best_idx1);
int b7_score_finish_global(b7_ctx *ctx, double divisor, int64_t global_row_offset, double *best_val, int64_t *best_idx1);
int b7_nominate_commit(b7_ctx *ctx, int64_t idx1_global, int64_t *global_row_offset, double *row_out);
int b7_shard_commit_rule(int64_t idx1_global, int64_t offset, int64_t M_local, int64_t *local_idx1, int64_t *new_offset);
int b7_exchange_info(b7_ctx *ctx, int *world, int64_t *rows_per_rank, int64_t *winner_idx1, int *winner_rank, double *winner_row);
typedef struct { int kind; double tradeoff; int upper; double sign; const double *fmin; } b7_score_spec;
int b7_eval_nominate(b7_ctx *ctx, int S, const b7_hyp *hyps, const b7_score_spec *spec, int64_t global_row_offset, double *best_val, int64_t *best_idx1, double *jitter_out, int *info_out);
int b7_blr_eval_nominate(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec, double beta, double mean, const b7_score_spec *spec, int64_t global_row_offset, double *best_val, int64_t *best_idx1, double *jitter_used);
int b7_blr_eval_nominate_marg(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, int S, const double *alpha_prec, const double *beta, const double *mean, const b7_score_spec *spec, int64_t global_row_offset, double *best_val, int64_t *best_idx1, double *nll_out, double *jitter_used);
typedef struct b7_group b7_group;
int b7_group_create(b7_group **out, int n, const int *device_ids);
void b7_group_destroy(b7_group *g);
const char *b7_group_last_error(const b7_group *g);
int b7_group_info(b7_group *g, int *n, int *uses_rccl);
b7_ctx *b7_group_ctx(b7_group *g, int rank);
int b7_group_set_workspace(b7_group *g, int64_t bytes);
int b7_group_gp_set_opts(b7_group *g, const b7_gp_opts *opts);
int b7_group_grid_sobol(b7_group *g, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes);
int b7_group_grid_random(b7_group *g, int64_t size, int dims, uint64_t seed, const double *mins, const double *maxes);
int b7_group_grid_onesided(b7_group *g, const double *mins, const double *maxes);
int b7_group_grid_upload(b7_group *g, const double *X_hid, int64_t M, int d);
int b7_group_grid_shape(b7_group *g, int64_t *M_global, int *d, int64_t *offsets);
int b7_group_grid_download(b7_group *g, int64_t row0 , int64_t rows, double *out_host);
int b7_group_grid_remove_rows(b7_group *g, const int64_t *idx1, int64_t n, double *rows_out);
int b7_group_gp_set_data(b7_group *g, const double *X_obs, const double *Y_obs, int N, int d, int ycols);
int b7_group_eval_nominate(b7_group *g, int S, const b7_hyp *hyps, const b7_score_spec *spec, double *best_val, int64_t *best_idx1, double *jitter_out, int *info_out);
int b7_group_nominate_commit(b7_group *g, int64_t idx1_global, double *row_out);
int b7_ei_compute(b7_ctx *ctx, const double *mean, const double *var, const double *fmin, double tradeoff, int64_t M, int c, double *out);
int b7_cb_compute(b7_ctx *ctx, const double *mean, const double *var, double tradeoff, int upper, double sign, int64_t M, int c, double *out);
int b7_argmax(b7_ctx *ctx, const double *scores, int64_t M, double *best_val, int64_t *best_idx1);
int b7_timer_start(b7_ctx *ctx, int slot);
int b7_timer_stop(b7_ctx *ctx, int slot);
int b7_timer_ms(b7_ctx *ctx, int slot, float *ms_out);
int b7_profile_enable(b7_ctx *ctx, int on);
int b7_profile_reset(b7_ctx *ctx);
int b7_profile_get(b7_ctx *ctx, const char *phase, double *ms_total, int64_t *launches);
int b7_persist_fallbacks(b7_ctx *ctx);
/* END generated */
]]

local C = ffi.load(os.getenv('BOT7HIP_LIB') or 'bot7hip')
local M = {C = C}

-- the header's #define constants (B7_ prefix dropped)
-- BEGIN generated constants
M.ABI_VERSION = 1
M.OK = 0
M.ERR_INVALID = -1
M.ERR_HIP = -2
M.ERR_NOMEM = -3
M.ERR_STATE = -4
M.ERR_UNSUPPORTED = -5
M.ERR_RANGE = -6
M.ERR_COMM = -7
M.COMM_ID_BYTES = 128
M.COMM_SUM = 0
M.COMM_MAX = 1
M.COMM_MIN = 2
M.SCORE_EI = 1
M.SCORE_CB = 2
M.MAX_TIMERS = 16
-- END generated constants

-- one context per process (= per GPU); LOCAL_RANK picks the device when launched one process per GPU
local ctxp = ffi.new('b7_ctx*[1]')
local dev  = tonumber(os.getenv('LOCAL_RANK') or '0')
if C.b7_create(ctxp, dev) ~= 0 then
  error('bot7hip: ' .. ffi.string(C.b7_last_error(nil)))   -- hard error: there is no CPU fallback
end
M.ctx = ffi.gc(ctxp[0], C.b7_destroy)

-- status -> Lua error(), the reference's hard-error convention (utils/math.lua:168 pcall catches it)
function M.check(rc)
  if rc ~= 0 then error('bot7hip(' .. rc .. '): ' .. ffi.string(C.b7_last_error(M.ctx)), 2) end
  return rc
end

-- ---- one LuaJIT process, several GPUs ------------------------------------------------------------------------------
-- The reference is a single process (bots/abstract.lua:155-169).  M.use_group{0,1,...,7} turns this process's context into
-- a GROUP of contexts, one per listed device (b7_group_*): the candidate grid is sharded over them, the observations go to
-- all of them, bots_bayesopt_hip's nominate runs bayesopt:eval on every GPU at once with one exchange for the winner.  The
-- driver's own bookkeeping (self.candidates as a host tensor, steal, pending, observed, the ONE objective evaluation per
-- trial, Torch's RNG stream) stays exactly as the reference has it.  Call before any grid is generated.
-- M.ctx becomes member 0 (the sampler's likelihood evaluations run there).
M.group = nil
function M.use_group(device_ids)
  assert(M.group == nil, 'bot7hip: a group is already in use')
  local n   = #device_ids
  local ids = ffi.new('int[?]', n)
  for i = 1, n do ids[i-1] = device_ids[i] end
  local gp = ffi.new('b7_group*[1]')
  if C.b7_group_create(gp, n, ids) ~= 0 then error('bot7hip: ' .. ffi.string(C.b7_last_error(nil))) end
  M.group = ffi.gc(gp[0], C.b7_group_destroy)
  M.solo  = M.ctx                       -- keep the stand-alone context alive (its finaliser would run otherwise)
  M.ctx   = C.b7_group_ctx(M.group, 0)  -- owned by the group: no finaliser
  M.forget_resident()
  return n
end
function M.gcheck(rc)
  if rc ~= 0 then error('bot7hip(' .. rc .. '): ' .. ffi.string(C.b7_group_last_error(M.group)), 2) end
  return rc
end

-- DoubleTensor -> contiguous DoubleTensor whose storage the C call may read.  The CALLER keeps the returned tensor
-- in a local until the call has returned (t:contiguous() of a non-contiguous t is a temporary that LuaJIT may
-- otherwise collect between the pointer being taken and the call being made) and passes torch.data(c).
function M.pin(t)
  if t == nil then return nil end
  assert(t:type() == 'torch.DoubleTensor', 'bot7hip: DoubleTensor expected, got ' .. t:type())
  return t:contiguous()
end
function M.data(c) if c == nil then return nil end return torch.data(c) end

-- ---- which host tensor is "the candidate grid resident on the GPU" ------------------------------------------------
-- The driver keeps self.candidates as a plain DoubleTensor (bots/abstract.lua:35); a shim that has just produced or
-- uploaded a grid records (data pointer, rows, cols) here, and model:predict_device skips the upload when X_hid is
-- that very tensor.  Anything that edits the grid on either side goes through M.set_resident / M.forget_resident.
M.resident = nil
function M.set_resident(t) M.resident = {ptr = torch.data(t), rows = t:size(1), cols = t:size(2)} end
function M.forget_resident() M.resident = nil end
function M.is_resident(t)
  local r = M.resident
  return r ~= nil and torch.isTensor(t) and t:dim() == 2 and t:isContiguous() and r.rows == t:size(1)
         and r.cols == t:size(2) and r.ptr == torch.data(t)
end
-- resident on M.ctx itself as ONE grid (what b7_gp_predict / b7_score_* over "the grid" mean): never so with a group,
-- where M.ctx holds a shard
function M.is_resident_on_ctx(t) return M.group == nil and M.is_resident(t) end
function M.upload_grid(X_hid)
  local X = M.pin(X_hid)
  if M.group then M.gcheck(C.b7_group_grid_upload(M.group, M.data(X), X:size(1), X:size(2)))
  else M.check(C.b7_grid_upload(M.ctx, M.data(X), X:size(1), X:size(2))) end
  if X_hid:isContiguous() then M.set_resident(X_hid) else M.forget_resident() end
end

-- ---- keep the resident grid alive across trials --------------------------------------------------------------------
-- bots/abstract.lua:118 moves the nominated row out of self.candidates with utils.tensor.steal (utils/tensor.lua:
-- 175-193), which builds a NEW (M-1) x d tensor on the host.  Wrapped here: when `src` is the resident grid, the same
-- stable deletion runs on the device (b7_grid_remove_rows) and the new host tensor becomes the resident one, so the
-- next nomination uploads nothing.  Call once after require('bot7'):  require('bot7hip.bot7hip_ffi').install_steal_hook()
function M.install_steal_hook()
  local T = require('bot7.utils').tensor
  if M._orig_steal then return end
  M._orig_steal = T.steal
  T.steal = function(res, src, idx, axis_r, axis_s)
    local hook = M.is_resident(src) and (axis_s or 1) == 1
    local res2, src2 = M._orig_steal(res, src, idx, axis_r, axis_s)
    if hook then
      -- idx indexes `src`, the driver's host tensor: the WHOLE candidate set.  Without a group that is the grid on M.ctx;
      -- with one it is the union of the members' shards, and the group maps it to (member, local row) itself.
      local list = torch.isTensor(idx) and idx:long():view(-1) or torch.LongTensor{idx}
      local n    = list:nElement()
      local arr  = ffi.new('int64_t[?]', n)
      for i = 1, n do arr[i-1] = list[i] end
      if M.group and n == 1 then
        M.gcheck(C.b7_group_nominate_commit(M.group, arr[0], nil))   -- bots/abstract.lua:118: the nominee of this trial
      elseif M.group then
        M.gcheck(C.b7_group_grid_remove_rows(M.group, arr, n, nil))
      else
        M.check(C.b7_grid_remove_rows(M.ctx, arr, n, nil))
      end
      if src2 ~= nil then M.set_resident(src2) else M.forget_resident() end
    end
    return res2, src2
  end
end

return M
