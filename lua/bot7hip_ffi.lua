--[[
LuaJIT FFI binding of libbot7hip.so (include/bot7hip.h) for Torch7.

Delivered as source: this pipeline has no LuaJIT/Torch7 runtime, so the same C ABI is exercised by the
Python ctypes harness (bot7_amd/_lib.py, tests/).  The cdef below is the header's declarations verbatim
(comments stripped); keep the two in sync (tests/test_abi_and_host.py compares the symbol lists).
--]]
local ffi = require('ffi')

ffi.cdef[[
typedef struct b7_ctx b7_ctx;
int  b7_abi_version(void);
int  b7_create(b7_ctx **out, int device_id);
void b7_destroy(b7_ctx *ctx);
const char *b7_last_error(const b7_ctx *ctx);
int  b7_device_info(b7_ctx *ctx, char *name_out, int *compute_units, int64_t *hbm_bytes);
int  b7_sync(b7_ctx *ctx);
int  b7_set_workspace(b7_ctx *ctx, int64_t bytes);
int  b7_grid_sobol(b7_ctx *ctx, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes, double *out_host);
int  b7_sobol_direction_numbers(int dims, uint32_t *out);
int  b7_grid_random(b7_ctx *ctx, int64_t size, int dims, uint64_t seed, int64_t row_offset, const double *mins, const double *maxes, double *out_host);
int  b7_grid_upload(b7_ctx *ctx, const double *X_hid, int64_t M, int d);
int  b7_grid_download(b7_ctx *ctx, int64_t row0, int64_t rows, double *out_host);
int  b7_grid_shape(b7_ctx *ctx, int64_t *M, int *d);
int  b7_grid_remove(b7_ctx *ctx, int64_t idx1, double *row_out);
typedef struct { const double *lenscale_sq; double amp; double noise; double mean; } b7_hyp;
typedef struct { double jitter_eps; double jitter_growth; int var_with_noise; int var_clamp; double var_min; } b7_gp_opts;
int  b7_gp_default_opts(b7_gp_opts *out);
int  b7_gp_set_opts(b7_ctx *ctx, const b7_gp_opts *opts);
int  b7_gp_fit(b7_ctx *ctx, const double *X_obs, const double *Y_obs, int N, int d, int ycols, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info);
int  b7_chol(b7_ctx *ctx, const double *src_host, int n, double *res_host, double *jitter_used, int *info);
int  b7_gp_predict(b7_ctx *ctx, double *mean_host, double *var_host);
int  b7_gp_predict_at(b7_ctx *ctx, const double *X1, int64_t M1, double *mean_host, double *var_host);
int  b7_gp_fantasize(b7_ctx *ctx, const double *X_pend, int P, int nFantasies, uint64_t seed, double *Y_out, double *mean_out, double *cov_out);
int  b7_gp_append(b7_ctx *ctx, const double *x_new, const double *y_new);
int  b7_gp_download(b7_ctx *ctx, double *L_host, double *alpha_host, double *Linv_host);
typedef struct { int n_layers; const int *dims; const double *const *W; const double *const *b; int activation; } b7_mlp;
int  b7_blr_basis(b7_ctx *ctx, const b7_mlp *net, const double *X, int64_t M, double *Z_host);
int  b7_blr_features(b7_ctx *ctx, const double *Z1, int64_t M, int z);
int  b7_blr_fit(b7_ctx *ctx, const double *Z0, const double *Y0, int N, int z, double alpha_prec, double beta, double mean, double *nll_out);
int  b7_blr_fit_x(b7_ctx *ctx, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec, double beta, double mean, double *nll_out);
int  b7_blr_predict(b7_ctx *ctx, double *mean_host, double *var_host);
int  b7_score_reset(b7_ctx *ctx);
int  b7_score_ei(b7_ctx *ctx, const double *fmin, double tradeoff);
int  b7_score_cb(b7_ctx *ctx, double tradeoff, int upper, double sign);
int  b7_score_finish(b7_ctx *ctx, double divisor, double *best_val, int64_t *best_idx1, double *scores_host);
int  b7_ei_compute(b7_ctx *ctx, const double *mean, const double *var, const double *fmin, double tradeoff, int64_t M, int c, double *out);
int  b7_cb_compute(b7_ctx *ctx, const double *mean, const double *var, double tradeoff, int upper, double sign, int64_t M, int c, double *out);
int  b7_argmax(b7_ctx *ctx, const double *scores, int64_t M, double *best_val, int64_t *best_idx1);
int  b7_timer_start(b7_ctx *ctx, int slot);
int  b7_timer_stop(b7_ctx *ctx, int slot);
int  b7_timer_ms(b7_ctx *ctx, int slot, float *ms_out);
int  b7_profile_enable(b7_ctx *ctx, int on);
int  b7_profile_reset(b7_ctx *ctx);
int  b7_profile_get(b7_ctx *ctx, const char *phase, double *ms_total, int64_t *launches);
]]

local C = ffi.load(os.getenv('BOT7HIP_LIB') or 'bot7hip')
local M = {C = C}

-- one context per process (= per GPU); LOCAL_RANK picks the device when launched one process per GPU
local ctxp = ffi.new('b7_ctx*[1]')
local dev  = tonumber(os.getenv('LOCAL_RANK') or '0')
if C.b7_create(ctxp, dev) ~= 0 then
  error('bot7hip: ' .. ffi.string(C.b7_last_error(nil)))   -- hard error: there is no CPU fallback
end
M.ctx = ffi.gc(ctxp[0], C.b7_destroy)
M.grid_version = 0   -- bumped whenever the resident grid changes

-- status -> Lua error(), the reference's hard-error convention (utils/math.lua:168 pcall catches it)
function M.check(rc)
  if rc ~= 0 then error('bot7hip(' .. rc .. '): ' .. ffi.string(C.b7_last_error(M.ctx)), 2) end
end

-- contiguous DoubleTensor -> double*
function M.ptr(t)
  if t == nil then return nil end
  assert(t:type() == 'torch.DoubleTensor')
  return torch.data(t:contiguous())
end

return M
