// Internal declarations shared by the translation units of libbot7hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "bot7hip.h"

// Tile constants.  Every matrix dimension that a GEMM-class kernel sees is padded to these.
constexpr int B7_NPAD = 128;  // observations padded to a multiple of this (post kernel's n-tile)
constexpr int B7_PANEL = 64;  // Cholesky panel width / small-GEMM tile
constexpr int B7_PERSIST_NMAX = 4096;  // largest padded N of the persistent Cholesky (64 panels: flag block, the vector job's LDS)
constexpr int B7_MROWS = 256; // chunk rows are multiples of this (largest candidates-per-block of any post variant)
constexpr int B7_MAX_D = 96;  // LDS budget of the covariance kernel: (64 + 2*32) rows x (dpad+1) doubles at dpad = 96

// The exchange table of a candidate-sharded nomination (comm.hip): one fixed-width record of 64-bit words per rank, so
// that ONE all-reduce carries the arg-max pair, the winner's grid row (what bots/abstract.lua:118-121 steals into
// `pending`), the shard's row count and a failure flag.  The width does not depend on the grid's dims: every rank issues
// the same collective whatever it holds (an empty shard has no grid to take dims from).
constexpr int B7_TAB_VAL = 0;     // bits of the local maximum
constexpr int B7_TAB_IDX = 1;     // its 1-based GLOBAL index; 0 = this shard is empty
constexpr int B7_TAB_STATUS = 2;  // 0 = fine; otherwise -(B7_ERR_*) of the rank that could not score its shard
constexpr int B7_TAB_ROWS = 3;    // candidate rows this shard holds
constexpr int B7_TAB_ROW0 = 4;    // d doubles: the grid row of the local maximum
constexpr int B7_TAB_W = B7_TAB_ROW0 + B7_MAX_D;
constexpr int B7_MAX_WORLD = 64;

// Padded input dimension: the covariance kernel is instantiated per class so its MFMA chain unrolls.
static inline int b7_dpad_class(int d) {
  return d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : d <= 48 ? 48 : d <= 64 ? 64 : 96;
}

// Device result block of a fit: int info[4] | double nll_terms[1 + 256] | the persistent schedule's hand-off flags (up to
// nb = 32 panels), so that ONE memset ahead of a persistent launch zeroes info and flags together.
constexpr size_t B7_INFO_HEAD_BYTES = 16 + sizeof(double) * 257 + 8;                       // 2080: a multiple of 16
constexpr size_t B7_PERSIST_FLAG_WORDS_MAX = 16 + 2 * 64 * 64 + 2 * 64;                    // FLAG_HDR + 2 nb^2 + 2 nb, nb <= 64
constexpr size_t B7_INFO_BYTES = B7_INFO_HEAD_BYTES + 4 * B7_PERSIST_FLAG_WORDS_MAX;

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct PhaseStat {
  double ms = 0.0;
  int64_t launches = 0;
};

struct b7_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_fit = nullptr;  // behind the fit report's copy: b7_gp_predict_hyp waits for it, not for the prediction
  std::string err;
  int cus = 0;

  // ---- candidate grid (row-major M x d), ping-pong for stable row removal
  DevBuf grid[2];
  int grid_cur = 0;
  int64_t M = 0;
  int d = 0;

  // ---- fit state
  bool fitted = false;
  bool have_data = false;  // b7_gp_set_data has put X_obs / Y_obs on the device
  int model_kind = 0;  // 0 = GP regressor, 1 = Bayesian linear head (DNGO): selects the predict path
  int N = 0, Npad = 0, dfit = 0, dpad = 0, ycols = 0;
  int yld = 1;   // leading dimension of alpha: 1 for one column, else ycols rounded up to 64 (zero padded)
  double amp = 0, noise = 0, mean = 0;
  b7_gp_opts opts;
  DevBuf xobs;   // N x d raw observations
  DevBuf w;      // dpad inverse squared lengthscales (0 in the padding)
  DevBuf zsc;    // Npad x dpad: observations scaled by w (0 in the padding)
  DevBuf zss;    // Npad: (sum z^2 w)/2  (+inf in the padding -> covariance 0)
  DevBuf K;      // Npad x Npad: K(X,X)+noise*I as assembled (kept for the jitter retries)
  DevBuf L;      // Npad x Npad: lower Cholesky factor (upper triangle zero)
  DevBuf Linv;   // Npad x Npad: explicit inverse of L (upper triangle zero)
  DevBuf W;      // Npad x Npad: scratch of the triangular inversion
  DevBuf dinv;   // (Npad/64) x 64 x 64: inverses of L's diagonal blocks
  DevBuf alpha;  // Npad x yld (0 in the padding)
  DevBuf resid;  // Npad x ycols: Y - mean, then L^-1 (Y - mean)
  DevBuf atmp;   // launch_alpha's intermediates: t = Linv resid (Npad x ycols) + slice partials (nslices x ycols x Npad)
  DevBuf info;   // int[4]: first failing pivot (1-based), 0 if none
  DevBuf ybuf;   // N x ycols raw

  // ---- predict / score state over the resident grid
  bool predicted = false;
  int64_t Mpred = 0;
  DevBuf mu;     // M x ycols
  DevBuf var;    // M
  DevBuf acc;    // M score accumulator
  bool acc_valid = false;
  int spin_us = 500;         // how long a call spins on a completion word in mapped host memory before it waits for the stream (B7_SPIN_US; 0: never)
  bool npad_small = true;    // N <= 64 (and <= 64 basis features) padded to ONE 64-block (B7_NPAD_SMALL=0: to 128)
  bool potrf_small = true;   // Npad == 64: factorisation + inverse (+ alpha) in one workgroup of one launch (blr_small.hip); B7_POTRF_SMALL=0 / any explicit B7_POTRF_SCHED: off
  bool blr_small = true;     // b7_blr_eval_nominate: the head for z <= 64 features in one workgroup of one launch (blr_small.hip)
  double fmin_scalar = 0.0;  // f_min of a single response column: a kernel argument of the EI kernels (launched with fmin_dev == nullptr), no staging copy
  bool acc_fresh = false;  // the accumulator stands for zeros that were never written: the next score launch onto it starts from 0.0
  // b7_eval_nominate's batched score (all S samples' mean / variance on the device), not launched yet: the exchange step runs it
  // fused with score:div, the arg-max and the record (score.hip: score_finish_slot_kernel); anybody else who needs the
  // accumulator first flushes it through the plain batch kernel (score_flush_pending)
  struct PendingScore {
    bool on = false;
    int kind = 0, S = 0, upper = 0;
    const double *mu = nullptr, *var = nullptr, *fd = nullptr;
    int64_t stride = 0;
    double tradeoff = 0.0, sign = 0.0;
  } pend;
  DevBuf ks;     // K(X*,X) chunk workspace
  size_t ks_bytes = (size_t)4 << 30;
  int diag_variant = 1;  // 64x64 diagonal-block kernel: 0 = rsqrt pivot chain, 1 = square-root-free chain with the DPP-fused
                         // multiply-add, 2 = the same with mov_dpp + fma (the bit-for-bit reference of 1) (B7_DIAG_VARIANT)
  int inverse_inline = 1;  // build inv(L) inside the factorisation launches: 0 never (separate trtri passes), 1 for
                           // Npad <= 8192, 2 always (B7_INVERSE_INLINE)
  // 8 KiB of pinned, device-mapped host memory for the small blocks: [0, 2304) fit report, [2304, 2320) arg-max
  // result, [4096, 6144) fmin staging, [6144, 8192) host copy of the exchange table, [8192, 8960) lengthscale
  // staging: kernels write
  // them directly or a copy lands without pageable staging; read after a stream synchronisation
  void *pinned = nullptr;
  std::vector<double> net_host;  // the basis network last uploaded to netbuf (packed W, b per layer)
  void *pin_blr = nullptr;  // b7_blr_eval_nominate: pinned staging of the observations and beta (y - mean)
  size_t pin_blr_bytes = 0;
  void *pin_eval_dev = nullptr;  // device address of pin_eval (mapped)
  void *pin_eval = nullptr;  // b7_eval_nominate: [S][4] pivot reports + [S][d] lengthscale staging (pinned)
  size_t pin_eval_bytes = 0;
  void *pinned_dev = nullptr;
  void *pin_nll = nullptr, *pin_nll_dev = nullptr;  // b7_gp_nll_batch's small path: hypers in, results out (pinned, device-mapped)
  size_t pin_nll_bytes = 0;
  bool fmin_staged = false;  // the fmin staging slot of the pinned block holds a caller's values
  bool potrf_attrs_set = false;  // dynamic-LDS limits of the Cholesky kernels raised (once per context)
  bool linv_done = false;  // launch_potrf produced Linv for the current factor
  int potrf_sched = 3;   // 1: one panel at a time (near update fused into the panel solve, far update riding on the
                         // next diagonal-block launch) for Npad <= 4096, 2: always; 0: panel groups with separate
                         // update launches; 3: ONE persistent launch for Npad <= 4096, else as 1 (B7_POTRF_SCHED)
  int syrk_small = 1;    // whole-K single-stage kernel for trailing updates with <= 256 tiles (B7_SYRK_SMALL)
  int potrf_defer = 1;   // far part of each trailing update rides on the next diagonal-block launch (B7_POTRF_DEFER)
  int potrf_group = 2;   // panels per bulk trailing update of the Cholesky (B7_POTRF_GROUP overrides); A/B at
                         // N = 2048 (tools/potrf_ab.py): G = 1 1.068 ms, 2 1.067, 4 1.119, 8 1.274
  // ---- persistent Cholesky schedule (potrf_persist.hip)
  struct JobList { DevBuf buf; int n = 0; };
  std::map<int, JobList> pjobs_cache;  // job queues by (nb, mode)
  DevBuf pstamps;  // diagnostics (B7_PERSIST_STAMPS)
  int pjobs_nb = 0, pjobs_n = 0;   // shape of the last single persistent launch (for the stamp reader)
  // b7_gp_nll_batch: B fits of the resident data in likelihood mode
  DevBuf bhyp, bw, bzsc, bzss, bK, bL, bdinv, bflags, binfo, bresid, bterms, bLinv, balpha, bmu, bvar;
  bool persist_attr_set = false, persist_stamps = false;
  int persist_helpers = 0;   // cap on the helper workgroups (B7_PERSIST_HELPERS; 0 = one per remaining CU)
  int persist_aborts = 0;    // launches that gave up waiting and were redone with the launch schedule
  int nll_small = 1;         // b7_gp_nll_batch at Npad <= 128, d <= 32: the one-workgroup-per-evaluation kernel (B7_NLL_SMALL: 0 general
                             // path, 1 gp_small_kernel (eight waves), 2 round 3's four-wave nll_small_kernel)
  bool kpost_small = true;   // GP posterior at Npad <= 128, d <= 32, one response column: K(X*,X), mean and variance in one kernel, K*
                             // never stored (kpost_small.hip; B7_KPOST_SMALL=0: ksx_kernel + post_kernel)
  bool fit_small = true;     // b7_eval_nominate / b7_gp_fit_hyp at Npad <= 128, d <= 32, one response column: the whole fit of a hyper
                             // vector in one workgroup of one launch (gp_small.hip; B7_FIT_SMALL=0: the general schedule)
  int persist_fault = -1;    // tests only (B7_PERSIST_FAULT): panel whose flag workgroup 0 withholds, to exercise the time-out
  int potrf_sched_saved = 0; // the schedule to return to after such a redo
  DevBuf part;   // argmax partials (value, index)
  DevBuf ticket; // score_finish_slot_kernel's arrival counter (zero between launches)
  DevBuf scratch; // misc (fmin upload, results)
  DevBuf tmpgrid; // predict_at temporary grid
  DevBuf tmpmu, tmpvar;
  DevBuf fant;   // fantasize workspace (pending-point covariance pieces)
  DevBuf feat;   // DNGO basis features of the resident grid: Mfeat x Npad (zero-padded columns)
  size_t feat_zeroed_bytes = 0;  // how much of `feat` was zeroed when it was laid out for feat_z columns
  int feat_z = -1;
  int64_t Mfeat = 0;
  int zdim = 0;
  uint64_t feat_version = 0, grid_version = 0;
  DevBuf netbuf; // MLP weights/biases of the basis network

  // ---- the arg-max exchange (comm.hip): RCCL communicator of this rank, one per context = per GPU = per process
  void *comm = nullptr;  // ncclComm_t
  int comm_rank = 0, comm_world = 1;
  DevBuf slots;  // [world, B7_TAB_W] x u64 exchange table (also the staging of b7_comm_allreduce_f64)
  uint64_t *tab_host = nullptr;  // pinned, device-mapped host copy of the table: B7_MAX_WORLD records, a staging record, the completion word
  uint64_t *tab_host_dev = nullptr;  // its device address
  struct b7_group *group = nullptr;  // set while the context is a member of a single-process group (group.hip)
  bool group_busy = false;           // ... and this while the group itself is calling the member's grid mutators
  // the last exchange as every rank saw it: the winner (index, owning rank, grid row) and the row count of every shard.
  // b7_nominate_commit takes the row from here, so that a model-based trial needs no second collective.
  bool win_valid = false;
  int64_t win_idx1 = 0;
  int win_rank = -1, win_d = 0;
  double win_row[B7_MAX_D];
  int64_t shard_rows[B7_MAX_WORLD];
  int shard_world = 0;

  // ---- measurement
  hipEvent_t tev[B7_MAX_TIMERS][2];
  bool tev_init = false;
  bool profile = false;
  std::map<std::string, PhaseStat> phases;
  // phase profiling without host synchronisation: event pairs are recorded into `pending` and turned into times
  // when somebody asks (b7_profile_get / _reset), after one stream synchronisation; events are recycled
  struct PendingPhase { const char *name; hipEvent_t e0, e1; };
  std::vector<PendingPhase> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t phase_e0 = nullptr;  // start event of the phase being recorded
};

// ---- error helpers ---------------------------------------------------------------------------------
int b7_fail(b7_ctx *c, int code, const char *fmt, ...);
// (the message names the call site, not the expression: the library's strings hold no internal constant names)
#define B7_HIP(c, expr)                                                                          \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return b7_fail((c), e__ == hipErrorOutOfMemory ? B7_ERR_NOMEM : B7_ERR_HIP, "%s:%d: %s", \
                     __FILE_NAME__, __LINE__, hipGetErrorString(e__));                           \
  } while (0)
#define B7_TRY(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != B7_OK) return rc__; \
  } while (0)

int b7_ensure(b7_ctx *c, DevBuf &b, size_t bytes);
void b7_release(DevBuf &b);

// Phase timing (no-ops unless profiling is on).
struct PhaseScope {
  b7_ctx *c;
  const char *name;
  PhaseScope(b7_ctx *c, const char *name);
  ~PhaseScope();
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// The padded size of a system of n observations (or basis features): one 64-block up to n = 64 -- the first 64 trials of every
// run at the reference's defaults, and the 50 features of models/dngo.lua's head: a quarter of the posterior's work and
// half the K* bytes of the 128 padding --, multiples of 128 (the posterior kernels' n-tile) above.  B7_NPAD_SMALL=0 (read in
// b7_create, like every switch): 128 always.
int npad_of(const b7_ctx *c, int64_t n);

// ---- kernel launchers (each enqueues on c->stream and returns a B7 code) ----------------------------
// sobol.hip
int launch_sobol(b7_ctx *c, double *out, int64_t size, int dims, int64_t skip, const double *mins,
                 const double *maxes);
int launch_random_grid(b7_ctx *c, double *out, int64_t size, int dims, uint64_t seed, int64_t row_offset,
                       const double *mins, const double *maxes);
int launch_remove_row(b7_ctx *c, const double *src, double *dst, int64_t M, int d, int64_t idx0);
int launch_remove_rows(b7_ctx *c, const double *src, double *dst, int64_t M, int d, const int64_t *cuts_dev, int ncut);
int launch_colrange(b7_ctx *c, const double *grid, int64_t M, int d, double *out_dev);
int launch_col_affine(b7_ctx *c, double *grid, int64_t M, int d, const double *v_dev, bool mul);
int launch_gather_rows(b7_ctx *c, const double *src, double *out, const int64_t *idx0_dev, int64_t n, int d);

// covar.hip
struct KBatchDesc {   // a batch of fits over the same observations: strides (doubles) per fit, per-fit amplitudes
  int64_t s_w = 0, s_zsc = 0, s_zsh = 0, s_out = 0;
  const double *amp = nullptr;
  // the posterior mean of every fit alongside (K(X*,X) of a batch of fits over the same candidates): per-fit alpha stride,
  // constant means, output stride
  int64_t s_alpha = 0, s_mu = 0;
  const double *mean = nullptr;
};
struct ObsSet {
  const double *zsc;  // npad x dpad scaled observations
  const double *zsh;  // npad half norms (+inf in the padding)
  int npad;
  const double *w = nullptr;  // inverse squared lengthscales (dpad); null = the context's current ones
  KBatchDesc batch;
};
int launch_prep_obs_aux(b7_ctx *c, const double *xobs, const double *ls_dev, int N, int npad, double *zsc,
                        double *zsh);
int launch_k_generic(b7_ctx *c, const double *xq, int64_t rows, int64_t Mtotal, const ObsSet &o, double *out);
int launch_prep_obs(b7_ctx *c, const double *xobs, const double *lenscale_sq_dev, int N, int d);
int launch_kxx(b7_ctx *c, double diag_add);
int launch_ksx_batch(b7_ctx *c, int S, const double *xq, int64_t rows, int64_t Mtotal, const double *w, const double *zsc,
                     const double *zss, const double *amp_dev, const double *mean_dev, const double *alpha, double *ks,
                     int64_t s_out, double *mu, int64_t s_mu);
int launch_ei_batch(b7_ctx *c, int S, const double *mu, const double *var, int64_t stride, const double *fmin_dev,
                    double tradeoff, int64_t M, double *acc);
int launch_cb_batch(b7_ctx *c, int S, const double *mu, const double *var, int64_t stride, double tradeoff, int upper,
                    double sign, int64_t M, double *acc);
int launch_kxx_batch(b7_ctx *c, int B, const double *ls_dev, const double *amp_dev, const double *noise_dev, double *w,
                     double *zsc, double *zss, double *K);
int launch_ksx(b7_ctx *c, const double *xq, int64_t row0, int64_t rows, int64_t Mtotal, int d, double *ks,
               double *mu, int ycols);

// potrf.hip
// What a factorisation hands to the launch_alpha that follows it (the one-block kernel does more than factor): passed by the
// caller from one to the other, not kept in the context -- a caller that changes the residual or the response columns in
// between simply does not pass it on.
struct FactorNote {
  bool alpha_done = false;        // alpha (one response column) is already in c->alpha
  int *report_written = nullptr;  // where the pivot report was mirrored (mapped host memory), or null
};
// K + extra*I -> L, dinv, info (+ Linv, using W).  report_hint: where a one-block factorisation should mirror its pivot report
// itself (mapped host memory), or null; note (nullable): see FactorNote
int launch_potrf(b7_ctx *c, double extra, bool with_inverse, int *report_hint = nullptr, FactorNote *note = nullptr);
int launch_trtri(b7_ctx *c);           // L, dinv -> Linv (no-op when launch_potrf already built it)
int launch_potrf_persist(b7_ctx *c, double extra, bool with_inverse);  // the same in one persistent launch (Npad <= 4096)
int launch_nll_batch(b7_ctx *c, int B, const double *K, double *L, double *dinv, unsigned *flags, int *info,
                     const double *resid, double *terms, const double *unused);
int launch_nll_one(b7_ctx *c, const double *K, double *L, double *dinv, unsigned *flags, int *info, const double *resid,
                   double *terms, double extra);
size_t persist_flag_words_host(int nb);
// nll_small.hip
bool nll_small_applies(const b7_ctx *c);
int launch_potrf_small(b7_ctx *c, int B, const double *K, double *L, double *Linv, double *dinv, const double *resid, double *alpha,
                       double extra, int *info, int *report_dev, int64_t sK, int64_t sL, int64_t sdinv, int64_t svec, int sinfo);
int launch_blr_head_small(b7_ctx *c, const double *Z, int N, int z, int ldz, const double *yv, double alpha_prec, double beta,
                          int *report_dev);
int launch_blr_heads_small(b7_ctx *c, int S, const double *Z, int N, int z, int ldz, const double *yraw_dev, const double *hyp3_dev,
                           double *L, double *Linv, double *m, double *b, int *info, int *report_dev, double *terms);
int launch_post_heads(b7_ctx *c, int S, const double *Linv, const double *feat, int64_t rows, int64_t Mtotal, double *var, int64_t svar,
                      const double *zero_dev, const double *invbeta_dev);
int launch_gemv_rows_batch(b7_ctx *c, int S, const double *A, int lda, const double *x, int64_t sx, int n, const double *base_dev,
                           int64_t rows, double *y, int64_t sy);
int launch_nll_small(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *terms_dev, int *info_dev,
                     unsigned *done_dev);
// kpost_small.hip
bool kpost_small_applies(const b7_ctx *c);
int launch_kpost_small(b7_ctx *c, int S, const double *xq, int64_t M, const double *w, const double *zsc, const double *zss,
                       const double *Linv, const double *alpha, const double *hyp_dev, double amp, double noise, double mean,
                       double *mu, double *var, int64_t sout);
// gp_small.hip
bool gp_small_applies(const b7_ctx *c);
int launch_nll_small8(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *terms_dev, int *info_dev,
                      unsigned *done_dev);
int launch_fit_small(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *hyp_out, double *w, double *zsc,
                     double *zss, double *L, double *Linv, double *dinv, double *alpha, double *resid, int *info_dev,
                     int *report_dev);
// report_dev (nullable): device address of mapped host memory that receives the first report_words ints of the pivot report
// resid, Linv -> alpha; report_dev (nullable): the pivot report's mirror; note: what the factorisation just before already did
int launch_alpha(b7_ctx *c, int *report_dev = nullptr, int report_words = 0, const FactorNote &note = FactorNote());
int launch_alpha_batch(b7_ctx *c, int B, const double *Linv, const double *resid, double *alpha, const int *report_src = nullptr,
                       int *report_dev = nullptr, int report_words = 0);  // B single-column fits
int launch_fit_batch(b7_ctx *c, int B, const double *K, double *L, double *Linv, double *dinv, unsigned *flags, int *info);
int launch_nll_terms(b7_ctx *c, double *out_dev);  // out[0] = sum log L_ii, out[1 + k] = r_k' alpha_k
int launch_fro_norm_sq(b7_ctx *c, const double *A, int n, int ld, double *out_dev);  // sum of squares of A[0:n, 0:n]
int launch_set_identity(b7_ctx *c);    // L = I (Npad x Npad), dinv = identity blocks: the chol(I) fallback

// posterior.hip
int launch_post(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *var);
int launch_post_batch(b7_ctx *c, int S, const double *Linv, const double *ks, int64_t sks, int64_t rows, int64_t Mtotal,
                      double *var, int64_t svar, const double *amp_dev, const double *noise_dev);

// extras.hip
int launch_mean_multi(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *mu);
int launch_gemm_nt(b7_ctx *c, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int m, int n,
                   int k);
int launch_fantasy_cov(b7_ctx *c, const double *kpp, const double *g, double *S, int P, double diag_add);
int launch_fantasy_factor(b7_ctx *c, double *S, double *dinv_tmp, int *info_dev);
int launch_fantasy_sample(b7_ctx *c, const double *Lp, const double *mu, int P, int n, uint64_t seed, double *out);
int launch_add_diag(b7_ctx *c, double *S, int ld, int n, double v);

int launch_append_finalize(b7_ctx *c, const double *krow, const double *lvec, const double *uvec, const double *evec,
                           int *status_dev);
int launch_append_vectors(b7_ctx *c, const double *krow, double *lvec, double *uvec, double *part, double *evec);
int launch_mlp_forward(b7_ctx *c, const double *X, int64_t M, int d, const double *net_dev, const int *dims,
                       int n_layers, int activation, double *out, int ld_out);
int launch_mlp_forward_mean(b7_ctx *c, const double *X, int64_t M, int d, const double *net_dev, const int *dims,
                            int n_layers, int activation, double *out, int ld_out, const double *mvec, double mean0,
                            double *mu, bool *mean_done);
int launch_gemv_rows(b7_ctx *c, const double *A, int lda, const double *x, int n, double base, int64_t row0,
                     int64_t rows, int64_t Mtotal, double *y);
int launch_transpose_pad(b7_ctx *c, const double *Z, int n, int ldz, int z, double *Zt, int zpad, int nk);
int launch_blr_assemble(b7_ctx *c, const double *G, double *K, int z, int zpad, double alpha_prec, double beta);

// score.hip
int launch_ei(b7_ctx *c, const double *mu, const double *var, const double *fmin_dev, double tradeoff,
              int64_t M, int ycols, double *out, bool accumulate);
int launch_cb(b7_ctx *c, const double *mu, const double *var, double tradeoff, int upper, double sign, int64_t M,
              int ycols, double *out, bool accumulate);
int launch_finish(b7_ctx *c, double *acc, int64_t M, double divisor, double *best_val, int64_t *best_idx1);
int launch_fill(b7_ctx *c, double *p, int64_t n, double v);
int launch_finish_slot(b7_ctx *c, double *acc, int64_t M, double divisor, uint64_t *tab_dev, int rank, int world,
                       int64_t offset, const double *grid, int d, bool all_slots, uint64_t *host_rec = nullptr,
                       unsigned *host_done = nullptr);

int launch_score_finish_slot(b7_ctx *c, const b7_ctx::PendingScore &ps, double *acc, int64_t M, double divisor, uint64_t *tab_dev,
                             int rank, int world, int64_t offset, const double *grid, int d, bool all_slots, uint64_t *host_rec = nullptr,
                             unsigned *host_done = nullptr);
int score_flush_pending(b7_ctx *c);
int launch_keep_record(b7_ctx *c, uint64_t *tab_dev, int rank, int world);
int launch_row_slot(b7_ctx *c, uint64_t *tab_dev, int rank, int world, int64_t idx1_global, int64_t local0, const double *grid,
                    int d);

// comm.hip: the pieces of a sharded nomination that b7_eval_nominate, b7_score_finish_global and the single-process
// group (group.hip) are assembled from
int exch_table_ensure(b7_ctx *c, int world);
int acc_materialize(b7_ctx *c);  // acc_fresh -> real zeros (before anything reads the accumulator)
int exch_local(b7_ctx *c, double divisor, int64_t offset, int rank, int world, bool all_slots, bool mirror = false);
int exch_wait_mirror(b7_ctx *c);  // after exch_local(..., mirror = true): spin on the completion word, then (or instead, when it takes long) the stream  // enqueue: score:div, local arg-max, this rank's record
int exch_fail_record(b7_ctx *c, int rank, int world, int code);  // enqueue: this rank's record says "could not score"
int exch_allreduce(b7_ctx *c);                                   // enqueue: the collective (no-op without a communicator)
int exch_rewrite_record(b7_ctx *c, int rank, int world);         // enqueue: zero every record but this rank's (before a repeated all-reduce)
int exch_fetch(b7_ctx *c, int first_rank, int nranks);           // enqueue: records [first, first + n) -> pinned host copy
bool exch_pick(const uint64_t *tab, int world, int stride, double *val, int64_t *idx1, int *rank);
int exch_conclude(b7_ctx *c, const uint64_t *tab, int world, double *best_val, int64_t *best_idx1);  // statuses, winner, cache
void exch_forget(b7_ctx *c);

// api.hip: bayesopt:eval as stream work without a host wait, its report check and the per-sample redo
int eval_validate(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec);
int eval_enqueue(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec);
bool eval_reports_clean(b7_ctx *c, int S);
int eval_redo(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec, double *jitter_out, int *info_out);
int grid_drop_row(b7_ctx *c, int64_t local_idx1, double *row_out_sync);  // stable deletion, enqueued; row_out != NULL synchronises
