/*
 * b7_oracle.c -- CPU restatement of bot7's acquisition hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (bot7_amd/, libbot7hip.so) never does.
 *
 * Every function restates, operation for operation, a piece of the reference (paths relative to
 * /root/reference, the Torch7/Lua package montyhall/bot7).  One rounded IEEE-754 binary64 operation in
 * the Lua source is one rounded operation here: build with  -O2 -ffp-contract=off -fno-fast-math
 * (see oracle/Makefile) so the compiler neither fuses nor reorders them.
 *
 * Pinning status
 *   PINNED (reference-held vectors): Sobol helpers against the truth tables in grids/sobol.lua:97-121
 *     and :146-170; the A&S erf constants utils/math.lua:263-265 against the documented 1.5e-7 bound.
 *   PINNED (traced from the reference recurrence, tests/golden/sobol_*.json): Sobol points.
 *   PARITY UNPINNED: everything that depends on the GP posterior (gp.models.gp_regressor lives in the
 *     un-vendored, unversioned luarocks package `gp`, models/init.lua:15) -- see oracle/gp.py.
 *     The reference holds no tests, fixtures or golden outputs for EI/CB/argmax either; those are
 *     restated from source text only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_DIMS 40 /* grids/sobol.lua:31 */
#define ORC_LOG_MAX 30  /* grids/sobol.lua:32 */

/* ------------------------------------------------------------------------------------------------
 * utils/bits.lua:27-82 -- XOR on doubles through a 32-entry 0/1 expansion.
 * dec2bin: lsb = modulus(dec,2); fill bit bp from the right; dec = floor((dec-lsb)/2)   (:50-57)
 * modulus(val,base) = val - base*floor(val*(1/base))                                   (utils/math.lua:44-48)
 * bitwise_xor: (bin(x)+bin(y)) == 1, then bin2dec = dot(bits, 2^(31..0))               (:64-67,:79-82)
 * ---------------------------------------------------------------------------------------------- */
static double orc_modulus(double val, double base) { return val + (-base) * floor(val * (1.0 / base)); }

static void orc_dec2bin(double dec, unsigned char bin[32]) {
  int bp = 32;
  memset(bin, 0, 32);
  double lsb = orc_modulus(dec, 2.0);
  while (bp > 0) {
    if (lsb != 0.0) bin[bp - 1] = 1;
    dec = floor((dec + (-lsb)) / 2.0);
    bp -= 1;
    lsb = orc_modulus(dec, 2.0);
  }
}

double orc_bitwise_xor(double x, double y) {
  unsigned char bx[32], by[32];
  orc_dec2bin(x, bx);
  orc_dec2bin(y, by);
  double acc = 0.0;
  for (int k = 0; k < 32; ++k) { /* torch.dot with weights 2^(31-k), ascending k */
    double bit = ((bx[k] + by[k]) == 1) ? 1.0 : 0.0;
    acc += bit * ldexp(1.0, 31 - k);
  }
  return acc;
}

/* grids/sobol.lua:92-139 */
int orc_i4_bit_hi1(double n) {
  double i = floor(n);
  int bit = 0;
  while (i > 0) {
    bit += 1;
    i = floor(i / 2);
  }
  return bit;
}

/* grids/sobol.lua:141-189 */
int orc_i4_bit_lo0(double n) {
  int bit = 1;
  double i = floor(n);
  double i2 = floor(i / 2);
  while (i != 2 * i2) {
    bit += 1;
    i = i2;
    i2 = floor(i / 2);
  }
  return bit;
}

/* Generator state: the fields of the Lua object that i4_sobol mutates (grids/sobol.lua:39-44,54). */
typedef struct {
  double bank[ORC_MAX_DIMS][ORC_LOG_MAX]; /* self.bank, 1-based in Lua, 0-based here */
  double lastq[ORC_MAX_DIMS];
  double recipd;
  double seed; /* self.seed, starts at -1 (:39) */
  int dims;    /* config.dims after initialisation; 0 = "nil" */
  int initialized;
  int maxcol;
} orc_sobol;

static const double orc_poly[ORC_MAX_DIMS] = {/* grids/sobol.lua:46-52 */
                                              1,   3,   7,   11,  13,  19,  25,  37,  59,  47,  61,  55,  41,  67,
                                              97,  91,  109, 103, 115, 131, 193, 137, 145, 143, 241, 157, 185, 167,
                                              229, 171, 213, 191, 253, 203, 211, 239, 247, 285, 369, 299};

/* grids/sobol.lua:338-391 create_bank: initial direction numbers, column c (1-based) from row r0. */
static void orc_create_bank(orc_sobol *g) {
  static const double c2[] = {1, 3, 1, 3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3, 1,
                              3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3, 1, 3, 1, 3}; /* rows 3..40 */
  static const double c3[] = {7, 5, 1, 3, 3, 7, 5, 5, 7, 7, 1, 3, 3, 7, 5, 1, 1, 5, 3,
                              3, 1, 7, 5, 1, 3, 3, 7, 5, 1, 1, 5, 7, 7, 5, 1, 3, 3}; /* rows 4..40 */
  static const double c4[] = {1, 7, 9,  13, 11, 1, 3,  7, 9, 5,  13, 13, 11, 3,  15, 5, 3, 15,
                              7, 9, 13, 9,  1,  11, 7, 5, 15, 1, 15, 11, 5,  3,  1,  7, 9}; /* rows 6..40 */
  static const double c5[] = {9,  3, 27, 15, 29, 21, 23, 19, 11, 25, 7,  13, 17, 1,  25, 29, 3,
                              31, 11, 5, 23, 27, 19, 21, 5,  1,  17, 13, 7,  15, 9,  31, 9}; /* rows 8..40 */
  static const double c6[] = {37, 33, 7,  5,  11, 39, 63, 27, 17, 15, 23, 29, 3, 21,
                              13, 31, 25, 9,  49, 33, 19, 29, 11, 19, 27, 15, 25}; /* rows 14..40 */
  static const double c7[] = {13, 33, 115, 41, 79, 17, 29,  119, 75, 73, 105,
                              7,  59, 65,  21, 3,  113, 61, 89,  45, 107}; /* rows 20..40 */
  static const double c8[] = {7, 23, 39};                                   /* rows 38..40 */
  memset(g->bank, 0, sizeof(g->bank));
  for (int r = 0; r < ORC_MAX_DIMS; ++r) g->bank[r][0] = 1.0; /* :342 */
  for (int k = 0; k < 38; ++k) g->bank[2 + k][1] = c2[k];
  for (int k = 0; k < 37; ++k) g->bank[3 + k][2] = c3[k];
  for (int k = 0; k < 35; ++k) g->bank[5 + k][3] = c4[k];
  for (int k = 0; k < 33; ++k) g->bank[7 + k][4] = c5[k];
  for (int k = 0; k < 27; ++k) g->bank[13 + k][5] = c6[k];
  for (int k = 0; k < 21; ++k) g->bank[19 + k][6] = c7[k];
  for (int k = 0; k < 3; ++k) g->bank[37 + k][7] = c8[k];
}

orc_sobol *orc_sobol_new(void) { /* grids/sobol.lua:27-56 */
  orc_sobol *g = (orc_sobol *)calloc(1, sizeof(orc_sobol));
  g->seed = -1;
  g->dims = 0;
  g->initialized = 0;
  orc_create_bank(g);
  return g;
}
void orc_sobol_free(orc_sobol *g) { free(g); }

/* grids/sobol.lua:216-335.  Returns 0, or -1 on the "Too many calls" branch (:317-324). */
int orc_i4_sobol(orc_sobol *g, int dims, double *seed_io, double *quasi) {
  double atmost = ldexp(1.0, ORC_LOG_MAX) - 1; /* :233 */
  g->maxcol = orc_i4_bit_hi1(atmost);         /* :234 */
  int l = 1;

  if (!g->initialized) { /* :237-241 */
    g->initialized = 1;
    for (int j = 0; j < g->maxcol; ++j) g->bank[0][j] = 1.0;
    g->dims = 0;
  }

  if (dims != g->dims) { /* :243-289 */
    g->dims = dims;
    for (int i = 0; i < dims; ++i) {
      double j = floor(orc_poly[i] / 2);
      int m = 0;
      while (j > 0) {
        m += 1;
        j = floor(j / 2);
      }
      int includ[16] = {0};
      j = orc_poly[i];
      for (int k = m; k >= 1; --k) {
        double j2 = floor(j / 2);
        if (j != 2 * j2) includ[k - 1] = 1;
        j = j2;
      }
      for (int jj = m + 1; jj <= g->maxcol; ++jj) {
        double v = g->bank[i][jj - m - 1];
        double ll = 1;
        for (int k = 1; k <= m; ++k) {
          ll = 2 * ll;
          if (includ[k - 1] == 1) v = orc_bitwise_xor(v, ll * g->bank[i][jj - k - 1]);
        }
        g->bank[i][jj - 1] = v;
      }
    }
    double ll = 1;
    for (int j = g->maxcol - 1; j >= 1; --j) { /* :281-285 */
      ll = ll * 2;
      for (int i = 0; i < dims; ++i) g->bank[i][j - 1] *= ll;
    }
    g->recipd = 0.5 / ll; /* :287 */
    memset(g->lastq, 0, sizeof(g->lastq));
  }

  double seed = fmax(0.0, floor(*seed_io)); /* :291 */

  if (seed == 0) { /* :293-315 */
    l = 1;
    memset(g->lastq, 0, sizeof(g->lastq));
  } else if (seed == g->seed + 1) {
    l = orc_i4_bit_lo0(seed);
  } else if (seed <= g->seed) {
    g->seed = 0;
    l = 1;
    memset(g->lastq, 0, sizeof(g->lastq));
    for (double st = g->seed; st <= seed - 1; st += 1) {
      l = orc_i4_bit_lo0(st);
      for (int i = 0; i < dims; ++i) g->lastq[i] = orc_bitwise_xor(g->lastq[i], g->bank[i][l - 1]);
    }
    l = orc_i4_bit_lo0(seed);
  } else if (g->seed + 1 < seed) {
    for (double st = g->seed + 1; st <= seed - 1; st += 1) {
      l = orc_i4_bit_lo0(st);
      for (int i = 0; i < dims; ++i) g->lastq[i] = orc_bitwise_xor(g->lastq[i], g->bank[i][l - 1]);
    }
    l = orc_i4_bit_lo0(seed);
  }

  if (g->maxcol < l) return -1; /* :317-324 */

  for (int i = 0; i < dims; ++i) { /* :327-331 */
    quasi[i] = g->lastq[i] * g->recipd;
    g->lastq[i] = orc_bitwise_xor(g->lastq[i], g->bank[i][l - 1]);
  }
  g->seed = seed;
  *seed_io = seed + 1;
  return 0;
}

/* grids/sobol.lua:58-90 generate with both mins and maxes, or neither (the one-sided maps: orc_affine on the result).
 * out is size x dims. */
int orc_sobol_generate(int64_t size, int dims, int64_t skip, const double *mins, const double *maxes,
                       double *out) {
  if (!(dims >= 1 && dims < ORC_MAX_DIMS)) return -2; /* :36 assert(dims < max_dims) */
  orc_sobol *g = orc_sobol_new();
  for (int64_t j = 1; j <= size; ++j) {
    double seed = (double)(j + skip - 1); /* :75 */
    if (orc_i4_sobol(g, dims, &seed, out + (j - 1) * dims) != 0) {
      orc_sobol_free(g);
      return -1;
    }
  }
  if (mins && maxes) { /* :79-81: cmul by (maxes + -mins), then add mins */
    for (int64_t j = 0; j < size; ++j)
      for (int i = 0; i < dims; ++i) {
        double w = maxes[i] + (-mins[i]);
        double v = out[j * dims + i] * w;
        out[j * dims + i] = v + mins[i];
      }
  }
  orc_sobol_free(g);
  return 0;
}

/* The scaled direction numbers after initialisation (for inspecting the table itself). */
int orc_sobol_bank(int dims, double *bank_out /* dims x 30 */) {
  if (!(dims >= 1 && dims < ORC_MAX_DIMS)) return -2;
  orc_sobol *g = orc_sobol_new();
  double seed = 0, q[ORC_MAX_DIMS];
  orc_i4_sobol(g, dims, &seed, q);
  for (int i = 0; i < dims; ++i)
    for (int j = 0; j < ORC_LOG_MAX; ++j) bank_out[i * ORC_LOG_MAX + j] = g->bank[i][j];
  orc_sobol_free(g);
  return 0;
}

/* grids/sobol.lua:79-85 = grids/random.lua:27-33: the map applied to a generated grid (the RNG stream of grids/random.lua
 * is Torch's).  Both given: cmul by (maxes + -mins), then add mins.  mins only (:82-83): grid:add(torch.add(mins,
 * grid:min(1)[1])) -- the column minimum is ADDED to mins, as the reference writes it.  maxes only (:84-85):
 * grid:cmul(torch.cdiv(maxes, grid:max(1)[1])).  One rounded operation per column for the shift / scale, one per element. */
void orc_affine(double *grid, int64_t size, int dims, const double *mins, const double *maxes) {
  if (mins && maxes) {
    for (int64_t j = 0; j < size; ++j)
      for (int i = 0; i < dims; ++i) {
        double w = maxes[i] + (-mins[i]);
        double v = grid[j * dims + i] * w;
        grid[j * dims + i] = v + mins[i];
      }
  } else if (mins && size > 0) {
    for (int i = 0; i < dims; ++i) {
      double lo = grid[i];
      for (int64_t j = 1; j < size; ++j)
        if (grid[j * dims + i] < lo) lo = grid[j * dims + i];
      double shift = mins[i] + lo;
      for (int64_t j = 0; j < size; ++j) grid[j * dims + i] = grid[j * dims + i] + shift;
    }
  } else if (maxes && size > 0) {
    for (int i = 0; i < dims; ++i) {
      double hi = grid[i];
      for (int64_t j = 1; j < size; ++j)
        if (grid[j * dims + i] > hi) hi = grid[j * dims + i];
      double scale = maxes[i] / hi;
      for (int64_t j = 0; j < size; ++j) grid[j * dims + i] = grid[j * dims + i] * scale;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * utils/math.lua:261-288 erf (Abramowitz & Stegun 7.1.26), :293-300 norm_pdf, :305-312 norm_cdf.
 * ---------------------------------------------------------------------------------------------- */
static const double ORC_SQRT2_INV = 0.70710678118654746;   /* 1/math.sqrt(2)         utils/math.lua:13 */
static const double ORC_SQRT2PI_INV = 0.3989422804014327; /* 1/math.sqrt(2*math.pi) utils/math.lua:15 */

double orc_erf(double x) {
  const double c1 = 0.254829592, c2 = -0.284496736, c3 = 1.421413741, c4 = -1.453152027, c5 = 1.061405429,
               p = 0.3275911;                     /* :263-265 */
  double t = 1.0 / ((fabs(x) * p) + 1.0);         /* :280 abs, mul(p), add(1), pow(-1) */
  double r = t * c5;                              /* :281 mul(buffer,c5) */
  r = r + c4;
  r = r * t;
  r = r + c3;
  r = r * t;                                      /* :282 */
  r = r + c2;
  r = r * t;
  r = r + c1;
  r = r * t;
  double e = exp((x * x) * -1.0);                 /* :283 pow(src,2):mul(-1):exp() */
  r = ((r * e) * -1.0) + 1.0;                     /* :284 */
  double s = ((x >= 0.0) ? 1.0 : 0.0) * 2.0 + -1.0; /* :285 ge(src,0):mul(2):add(-1) */
  return r * s;                                   /* :286 */
}

double orc_norm_cdf(double z) { /* :308-310 */
  double u = z * ORC_SQRT2_INV;
  return (orc_erf(u) + 1.0) * 0.5;
}

double orc_norm_pdf(double z) { /* :298 */
  return exp((z * z) * -0.5) * ORC_SQRT2PI_INV;
}

/* ------------------------------------------------------------------------------------------------
 * scores/expected_improvement.lua:69-88 EI.compute.  mean is M x c (row-major), var is M, fmin is c.
 * For c > 1 the row mean is taken (:83-85) as sum over columns in ascending order then one division.
 * ---------------------------------------------------------------------------------------------- */
void orc_ei(const double *mean, const double *var, const double *fmin, double tradeoff, int64_t M, int c,
            double *out) {
  for (int64_t j = 0; j < M; ++j) {
    double sigma = sqrt(var[j]); /* :73 */
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      double imprv = (fmin[k] + (-mean[j * c + k])) + (-tradeoff); /* :74 */
      double z = imprv / sigma;                                    /* :75 */
      double ei = (imprv * orc_norm_cdf(z)) + (sigma * orc_norm_pdf(z)); /* :78-79 */
      ei = (ei < 0.0) ? 0.0 : ei; /* :80 clamp(0, huge): NaN passes */
      if (c == 1)
        acc = ei;
      else
        acc += ei;
    }
    out[j] = (c == 1) ? acc : acc / (double)c;
  }
}

/* scores/confidence_bound.lua:70-106.  upper: 0 -> LCB (:102-106), 1 -> UCB (:96-100). */
void orc_cb(const double *mean, const double *var, double tradeoff, int upper, double sign, int64_t M, int c,
            double *out) {
  for (int64_t j = 0; j < M; ++j) {
    double s = sqrt(var[j]) * tradeoff;
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      double v = upper ? (mean[j * c + k] + s) : (mean[j * c + k] + (-s));
      if (c == 1)
        acc = v;
      else
        acc += v;
    }
    double val = (c == 1) ? acc : acc / (double)c; /* :84-86 */
    out[j] = (sign > 0.0) ? val : -val;            /* :89-93 */
  }
}

/* bots/bayesopt.lua:69-79: score:add(...) per sample, then score:div(nSamples). */
void orc_accumulate(double *acc, const double *score, int64_t M) {
  for (int64_t j = 0; j < M; ++j) acc[j] = acc[j] + score[j];
}
void orc_divide(double *acc, double divisor, int64_t M) {
  for (int64_t j = 0; j < M; ++j) acc[j] = acc[j] / divisor;
}

/* bots/bayesopt.lua:96 score:max(1): TH's max keeps the earlier element on ties and lets the first NaN
 * win ("if (!(value <= theMax))" then break on NaN) [TH semantics are public knowledge, not in reference].
 * Returns the 1-based index. */
int64_t orc_argmax_first(const double *s, int64_t M, double *val_out) {
  if (M <= 0) return 0;
  int64_t idx = 0;
  double best = s[0];
  for (int64_t i = 0; i < M; ++i) {
    double v = s[i];
    if (!(v <= best)) {
      idx = i;
      best = v;
      if (v != v) break;
    }
  }
  if (val_out) *val_out = best;
  return idx + 1;
}

/* utils/tensor.lua:158-170 remove (one index): rows after idx1 shift up by one, order preserved. */
void orc_remove_row(const double *src, int64_t M, int d, int64_t idx1, double *dst /* (M-1) x d */) {
  int64_t w = 0;
  for (int64_t j = 0; j < M; ++j) {
    if (j + 1 == idx1) continue;
    memcpy(dst + w * d, src + j * d, sizeof(double) * (size_t)d);
    ++w;
  }
}

/* ------------------------------------------------------------------------------------------------
 * utils/math.lua:65-111 pdist with lenscale (GEMM expansion), p = 2, no root.
 *   inv_ls = 1/lenscale (:72);  X_ss = (X.^2) * inv_ls (:78);  Z_ss likewise (:79)
 *   dist = X * (Z' .* inv_ls) (:80-82), then *(-2), + X_ss, + Z_ss' (:82), clamp(0, huge) (:106).
 * BLAS summation order is implementation-defined; this restatement sums k ascending.
 * Z == NULL means Z = X (:83-89).
 * ---------------------------------------------------------------------------------------------- */
void orc_pdist(const double *X, int64_t M, const double *Z, int64_t N, int d, const double *lenscale,
               double *out /* M x N */) {
  if (!Z) {
    Z = X;
    N = M;
  }
  double *inv = (double *)malloc(sizeof(double) * (size_t)d);
  double *xss = (double *)malloc(sizeof(double) * (size_t)M);
  double *zss = (double *)malloc(sizeof(double) * (size_t)N);
  for (int k = 0; k < d; ++k) inv[k] = 1.0 / lenscale[k];
  for (int64_t i = 0; i < M; ++i) {
    double a = 0.0;
    for (int k = 0; k < d; ++k) a += (X[i * d + k] * X[i * d + k]) * inv[k];
    xss[i] = a;
  }
  for (int64_t j = 0; j < N; ++j) {
    double a = 0.0;
    for (int k = 0; k < d; ++k) a += (Z[j * d + k] * Z[j * d + k]) * inv[k];
    zss[j] = a;
  }
  for (int64_t i = 0; i < M; ++i)
    for (int64_t j = 0; j < N; ++j) {
      double a = 0.0;
      for (int k = 0; k < d; ++k) a += X[i * d + k] * (Z[j * d + k] * inv[k]);
      double v = ((a * -2.0) + xss[i]) + zss[j];
      out[i * N + j] = (v < 0.0) ? 0.0 : v;
    }
  free(inv);
  free(xss);
  free(zss);
}

/* Unblocked lower Cholesky in place (the arithmetic torch.potrf delegates to LAPACK dpotrf,
 * utils/math.lua:165).  Returns 0, or the 1-based index of the first non-positive pivot. */
int orc_potrf_lower(double *A, int n) {
  for (int j = 0; j < n; ++j) {
    double ajj = A[j * n + j];
    for (int k = 0; k < j; ++k) ajj -= A[j * n + k] * A[j * n + k];
    if (!(ajj > 0.0)) return j + 1;
    ajj = sqrt(ajj);
    A[j * n + j] = ajj;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / ajj;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A[i * n + j] = 0.0;
  return 0;
}

/* utils/math.lua:159-218 jitter schedule around orc_potrf_lower.
 *   try potrf(src); while failing: if eps > max_eps then potrf(I) else eps *= growth; potrf(src + eps*I)
 *   defaults eps 1e-8, growth 1.1, max_eps = ||src||_F (:174-176).
 * jitter_out: 0 if the first attempt succeeded, the eps used otherwise, -1 if it fell back to chol(I). */
int orc_chol_jitter(const double *src, int n, double *res, double *jitter_out) {
  memcpy(res, src, sizeof(double) * (size_t)n * n);
  *jitter_out = 0.0;
  if (orc_potrf_lower(res, n) == 0) return 0;
  double max_eps = 0.0;
  for (int64_t i = 0; i < (int64_t)n * n; ++i) max_eps += src[i] * src[i];
  max_eps = sqrt(max_eps);
  double eps = 1e-8, growth = 1.1;
  int itr = 0;
  for (;;) {
    itr += 1;
    if (eps > max_eps) {
      memset(res, 0, sizeof(double) * (size_t)n * n);
      for (int i = 0; i < n; ++i) res[i * n + i] = 1.0;
      *jitter_out = -1.0;
      return itr;
    }
    eps = eps * growth;
    memcpy(res, src, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) res[i * n + i] = src[i * n + i] + eps;
    if (orc_potrf_lower(res, n) == 0) {
      *jitter_out = eps;
      return itr;
    }
  }
}
