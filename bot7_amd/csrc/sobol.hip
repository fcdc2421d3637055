// Candidate-grid kernels: Sobol (closed form of the reference's Gray-code recurrence), counter-based uniform
// grid, and the stable row deletion behind utils.tensor.remove.
//
// Reference behaviour replaced:
//   grids/sobol.lua:58-90  generate       row j = point number j+skip-1, then x*(maxes-mins)+mins
//   grids/sobol.lua:216-335 i4_sobol      lastq ^= bank[i][lo0(seed)] per call; quasi = lastq * 2^-30
//   grids/sobol.lua:338-391 create_bank   initial direction numbers; :46-52 primitive polynomials
//   grids/random.lua:23-35                uniform grid + same affine map
//   utils/tensor.lua:158-170              remove: stable deletion of one row
//
// The recurrence XORs bank[i][lo0(k)] for k = 0..n-1 into lastq before emitting point n.  lo0(k) = l
// happens an odd number of times for k < n exactly when bit l-1 of the Gray code n ^ (n>>1) is set, so
//   q_n[i] = XOR over set bits b of gray(n) of V[i][b],   x_n[i] = q_n[i] * 2^-30,
// which needs no state and lets every (row, dim) be computed independently (and every GPU generate its own
// slice of the sequence).  Integer-exact; the two affine roundings are kept separate (contract off).
#pragma clang fp contract(off)
#include "b7_internal.h"

namespace {

constexpr int SOBOL_DIMS = 40;  // table rows (the reference admits dims < 40, grids/sobol.lua:36)
constexpr int SOBOL_BITS = 30;  // log_max, grids/sobol.lua:32

struct SobolTable {
  uint32_t v[SOBOL_DIMS][SOBOL_BITS];
};

// Direction numbers V[i][b] = m_{b+1} * 2^(29-b), Bratley & Fox section 2, from the reference's tables.
SobolTable make_table() {
  static const unsigned short poly[SOBOL_DIMS] = {1,   3,   7,   11,  13,  19,  25,  37,  59,  47,
                                                  61,  55,  41,  67,  97,  91,  109, 103, 115, 131,
                                                  193, 137, 145, 143, 241, 157, 185, 167, 229, 171,
                                                  213, 191, 253, 203, 211, 239, 247, 285, 369, 299};
  // Initial m values per dimension (rows of create_bank read across), degree = number of entries.
  static const unsigned char init[SOBOL_DIMS][8] = {
      {1},                              // dim 1: every m_j = 1 (grids/sobol.lua:239)
      {1},                              // poly 3
      {1, 1},                           // poly 7
      {1, 3, 7},                        // 11
      {1, 1, 5},                        // 13
      {1, 3, 1, 1},                     // 19
      {1, 1, 3, 7},                     // 25
      {1, 3, 3, 9, 9},                  // 37
      {1, 3, 7, 13, 3},                 // 59
      {1, 1, 5, 11, 27},                // 47
      {1, 3, 5, 1, 15},                 // 61
      {1, 1, 7, 3, 29},                 // 55
      {1, 3, 7, 7, 21},                 // 41
      {1, 1, 1, 9, 23, 37},             // 67
      {1, 3, 3, 5, 19, 33},             // 97
      {1, 1, 3, 13, 11, 7},             // 91
      {1, 1, 7, 13, 25, 5},             // 109
      {1, 3, 5, 11, 7, 11},             // 103
      {1, 1, 1, 3, 13, 39},             // 115
      {1, 3, 1, 15, 17, 63, 13},        // 131
      {1, 1, 5, 5, 1, 27, 33},          // 193
      {1, 3, 3, 3, 25, 17, 115},        // 137
      {1, 1, 3, 15, 29, 15, 41},        // 145
      {1, 3, 1, 7, 3, 23, 79},          // 143
      {1, 3, 7, 9, 31, 29, 17},         // 241
      {1, 1, 5, 13, 11, 3, 29},         // 157
      {1, 3, 1, 9, 5, 21, 119},         // 185
      {1, 1, 3, 1, 23, 13, 75},         // 167
      {1, 3, 3, 11, 27, 31, 73},        // 229
      {1, 1, 7, 7, 19, 25, 105},        // 171
      {1, 3, 5, 5, 21, 9, 7},           // 213
      {1, 1, 1, 15, 5, 49, 59},         // 191
      {1, 1, 1, 1, 1, 33, 65},          // 253
      {1, 3, 5, 15, 17, 19, 21},        // 203
      {1, 1, 7, 11, 13, 29, 3},         // 211
      {1, 3, 7, 5, 7, 11, 113},         // 239
      {1, 1, 5, 3, 15, 19, 61},         // 247
      {1, 3, 1, 1, 9, 27, 89, 7},       // 285
      {1, 1, 3, 7, 31, 15, 45, 23},     // 369
      {1, 3, 3, 9, 9, 25, 107, 39},     // 299
  };
  SobolTable t{};
  for (int i = 0; i < SOBOL_DIMS; ++i) {
    uint32_t m[SOBOL_BITS + 1] = {0};
    int deg = 0;
    for (unsigned p = poly[i] >> 1; p; p >>= 1) ++deg;
    if (i == 0) {
      for (int j = 1; j <= SOBOL_BITS; ++j) m[j] = 1;
    } else {
      for (int j = 1; j <= deg; ++j) m[j] = init[i][j - 1];
      for (int j = deg + 1; j <= SOBOL_BITS; ++j) {
        uint32_t nv = m[j - deg];
        for (int k = 1; k <= deg; ++k) {
          // coefficient of x^(deg-k) in the polynomial (bits below the leading one, high to low)
          if ((poly[i] >> (deg - k)) & 1u) nv ^= (m[j - k] << k);
        }
        m[j] = nv;
      }
    }
    for (int j = 1; j <= SOBOL_BITS; ++j) t.v[i][j - 1] = m[j] << (SOBOL_BITS - j);
  }
  return t;
}

const SobolTable &table() {
  static const SobolTable t = make_table();
  return t;
}

__global__ void __launch_bounds__(256) sobol_kernel(double *__restrict__ out, int64_t total, int dims, int64_t skip,
                                                    const uint32_t *__restrict__ vtab, const double *__restrict__ mm,
                                                    int affine) {
  __shared__ uint32_t sv[SOBOL_BITS][SOBOL_DIMS];  // [bit][dim]: lanes of a wave read consecutive dims
  __shared__ double smin[SOBOL_DIMS], sspan[SOBOL_DIMS];
  for (int t = threadIdx.x; t < SOBOL_BITS * SOBOL_DIMS; t += blockDim.x) {
    int b = t / SOBOL_DIMS, i = t % SOBOL_DIMS;
    sv[b][i] = (i < dims) ? vtab[i * SOBOL_BITS + b] : 0u;
  }
  if (threadIdx.x < dims) {
    double lo = affine ? mm[threadIdx.x] : 0.0;
    double hi = affine ? mm[dims + threadIdx.x] : 1.0;
    smin[threadIdx.x] = lo;
    sspan[threadIdx.x] = hi + (-lo);  // torch.add(maxes, -mins), grids/sobol.lua:80
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int64_t row = e / dims;
    int col = (int)(e - row * dims);
    int64_t n = row + skip;  // seed = j + skip - 1 with j = row + 1 (:75)
    if (n < 0) n = 0;        // seed = max(0, floor(seed)) (:291)
    uint32_t g = (uint32_t)(n ^ (n >> 1));
    uint32_t q = 0;
#pragma unroll
    for (int b = 0; b < SOBOL_BITS; ++b) q ^= ((g >> b) & 1u) ? sv[b][col] : 0u;
    double x = (double)q * 9.31322574615478515625e-10;  // recipd = 2^-30 (:287), exact
    if (affine) {
      x = x * sspan[col];  // cmul (:80)
      x = x + smin[col];   // add  (:81)
    }
    out[e] = x;
  }
}

__device__ inline uint64_t splitmix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) random_grid_kernel(double *__restrict__ out, int64_t total, int dims,
                                                          uint64_t seed, int64_t row_offset,
                                                          const double *__restrict__ mm, int affine) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int64_t row = e / dims;
    int col = (int)(e - row * dims);
    uint64_t ctr = (uint64_t)(row_offset + row) * (uint64_t)dims + (uint64_t)col;
    uint64_t z = splitmix64(seed + 0x9E3779B97F4A7C15ull * (ctr + 1));
    double x = (double)(z >> 11) * 1.1102230246251565404e-16;  // 2^-53
    if (affine) {
      double lo = mm[col], hi = mm[dims + col];
      x = x * (hi + (-lo));  // grids/random.lua:28
      x = x + lo;
    }
    out[e] = x;
  }
}

// grid:min(1) / grid:max(1) (grids/sobol.lua:83,85): column minima and maxima.  A launch uses a thread count that is a
// multiple of d, so a thread stays in one column while it strides through the (row-major, coalesced) grid; a block folds
// its threads per column through LDS and writes one partial per column, the last kernel folds the blocks.  min / max are
// exact in any order.
__global__ void __launch_bounds__(256) colrange_kernel(const double *__restrict__ g, int64_t total, int d, int64_t nthreads,
                                                       double *__restrict__ part) {
  __shared__ double smin[256], smax[256];
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double lo = __builtin_inf(), hi = -__builtin_inf();
  if (t < nthreads)
    for (int64_t e = t; e < total; e += nthreads) {
      const double x = g[e];
      lo = x < lo ? x : lo;
      hi = x > hi ? x : hi;
    }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  if ((int)threadIdx.x < d) {
    // threads j of this block with (block base + j) % d == my column
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int col = (int)((base + threadIdx.x) % d);
    for (int j = threadIdx.x + d; j < (int)blockDim.x; j += d) {
      lo = smin[j] < lo ? smin[j] : lo;
      hi = smax[j] > hi ? smax[j] : hi;
    }
    part[((size_t)blockIdx.x * d + col) * 2] = lo;
    part[((size_t)blockIdx.x * d + col) * 2 + 1] = hi;
  }
}
__global__ void __launch_bounds__(128) colrange_final_kernel(const double *__restrict__ part, int nblk, int d,
                                                             double *__restrict__ out) {
  const int k = threadIdx.x;
  if (k >= d) return;
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int b = 0; b < nblk; ++b) {
    const double a = part[((size_t)b * d + k) * 2], z = part[((size_t)b * d + k) * 2 + 1];
    lo = a < lo ? a : lo;
    hi = z > hi ? z : hi;
  }
  out[k] = lo;
  out[d + k] = hi;
}

// x += v[col] (grids/sobol.lua:83) or x *= v[col] (:85): one rounded operation per element
__global__ void __launch_bounds__(256) col_affine_kernel(double *__restrict__ g, int64_t total, int d,
                                                         const double *__restrict__ v, int mul) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int col = (int)(e % d);
    g[e] = mul ? g[e] * v[col] : g[e] + v[col];
  }
}

__global__ void __launch_bounds__(256) remove_row_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                         int64_t total_out, int64_t cut, int d) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_out; e += stride)
    dst[e] = src[e >= cut ? e + d : e];
}

// Stable deletion of several rows at once (utils.tensor.remove takes an index tensor, utils/tensor.lua:158-170):
// cuts[i] = (i-th smallest removed 0-based row) - i, ascending and non-decreasing; output row j comes from source
// row j + #{i : cuts[i] <= j} (upper bound by bisection: n <= a few thousand, 11-12 steps).
__global__ void __launch_bounds__(256) remove_rows_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                          int64_t total_out, const int64_t *__restrict__ cuts,
                                                          int ncut, int d) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_out; e += stride) {
    const int64_t row = e / d;
    int lo = 0, hi = ncut;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cuts[mid] <= row) lo = mid + 1; else hi = mid;
    }
    dst[e] = src[e + (int64_t)lo * d];
  }
}

// rows_out[i][:] = src[idx0[i]][:]  (src:index(axis, idx), the rows steal appends to `pending`)
__global__ void __launch_bounds__(256) gather_rows_kernel(const double *__restrict__ src, double *__restrict__ out,
                                                          const int64_t *__restrict__ idx0, int64_t total, int d) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t i = e / d;
  out[e] = src[idx0[i] * d + (e - i * d)];
}

int grid_blocks(b7_ctx *c, int64_t total) {
  int64_t b = (total + 255) / 256;
  int64_t cap = (int64_t)c->cus * 8;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// mins/maxes (host, nullable) -> device scratch [2*dims]; returns device pointer or nullptr.
int upload_minmax(b7_ctx *c, const double *mins, const double *maxes, int dims, const double **dev) {
  *dev = nullptr;
  if (!mins || !maxes) return B7_OK;
  double *p = (double *)((char *)c->scratch.p + 4096 + sizeof(SobolTable));
  B7_HIP(c, hipMemcpyAsync(p, mins, sizeof(double) * dims, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipMemcpyAsync(p + dims, maxes, sizeof(double) * dims, hipMemcpyHostToDevice, c->stream));
  *dev = p;
  return B7_OK;
}

}  // namespace

extern "C" int b7_sobol_direction_numbers(int dims, uint32_t *out) {
  if (dims < 1 || dims >= SOBOL_DIMS || !out) return B7_ERR_RANGE;
  const SobolTable &t = table();
  for (int i = 0; i < dims; ++i)
    for (int b = 0; b < SOBOL_BITS; ++b) out[i * SOBOL_BITS + b] = t.v[i][b];
  return B7_OK;
}

int launch_sobol(b7_ctx *c, double *out, int64_t size, int dims, int64_t skip, const double *mins,
                 const double *maxes) {
  PhaseScope ps(c, "sobol");
  uint32_t *vt = (uint32_t *)((char *)c->scratch.p + 4096);
  B7_HIP(c, hipMemcpyAsync(vt, &table(), sizeof(SobolTable), hipMemcpyHostToDevice, c->stream));
  const double *mm = nullptr;
  B7_TRY(upload_minmax(c, mins, maxes, dims, &mm));
  int64_t total = size * dims;
  if (total > 0) {
    hipLaunchKernelGGL(sobol_kernel, dim3(grid_blocks(c, total)), dim3(256), 0, c->stream, out, total, dims, skip, vt,
                       mm, mm ? 1 : 0);
    B7_HIP(c, hipGetLastError());
  }
  return B7_OK;
}

int launch_random_grid(b7_ctx *c, double *out, int64_t size, int dims, uint64_t seed, int64_t row_offset,
                       const double *mins, const double *maxes) {
  PhaseScope ps(c, "sobol");
  const double *mm = nullptr;
  B7_TRY(upload_minmax(c, mins, maxes, dims, &mm));
  int64_t total = size * dims;
  if (total > 0) {
    hipLaunchKernelGGL(random_grid_kernel, dim3(grid_blocks(c, total)), dim3(256), 0, c->stream, out, total, dims,
                       seed, row_offset, mm, mm ? 1 : 0);
    B7_HIP(c, hipGetLastError());
  }
  return B7_OK;
}

int launch_remove_row(b7_ctx *c, const double *src, double *dst, int64_t M, int d, int64_t idx0) {
  PhaseScope ps(c, "remove");
  int64_t total_out = (M - 1) * d;
  if (total_out > 0) {
    hipLaunchKernelGGL(remove_row_kernel, dim3(grid_blocks(c, total_out)), dim3(256), 0, c->stream, src, dst,
                       total_out, idx0 * d, d);
    B7_HIP(c, hipGetLastError());
  }
  return B7_OK;
}

int launch_remove_rows(b7_ctx *c, const double *src, double *dst, int64_t M, int d, const int64_t *cuts_dev, int ncut) {
  PhaseScope ps(c, "remove");
  int64_t total_out = (M - ncut) * d;
  if (total_out > 0) {
    hipLaunchKernelGGL(remove_rows_kernel, dim3(grid_blocks(c, total_out)), dim3(256), 0, c->stream, src, dst,
                       total_out, cuts_dev, ncut, d);
    B7_HIP(c, hipGetLastError());
  }
  return B7_OK;
}

int launch_gather_rows(b7_ctx *c, const double *src, double *out, const int64_t *idx0_dev, int64_t n, int d) {
  const int64_t total = n * d;
  if (total > 0) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, src, out,
                       idx0_dev, total, d);
    B7_HIP(c, hipGetLastError());
  }
  return B7_OK;
}

// out_dev[0..d) = column minima, [d..2d) = column maxima of the M x d grid (M > 0)
int launch_colrange(b7_ctx *c, const double *grid, int64_t M, int d, double *out_dev) {
  const int64_t total = M * d;
  int nblk = grid_blocks(c, total);
  if (nblk > 256) nblk = 256;
  const int64_t nthreads = (int64_t)nblk * 256 / d * d;  // >= d since d <= 96 < 256
  B7_TRY(b7_ensure(c, c->part, sizeof(double) * 2 * (size_t)nblk * d));
  hipLaunchKernelGGL(colrange_kernel, dim3(nblk), dim3(256), 0, c->stream, grid, total, d, nthreads, (double *)c->part.p);
  hipLaunchKernelGGL(colrange_final_kernel, dim3(1), dim3(128), 0, c->stream, (const double *)c->part.p, nblk, d, out_dev);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_col_affine(b7_ctx *c, double *grid, int64_t M, int d, const double *v_dev, bool mul) {
  const int64_t total = M * d;
  if (total <= 0) return B7_OK;
  hipLaunchKernelGGL(col_affine_kernel, dim3(grid_blocks(c, total)), dim3(256), 0, c->stream, grid, total, d, v_dev, mul ? 1 : 0);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
