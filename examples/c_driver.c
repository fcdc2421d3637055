/* A bot7 trial loop written against include/bot7hip.h alone, in C99 -- what any host language with a C FFI does (the LuaJIT
 * shims in lua/ make exactly these calls).  BASELINE config 1 in miniature: braninhoo on [0,1]^2, a 256-point grid from
 * torch.rand's stream (b7_grid_random_torch, seed 7), nInitial = 2 random picks (bots/bayesopt.lua:90-91, drawn from the same
 * MT19937 stream continued: b7_torch_rand), then GP + EI nominations with fixed hypers (one "sample"), the nominee stolen from
 * the candidate set (bots/abstract.lua:118) with b7_nominate_commit, until `trials` observations.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_driver.c -o c_driver -Lbot7_amd -lbot7hip -lm -Wl,-rpath,$PWD/bot7_amd
 *   ./c_driver [trials] [n_virtual_ranks]
 *
 * With n_virtual_ranks > 1 the same loop runs over a single-process group (b7_group_*) of that many contexts on device 0.
 * Prints one line per trial: trial, global index nominated, x, y; then the best. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "bot7hip.h"

static double braninhoo(const double *x) { /* benchmarks/braninhoo.lua:24-44 */
  const double pi = 3.14159265358979323846;
  const double c1 = -5.1 / (4.0 * pi * pi), c2 = 5.0 / pi, c3 = 10.0 - 10.0 / (8.0 * pi);
  const double z1 = x[0] * 15.0 - 5.0, z2 = x[1] * 15.0;
  const double t = z2 + z1 * z1 * c1 + z1 * c2 - 6.0;
  return t * t + (cos(z1) * c3 + 10.0);
}

#define CK(expr)                                                                         \
  do {                                                                                   \
    int rc_ = (expr);                                                                    \
    if (rc_ != B7_OK) {                                                                  \
      fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, g ? b7_group_last_error(g) : b7_last_error(c)); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

int main(int argc, char **argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 12, nranks = argc > 2 ? atoi(argv[2]) : 1;
  enum { D = 2, M0 = 256, NINIT = 2 };
  const double mins[D] = {0.0, 0.0}, maxes[D] = {1.0, 1.0};
  b7_ctx *c = NULL;
  b7_group *g = NULL;
  double grid[M0 * D], picks[NINIT];
  double X[64 * D], Y[64], row[D], best_y = INFINITY, best_x[D] = {0.0, 0.0};
  int64_t offset = 0, M = M0;
  int n = 0, best_t = 0;
  if (trials < 3 || trials > 64 || nranks < 1 || nranks > 16) return 2;
  if (nranks > 1) {
    int ids[16] = {0};
    if (b7_group_create(&g, nranks, ids) != B7_OK) {
      fprintf(stderr, "b7_group_create: %s\n", b7_last_error(NULL));
      return 1;
    }
    /* the grid is made once from torch.rand's stream (host) and handed to the group, which shards it */
    b7_torch_rand(7, M0 * D, 32, grid);
    CK(b7_group_grid_upload(g, grid, M0, D));
  } else {
    if (b7_create(&c, 0) != B7_OK) {
      fprintf(stderr, "b7_create: %s\n", b7_last_error(NULL));
      return 1;
    }
    CK(b7_grid_random_torch(c, M0, D, 7, 32, mins, maxes, grid));
  }
  { /* the random initial picks continue the same stream (the draws after the grid's M0 * D) */
    double *u = (double *)malloc(sizeof(double) * (M0 * D + NINIT));
    b7_torch_rand(7, M0 * D + NINIT, 32, u);
    for (int t = 0; t < NINIT; ++t) picks[t] = u[M0 * D + t];
    free(u);
  }
  for (int t = 1; t <= trials; ++t) {
    int64_t idx = 0;
    double val = 0.0;
    if (t <= NINIT) {
      idx = (int64_t)floor(picks[t - 1] * (double)M) + 1; /* torch.rand(1):mul(M):long():add(1) */
    } else {
      double lsq[D] = {0.25, 0.25}, mean = 0.0, var = 0.0, fmin = Y[0], jitter = 0.0;
      int info = 0;
      b7_hyp hyp;
      b7_score_spec spec;
      for (int i = 0; i < n; ++i) mean += Y[i] / n;
      for (int i = 0; i < n; ++i) var += (Y[i] - mean) * (Y[i] - mean) / n;
      for (int i = 1; i < n; ++i) fmin = Y[i] < fmin ? Y[i] : fmin;
      hyp.lenscale_sq = lsq, hyp.amp = var > 0.0 ? var : 1.0, hyp.noise = 1e-4 * hyp.amp, hyp.mean = mean;
      spec.kind = B7_SCORE_EI, spec.tradeoff = 0.0, spec.upper = 0, spec.sign = 0.0, spec.fmin = &fmin;
      if (g) {
        CK(b7_group_gp_set_data(g, X, Y, n, D, 1));
        CK(b7_group_eval_nominate(g, 1, &hyp, &spec, &val, &idx, &jitter, &info));
      } else {
        CK(b7_gp_set_data(c, X, Y, n, D, 1));
        CK(b7_eval_nominate(c, 1, &hyp, &spec, offset, &val, &idx, &jitter, &info));
      }
    }
    if (g) CK(b7_group_nominate_commit(g, idx, row));
    else CK(b7_nominate_commit(c, idx, &offset, row));
    M -= 1;
    X[n * D] = row[0], X[n * D + 1] = row[1];
    Y[n] = braninhoo(row);
    if (Y[n] < best_y) best_y = Y[n], best_x[0] = row[0], best_x[1] = row[1], best_t = t;
    printf("trial %2d  idx %3lld  x = (%.17g, %.17g)  y = %.17g\n", t, (long long)idx, row[0], row[1], Y[n]);
    n += 1;
  }
  printf("best y = %.17g at (%.17g, %.17g), trial %d\n", best_y, best_x[0], best_x[1], best_t);
  if (g) b7_group_destroy(g);
  else b7_destroy(c);
  return 0;
}
