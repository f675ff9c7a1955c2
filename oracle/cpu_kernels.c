/* Threaded elementwise legs of the CPU baseline (oracle/cpu_port.py, bench.py's cpu_baseline, kind "port").
 * TEST / MEASUREMENT INFRASTRUCTURE: nothing in bobe_amd links or loads this file.
 *
 * The torch-CPU form of the baseline spent 1.1 s of a 1.5 s value+gradient evaluation at N = 4096 in these two legs
 * (profiles/r03_cpu_port_breakdown.txt: assembly 0.31 s, gradient reductions 0.79 s, LAPACK 0.40 s): every N x N
 * temporary of `(WK * df * df).sum()` is a pass over memory.  A competent CPU port fuses them, which is all this file
 * does: same formulas, same direct-difference distances (gp.py:80-96, 124-154; gradient as in
 * bobe_oracle.mll_value_and_grad), one pass, OpenMP over row blocks, the symmetric half only.
 *
 *   gcc -O3 -fopenmp -mavx2 -mfma  (no -march=native: the .so is built in the build container and travels to the GPU box)
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>

#define BLK 64

/* Kt = kvar * exp(-0.5 |xs_i - xs_j|^2) for all i, j (xs already divided by the length scales), full symmetric matrix,
 * computed on the lower 64 x 64 blocks and mirrored; diag_add is added on the diagonal (noise, or 0) */
void ck_rbf_sym(const double* xs, int64_t n, int d, double kvar, double diag_add, double* K, int nthreads) {
  const int64_t nbk = (n + BLK - 1) / BLK;
  const int64_t ntask = nbk * (nbk + 1) / 2;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int64_t t = 0; t < ntask; ++t) {
    int64_t bi = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int64_t bj = t - bi * (bi + 1) / 2;
    const int64_t i0 = bi * BLK, j0 = bj * BLK;
    const int64_t i1 = i0 + BLK < n ? i0 + BLK : n, j1 = j0 + BLK < n ? j0 + BLK : n;
    for (int64_t i = i0; i < i1; ++i) {
      const double* xi = xs + i * d;
      double* row = K + i * n;
      for (int64_t j = j0; j < j1; ++j) {
        const double* xj = xs + j * d;
        double r2 = 0.0;
        for (int q = 0; q < d; ++q) {
          const double df = xi[q] - xj[q];
          r2 += df * df;
        }
        row[j] = kvar * exp(-0.5 * r2);
      }
    }
    if (bi != bj) {
      for (int64_t j = j0; j < j1; ++j)
        for (int64_t i = i0; i < i1; ++i) K[j * n + i] = K[i * n + j];
    } else {
      for (int64_t i = i0; i < i1; ++i) K[i * n + i] += diag_add;
    }
  }
}

/* out[i][j] = kvar * exp(-0.5 |xa_i - xb_j|^2), na x nb (K(X, C), K(X, Z), K(C, Z)) */
void ck_rbf_rect(const double* xa, int64_t na, const double* xb, int64_t nb, int d, double kvar, double* out, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t i = 0; i < na; ++i) {
    const double* xi = xa + i * d;
    double* row = out + i * nb;
    for (int64_t j = 0; j < nb; ++j) {
      const double* xj = xb + j * d;
      double r2 = 0.0;
      for (int q = 0; q < d; ++q) {
        const double df = xi[q] - xj[q];
        r2 += df * df;
      }
      row[j] = kvar * exp(-0.5 * r2);
    }
  }
}

/* g[q] = 1/2 sum_ab W_ab Kt_ab (xs_aq - xs_bq)^2 (q < d), g[d] = 1/2 sum_ab W_ab Kt_ab with W = alpha alpha^T - Kinv:
 * the RBF gradient of the data-term MLL wrt (log ls, log kvar).  Kinv and K are read on the lower triangle only
 * (off-diagonal entries count twice); `K_has_diag_add` is subtracted from K's diagonal to recover Kt. */
void ck_grad_rbf(const double* xs, int64_t n, int d, const double* alpha, const double* Kinv, const double* K,
                 double K_has_diag_add, double* g, int nthreads) {
  enum { MAXD = 32 };
  const int nt = nthreads > 0 ? nthreads : 1;
  double acc[256][MAXD + 1];
  if (d > MAXD || nt > 256) {
    for (int q = 0; q <= d; ++q) g[q] = NAN;
    return;
  }
  memset(acc, 0, sizeof(acc));
#pragma omp parallel num_threads(nt)
  {
    double loc[MAXD + 1];
    for (int q = 0; q <= d; ++q) loc[q] = 0.0;
#pragma omp for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
      const double* xi = xs + i * d;
      const double ai = alpha[i];
      const double* ki = Kinv + i * n;
      const double* kr = K + i * n;
      for (int64_t j = 0; j <= i; ++j) {
        const double kt = (j == i) ? kr[j] - K_has_diag_add : kr[j];
        const double wk = (ai * alpha[j] - ki[j]) * kt * ((j == i) ? 1.0 : 2.0);
        const double* xj = xs + j * d;
        for (int q = 0; q < d; ++q) {
          const double df = xi[q] - xj[q];
          loc[q] += wk * df * df;
        }
        loc[d] += wk;
      }
    }
    const int me = omp_get_thread_num();
    for (int q = 0; q <= d; ++q) acc[me][q] = loc[q];
  }
  for (int q = 0; q <= d; ++q) {
    double s = 0.0;
    for (int t = 0; t < nt; ++t) s += acc[t][q];      /* fixed order for a given thread count */
    g[q] = 0.5 * s;
  }
}

int ck_max_threads(void) { return omp_get_max_threads(); }
