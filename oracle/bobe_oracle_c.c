/* Plain-C second restatement of the BOBE GP hot path (TEST INFRASTRUCTURE - not product code).
 *
 * Scalar, unblocked, no BLAS/LAPACK: written independently of oracle/bobe_oracle.py (NumPy/SciPy) so that the
 * two CPU restatements can be checked against each other (tests/test_oracle.py) before either is used to judge the
 * HIP path.  PARITY STATUS: parity unpinned, for the same reason as the NumPy oracle (the reference cannot run here
 * and holds no numeric fixtures).  Only tests/ and __graft_entry__.build() touch this file.
 * Reference citations are file:line of Ameek94/BOBE @ 2025-12-26.  All matrices are row-major fp64.
 *
 *   gcc -O2 -shared -fPIC -o oracle/libbobe_oracle_c.so oracle/bobe_oracle_c.c -lm        (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SAFE_NOISE_FLOOR 1e-12 /* gp.py:16 */

/* gp.py:80-96 + 124-168: k(a, b) for scaled squared distance r2 */
static double kern_value(int kern, double r2, double kvar) {
  if (kern == 0) return kvar * exp(-0.5 * r2);                    /* rbf, gp.py:124-154 */
  double dd = sqrt(r2 < 1e-30 ? 1e-30 : r2);                      /* matern-5/2, gp.py:156-168 */
  return kvar * (1.0 + dd * (sqrt(5.0) + dd * 5.0 / 3.0)) * exp(-sqrt(5.0) * dd);
}

static double scaled_r2(const double* a, const double* b, int d, const double* ls) {
  double r2 = 0.0;
  for (int j = 0; j < d; ++j) {
    double df = a[j] / ls[j] - b[j] / ls[j];
    r2 += df * df;
  }
  return r2;
}

/* K[n1 x n2]; include_noise adds noise on the diagonal (square case only, gp.py:153, 167) */
void oc_kernel(int kern, const double* A, int n1, const double* B, int n2, int d, const double* ls, double kvar,
               double noise, int include_noise, double* K) {
  for (int i = 0; i < n1; ++i)
    for (int k = 0; k < n2; ++k) {
      double v = kern_value(kern, scaled_r2(A + (size_t)i * d, B + (size_t)k * d, d, ls), kvar);
      if (include_noise && i == k) v += noise;
      K[(size_t)i * n2 + k] = v;
    }
}

/* in-place lower Cholesky (upper part zeroed); 0 = ok, j+1 = pivot j not positive: whole matrix NaN (XLA semantics) */
int oc_cholesky(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0.0)) {
      for (size_t e = 0; e < (size_t)n * n; ++e) A[e] = NAN;
      return j + 1;
    }
    double ljj = sqrt(s);
    A[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double t = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = t / ljj;
    }
    for (int c = j + 1; c < n; ++c) A[(size_t)j * n + c] = 0.0;
  }
  return 0;
}

/* x <- L^-1 x (forward) / x <- L^-T x (backward), one right-hand side */
static void fwd(const double* L, int n, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}
static void bwd(const double* L, int n, double* x) {
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}

/* gp_mll (gp.py:170-178) and, when grad != NULL, its gradient wrt (log ls_1..d, log kvar) by the analytic formula
 * 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) - what jax.value_and_grad returns at optim.py:306-309.
 * L_out (n x n) and alpha_out (n) are optional.  Returns 0, or the failing pivot + 1 (outputs NaN). */
int oc_mll(int kern, const double* X, const double* y, int n, int d, const double* ls, double kvar, double noise,
           double* mll, double* grad, double* L_out, double* alpha_out) {
  double* L = (double*)malloc((size_t)n * n * sizeof(double));
  double* alpha = (double*)malloc((size_t)n * sizeof(double));
  oc_kernel(kern, X, n, X, n, d, ls, kvar, noise, 1, L);
  int info = oc_cholesky(L, n);
  if (info) {
    *mll = NAN;
    if (grad)
      for (int j = 0; j <= d; ++j) grad[j] = NAN;
    if (L_out)
      for (size_t e = 0; e < (size_t)n * n; ++e) L_out[e] = NAN;
    if (alpha_out)
      for (int i = 0; i < n; ++i) alpha_out[i] = NAN;
    free(L);
    free(alpha);
    return info;
  }
  memcpy(alpha, y, (size_t)n * sizeof(double));
  fwd(L, n, alpha);
  bwd(L, n, alpha);
  double quad = 0.0, logdet = 0.0;
  for (int i = 0; i < n; ++i) {
    quad += y[i] * alpha[i];
    logdet += log(L[(size_t)i * n + i]);
  }
  *mll = -0.5 * quad - logdet - 0.5 * n * log(2.0 * M_PI);
  if (grad) {
    /* K^-1 column by column: solve L L^T x = e_c */
    double* Kinv = (double*)malloc((size_t)n * n * sizeof(double));
    double* col = (double*)malloc((size_t)n * sizeof(double));
    for (int c = 0; c < n; ++c) {
      for (int i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
      fwd(L, n, col);
      bwd(L, n, col);
      for (int i = 0; i < n; ++i) Kinv[(size_t)i * n + c] = col[i];
    }
    for (int j = 0; j <= d; ++j) grad[j] = 0.0;
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < n; ++k) {
        const double w = alpha[i] * alpha[k] - Kinv[(size_t)i * n + k];
        const double* a = X + (size_t)i * d;
        const double* b = X + (size_t)k * d;
        const double r2 = scaled_r2(a, b, d, ls);
        const double kv = kern_value(kern, r2, kvar);
        /* d k / d log ls_j = f(r2) * D_j with D_j the squared scaled difference in dimension j */
        double f;
        if (kern == 0) {
          f = kv;
        } else if (r2 < 1e-30) {
          f = 0.0;
        } else {
          double dd = sqrt(r2);
          f = kvar * (5.0 / 3.0) * (1.0 + sqrt(5.0) * dd) * exp(-sqrt(5.0) * dd);
        }
        for (int j = 0; j < d; ++j) {
          double df = a[j] / ls[j] - b[j] / ls[j];
          grad[j] += 0.5 * w * f * df * df;
        }
        grad[d] += 0.5 * w * kv; /* d k / d log kvar = k (the noise term does not depend on kvar) */
      }
    free(Kinv);
    free(col);
  }
  if (L_out) memcpy(L_out, L, (size_t)n * n * sizeof(double));
  if (alpha_out) memcpy(alpha_out, alpha, (size_t)n * sizeof(double));
  free(L);
  free(alpha);
  return 0;
}

/* predict_single / predict_batched in standardised units (gp.py:476-493): mean = k^T alpha,
 * var = kvar + noise - |L^-1 k|^2 with NaN -> 1e-12 and < 1e-12 -> 1e-12 */
void oc_predict(int kern, const double* X, int n, int d, const double* L, const double* alpha, const double* ls,
                double kvar, double noise, const double* Xq, int c, double* mean, double* var) {
  double* k = (double*)malloc((size_t)n * sizeof(double));
  for (int q = 0; q < c; ++q) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) {
      k[i] = kern_value(kern, scaled_r2(X + (size_t)i * d, Xq + (size_t)q * d, d, ls), kvar);
      m += k[i] * alpha[i];
    }
    fwd(L, n, k);
    double vv = 0.0;
    for (int i = 0; i < n; ++i) vv += k[i] * k[i];
    double v = kvar + noise - vv;
    if (isnan(v) || v < SAFE_NOISE_FLOOR) v = SAFE_NOISE_FLOOR;
    mean[q] = m;
    var[q] = v;
  }
  free(k);
}

/* GP.fantasy_var (gp.py:552-576) in its LITERAL form: extend the factor by one row with fast_update_cholesky
 * (gp.py:181-197; a negative pivot makes the new diagonal NaN), then solve the (N+1) system for every integration
 * point.  out[m] = y_std^2 * max(var, 1e-12) with NaN -> 1e-12. */
void oc_fantasy_var(int kern, const double* X, int n, int d, const double* L, const double* ls, double kvar,
                    double noise, const double* xnew, const double* Z, int m, double y_std, double* out) {
  const int n1 = n + 1;
  double* L1 = (double*)calloc((size_t)n1 * n1, sizeof(double));
  double* v = (double*)malloc((size_t)n1 * sizeof(double));
  for (int i = 0; i < n; ++i) {
    memcpy(L1 + (size_t)i * n1, L + (size_t)i * n, (size_t)(i + 1) * sizeof(double));
    v[i] = kern_value(kern, scaled_r2(X + (size_t)i * d, xnew, d, ls), kvar);
  }
  fwd(L, n, v);
  double vv = 0.0;
  for (int i = 0; i < n; ++i) {
    L1[(size_t)n * n1 + i] = v[i];
    vv += v[i] * v[i];
  }
  L1[(size_t)n * n1 + n] = sqrt(kvar + noise - vv); /* NaN when the pivot is negative, like jnp.sqrt */
  for (int z = 0; z < m; ++z) {
    const double* zp = Z + (size_t)z * d;
    for (int i = 0; i < n; ++i) v[i] = kern_value(kern, scaled_r2(X + (size_t)i * d, zp, d, ls), kvar);
    v[n] = kern_value(kern, scaled_r2(xnew, zp, d, ls), kvar);
    /* forward substitution with the extended factor (NaN propagates from the last row) */
    double q = 0.0;
    for (int i = 0; i < n1; ++i) {
      double s = v[i];
      for (int k = 0; k < i; ++k) s -= L1[(size_t)i * n1 + k] * v[k];
      v[i] = s / L1[(size_t)i * n1 + i];
      q += v[i] * v[i];
    }
    double var = kvar + noise - q;
    if (isnan(var) || var < SAFE_NOISE_FLOOR) var = SAFE_NOISE_FLOOR;
    out[z] = var * y_std * y_std;
  }
  free(L1);
  free(v);
}
