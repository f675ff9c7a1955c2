/* Extended-precision "truth" for the GP hot path (TEST INFRASTRUCTURE - not product code).
 *
 * The same quantities as oracle/bobe_oracle.py and the HIP path compute in fp64 - log marginal likelihood and its
 * gradient (gp.py:170-178, optim.py:306-309), posterior mean / variance (gp.py:450-474), the fantasy variance
 * var+(z | c) behind WIPV / WIPStd (gp.py:552-576, acquisition.py:438-465) - carried out in a WIDER floating-point type
 * from the fp64 inputs: x87 `long double` (64-bit significand, unit roundoff 5.4e-20: 2048 times finer than fp64) by
 * default, `__float128` (113-bit significand) with -DXP_QUAD.  At the reference's default noise of 1e-8 and the
 * hyper-parameters its fits reach (kernel variance up to 1e6: cond K up to ~1e14 and beyond) two fp64 implementations
 * can differ by more than any fixed tolerance while both being as good as fp64 allows; this file gives the reference
 * point both are measured against (tests/test_gpu_conditioning.py).  Plain triple loops, no BLAS; OpenMP over rows.
 * Only tests/ and __graft_entry__.build() touch this file.  All arrays are row-major; inputs and outputs are fp64.
 *
 *   gcc -O2 -fopenmp -shared -fPIC -o oracle/libbobe_oracle_xp.so oracle/bobe_oracle_xp.c -lm             (long double)
 *   gcc -O2 -fopenmp -DXP_QUAD -shared -fPIC -o oracle/libbobe_oracle_xq.so oracle/bobe_oracle_xp.c -lquadmath -lm
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef XP_QUAD
#include <quadmath.h>
typedef __float128 xr;
#define XSQRT sqrtq
#define XEXP expq
#define XLOG logq
#define XC(v) v##Q
#else
typedef long double xr;
#define XSQRT sqrtl
#define XEXP expl
#define XLOG logl
#define XC(v) v##L
#endif

/* significand bits of the working type (tests print it) */
int xp_digits(void) {
#ifdef XP_QUAD
  return 113;
#else
  return LDBL_MANT_DIG;
#endif
}

/* gp.py:124-168 on the scaled squared distance */
static xr kern_value(int kern, xr r2, xr kvar) {
  if (kern == 0) return kvar * XEXP(-XC(0.5) * r2);
  xr dd = XSQRT(r2 < XC(1e-30) ? XC(1e-30) : r2);
  xr s5 = XSQRT(XC(5.0));
  return kvar * (XC(1.0) + dd * (s5 + dd * XC(5.0) / XC(3.0))) * XEXP(-s5 * dd);
}
/* d k / d log ls_j = gfac * D_j  (D_j = squared scaled difference in dimension j) */
static xr kern_gfac(int kern, xr r2, xr kvar, xr kval) {
  if (kern == 0) return kval;
  if (r2 < XC(1e-30)) return XC(0.0);
  xr s5 = XSQRT(XC(5.0)), dd = XSQRT(r2);
  return kvar * (XC(5.0) / XC(3.0)) * (XC(1.0) + s5 * dd) * XEXP(-s5 * dd);
}
static xr scaled_r2(const double* a, const double* b, int d, const double* ls) {
  xr r2 = 0;
  for (int j = 0; j < d; ++j) {
    xr df = (xr)a[j] / (xr)ls[j] - (xr)b[j] / (xr)ls[j];      /* dist_sq(xa / ls, xb / ls), gp.py:149 */
    r2 += df * df;
  }
  return r2;
}

/* in-place lower Cholesky; returns 0 or 1 + the column of the first non-positive pivot */
static int chol(xr* A, int n) {
  for (int j = 0; j < n; ++j) {
    xr s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0)) return j + 1;
    xr ljj = XSQRT(s);
    A[(size_t)j * n + j] = ljj;
#pragma omp parallel for schedule(static)
    for (int i = j + 1; i < n; ++i) {
      xr t = A[(size_t)i * n + j];
      const xr* ri = A + (size_t)i * n;
      const xr* rj = A + (size_t)j * n;
      for (int k = 0; k < j; ++k) t -= ri[k] * rj[k];
      A[(size_t)i * n + j] = t / ljj;
    }
  }
  return 0;
}
static void fwd(const xr* L, int n, xr* x) { /* L x = b */
  for (int i = 0; i < n; ++i) {
    xr s = x[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}
static void bwd(const xr* L, int n, xr* x) { /* L^T x = b */
  for (int i = n - 1; i >= 0; --i) {
    xr s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}

/* Everything at once for one (data, hyper-parameter) pair.  y: standardised targets.  Any output may be NULL.
 *   mll            data term of the log marginal likelihood (gp.py:170-178)
 *   grad[d+1]      d mll / d log ls_j (j < d), d mll / d log kvar
 *   mean[c], var[c]   posterior at Xq in standardised units, noise included, NO floors (gp.py:459-466)
 *   var_z[m]          the same variance at Z
 *   fantasy[c*m]      var+(z | c) in standardised units, NO floors (gp.py:552-576)
 *   cross[c*m]        the cross term k(x_c, z) - v_c . v_z behind it
 *   min_pivot         smallest L_jj^2 of the factorisation
 * returns 0, or 1 + the column of the first non-positive pivot (nothing else is written then). */
int xp_gp_truth(int kern, const double* X, const double* y, int n, int d, const double* ls, double kvar, double noise,
                const double* Xq, int c, const double* Z, int m, double* mll, double* grad, double* mean, double* var,
                double* var_z, double* fantasy, double* min_pivot, double* cross) {
  const size_t nn = (size_t)n * n;
  xr* L = (xr*)malloc(nn * sizeof(xr));
  xr* Kt = grad ? (xr*)malloc(nn * sizeof(xr)) : NULL;       /* the kernel matrix without the noise term */
  if (!L || (grad && !Kt)) {
    free(L);
    free(Kt);
    return -1;
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) {
      xr v = kern_value(kern, scaled_r2(X + (size_t)i * d, X + (size_t)k * d, d, ls), (xr)kvar);
      if (Kt) Kt[(size_t)i * n + k] = v;
      L[(size_t)i * n + k] = (i == k) ? v + (xr)noise : v;
    }
  int info = chol(L, n);
  if (info) {
    free(L);
    free(Kt);
    return info;
  }
  if (min_pivot) {
    xr mp = L[0] * L[0];
    for (int i = 1; i < n; ++i) {
      xr p = L[(size_t)i * n + i] * L[(size_t)i * n + i];
      if (p < mp) mp = p;
    }
    *min_pivot = (double)mp;
  }
  xr* alpha = (xr*)malloc((size_t)n * sizeof(xr));
  for (int i = 0; i < n; ++i) alpha[i] = (xr)y[i];
  fwd(L, n, alpha);
  xr quad = 0, logdet = 0;
  for (int i = 0; i < n; ++i) {
    quad += alpha[i] * alpha[i];                    /* y^T K^-1 y = |L^-1 y|^2 */
    logdet += XLOG(L[(size_t)i * n + i]);
  }
  bwd(L, n, alpha);
  if (mll) *mll = (double)(-XC(0.5) * quad - logdet - XC(0.5) * (xr)n * XLOG(XC(2.0) * XC(3.14159265358979323846264338327950288)));
  if (grad) {
    /* K^-1 = Linv^T Linv, Linv column by column (independent forward solves) */
    xr* Li = (xr*)calloc(nn, sizeof(xr));           /* Li[col][row]: column `col` of L^-1, contiguous */
#pragma omp parallel for schedule(dynamic, 8)
    for (int col = 0; col < n; ++col) {
      xr* x = Li + (size_t)col * n;
      x[col] = XC(1.0) / L[(size_t)col * n + col];
      for (int i = col + 1; i < n; ++i) {
        xr s = 0;
        const xr* ri = L + (size_t)i * n;
        for (int k = col; k < i; ++k) s -= ri[k] * x[k];
        x[i] = s / ri[i];
      }
    }
    /* W = alpha alpha^T - K^-1;  K^-1[a][b] = sum_{r >= max(a,b)} Linv[r][a] Linv[r][b] = sum_r Li[a][r] Li[b][r] */
    xr* g = (xr*)calloc((size_t)(d + 1) * n, sizeof(xr));   /* per-row partial sums (deterministic reduction) */
#pragma omp parallel for schedule(dynamic, 4)
    for (int a = 0; a < n; ++a) {
      const xr* la = Li + (size_t)a * n;
      for (int b = 0; b < n; ++b) {
        const xr* lb = Li + (size_t)b * n;
        xr kinv = 0;
        for (int r = (a > b ? a : b); r < n; ++r) kinv += la[r] * lb[r];
        xr w = alpha[a] * alpha[b] - kinv;
        xr r2 = scaled_r2(X + (size_t)a * d, X + (size_t)b * d, d, ls);
        xr kv = Kt[(size_t)a * n + b];
        xr gf = kern_gfac(kern, r2, (xr)kvar, kv);
        for (int j = 0; j < d; ++j) {
          xr df = (xr)X[(size_t)a * d + j] / (xr)ls[j] - (xr)X[(size_t)b * d + j] / (xr)ls[j];
          g[(size_t)j * n + a] += w * gf * df * df;
        }
        g[(size_t)d * n + a] += w * kv;
      }
    }
    for (int j = 0; j <= d; ++j) {
      xr s = 0;
      for (int a = 0; a < n; ++a) s += g[(size_t)j * n + a];
      grad[j] = (double)(XC(0.5) * s);
    }
    free(g);
    free(Li);
  }
  /* v = L^-1 k(X, q) for the query points and the integration points */
  const int tot = c + m;
  xr* V = tot > 0 ? (xr*)malloc((size_t)tot * n * sizeof(xr)) : NULL;       /* V[q][row] */
  xr* vq = tot > 0 ? (xr*)malloc((size_t)tot * sizeof(xr)) : NULL;          /* kvar + noise - |v|^2 */
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < tot; ++q) {
    const double* p = q < c ? Xq + (size_t)q * d : Z + (size_t)(q - c) * d;
    xr* v = V + (size_t)q * n;
    xr mu = 0;
    for (int i = 0; i < n; ++i) {
      v[i] = kern_value(kern, scaled_r2(X + (size_t)i * d, p, d, ls), (xr)kvar);
      mu += v[i] * alpha[i];
    }
    fwd(L, n, v);
    xr s = 0;
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    vq[q] = (xr)kvar + (xr)noise - s;
    if (q < c) {
      if (mean) mean[q] = (double)mu;
      if (var) var[q] = (double)vq[q];
    } else if (var_z) {
      var_z[q - c] = (double)vq[q];
    }
  }
  if (fantasy) {
#pragma omp parallel for schedule(static)
    for (int a = 0; a < c; ++a)
      for (int z = 0; z < m; ++z) {
        const xr* va = V + (size_t)a * n;
        const xr* vz = V + (size_t)(c + z) * n;
        xr dot = 0;
        for (int i = 0; i < n; ++i) dot += va[i] * vz[i];
        xr cov = kern_value(kern, scaled_r2(Xq + (size_t)a * d, Z + (size_t)z * d, d, ls), (xr)kvar) - dot;
        fantasy[(size_t)a * m + z] = (double)(vq[c + z] - cov * cov / vq[a]);
        if (cross) cross[(size_t)a * m + z] = (double)cov;
      }
  }
  free(V);
  free(vq);
  free(alpha);
  free(L);
  free(Kt);
  return 0;
}
