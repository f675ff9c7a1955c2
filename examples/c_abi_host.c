/* A host program in plain C over include/bobe_gp.h — the drop-in boundary without Python or torch.
 *
 *   gcc -std=c99 -O2 -I include examples/c_abi_host.c -o c_abi_host -L bobe_amd -lbobe_gp -Wl,-rpath,$PWD/bobe_amd -lm
 *
 * It does what a maintainer's binding would do for one BO iteration (reference call sites in brackets):
 *   GP.__init__ + recompute_cholesky   [gp.py:201-281, 544-550]   bobe_gp_create / set_data / set_hyper / factor
 *   value + gradient of the MLL        [gp.py:170-178 under optim.py:306-309]   bobe_gp_mll
 *   posterior mean / variance          [gp.py:476-493]            bobe_gp_predict
 *   WIPV / WIPStd sweep + argmin       [acquisition.py:385-398, 438-465]   bobe_gp_wip_sweep
 * and checks the results against closed forms that need no oracle: the gradient against central differences of the
 * value, the posterior at training points (interpolation), the returned argmin against the returned scores.
 * Exit code 0 = all checks passed; 77 = no HIP device (nothing was computed: the library has no CPU path). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "bobe_gp.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int st_ = (call);                                                                   \
    if (st_ < 0) {                                                                      \
      fprintf(stderr, "%s failed (%d): %s\n", #call, st_, bobe_last_error());           \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

static double lcg(unsigned long long* s) { /* uniform in (0,1), seeded */
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((double)(*s >> 11) + 0.5) / 9007199254740992.0;
}

int main(void) {
  enum { N = 300, D = 3, C = 1000, M = 64 };
  if (bobe_device_count() < 1) {
    fprintf(stderr, "no HIP device: %s has no CPU compute path\n", bobe_version());
    return 77;
  }
  static double X[N * D], y[N], cand[C * D], Z[M * D], mean[C], var[C], wipv[C], wipstd[C];
  unsigned long long seed = 12345;
  double ym = 0.0, ys = 0.0;
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < D; ++j) X[i * D + j] = lcg(&seed);
    y[i] = sin(3.0 * X[i * D]) + X[i * D + 1] * X[i * D + 1] - X[i * D + 2];
    ym += y[i];
  }
  ym /= N;
  for (int i = 0; i < N; ++i) ys += (y[i] - ym) * (y[i] - ym);
  ys = sqrt(ys / N);
  for (int i = 0; i < N; ++i) y[i] = (y[i] - ym) / ys; /* the wrapper's standardisation, gp.py:283-307 */
  for (int i = 0; i < C * D; ++i) cand[i] = lcg(&seed);
  for (int i = 0; i < M * D; ++i) Z[i] = lcg(&seed);

  bobe_gp_t* gp = NULL;
  CHECK(bobe_gp_create(&gp, 0, BOBE_KERNEL_RBF, D));
  CHECK(bobe_gp_set_data(gp, X, y, N));
  double ls[D] = {0.4, 0.5, 0.6};
  const double kvar = 1.3, noise = 1e-6;
  CHECK(bobe_gp_set_hyper(gp, ls, kvar, noise));
  CHECK(bobe_gp_factor(gp));

  /* value + gradient wrt (log ls, log kvar); central differences of the value as the check */
  double mll = 0.0, grad[D + 1];
  CHECK(bobe_gp_mll(gp, ls, kvar, &mll, grad));
  int bad = 0;
  for (int j = 0; j <= D; ++j) {
    double lp[D], lm[D], kp = kvar, km = kvar, fp, fm;
    const double e = 1e-5;
    for (int q = 0; q < D; ++q) lp[q] = lm[q] = ls[q];
    if (j < D) {
      lp[j] = ls[j] * exp(e);
      lm[j] = ls[j] * exp(-e);
    } else {
      kp = kvar * exp(e);
      km = kvar * exp(-e);
    }
    CHECK(bobe_gp_mll(gp, lp, kp, &fp, NULL));
    CHECK(bobe_gp_mll(gp, lm, km, &fm, NULL));
    const double fd = (fp - fm) / (2.0 * e);
    if (fabs(fd - grad[j]) > 1e-5 * fmax(1.0, fabs(grad[j]))) {
      fprintf(stderr, "gradient component %d: analytic %.10g, central difference %.10g\n", j, grad[j], fd);
      bad = 1;
    }
  }
  /* posterior: interpolation at training points, variance between the floor and the prior variance */
  CHECK(bobe_gp_predict(gp, X, 64, mean, var, 1));
  for (int i = 0; i < 64; ++i)
    if (fabs(mean[i] - y[i]) > 1e-3 || !(var[i] >= 1e-12) || var[i] > kvar + noise) bad = 1;
  /* sweep: scores, mean, variance of all candidates and both argmins in one call */
  int64_t av = -1, as = -1;
  double mv = 0.0, ms = 0.0;
  CHECK(bobe_gp_wip_sweep(gp, cand, C, Z, M, ys, wipv, wipstd, mean, var, &av, &mv, &as, &ms));
  int64_t bv = 0, bs = 0;
  for (int i = 1; i < C; ++i) {
    if (wipv[i] < wipv[bv]) bv = i;
    if (wipstd[i] < wipstd[bs]) bs = i;
  }
  if (bv != av || bs != as || mv != wipv[av] || ms != wipstd[as]) bad = 1;
  for (int i = 0; i < C; ++i)
    if (!(wipv[i] > 0.0) || wipstd[i] * wipstd[i] > wipv[i] * (1.0 + 1e-12)) bad = 1; /* Jensen */
  printf("%s: N=%d d=%d  MLL=%.6f  |grad|_inf=%.4g  argmin WIPV=%lld WIPStd=%lld  %s\n", bobe_version(), N, D, mll,
         fmax(fmax(fabs(grad[0]), fabs(grad[1])), fmax(fabs(grad[2]), fabs(grad[3]))), (long long)av, (long long)as,
         bad ? "CHECKS FAILED" : "all checks passed");
  bobe_gp_destroy(gp);
  return bad;
}
