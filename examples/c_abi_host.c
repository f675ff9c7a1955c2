/* A host program in plain C over include/bobe_gp.h — the drop-in boundary without Python or torch.
 *
 *   gcc -std=c99 -O2 -I include examples/c_abi_host.c -o c_abi_host -L bobe_amd -lbobe_gp -Wl,-rpath,$PWD/bobe_amd -lm
 *
 * It does what a maintainer's binding would do for one BO iteration (reference call sites in brackets):
 *   GP.__init__ + recompute_cholesky   [gp.py:201-281, 544-550]   bobe_gp_create / set_data / set_hyper / factor
 *   value + gradient of the MLL        [gp.py:170-178 under optim.py:306-309]   bobe_gp_mll
 *   posterior mean / variance          [gp.py:476-493]            bobe_gp_predict
 *   WIPV / WIPStd sweep + argmin       [acquisition.py:385-398, 438-465]   bobe_gp_wip_sweep
 *   GPwithClassifier's gate            [clf_gp.py:173-205, clf.py:188-213]  bobe_gp_set_gate / bobe_gp_gate_eval
 *   nested sampling's random walks     [samplers.py:112-115, 152]   bobe_gp_rwalk
 * and checks the results against closed forms that need no oracle: the gradient against central differences of the
 * value, the posterior at training points (interpolation), the returned argmin against the returned scores, the gate's
 * decision values against the sum written out in C, the walkers' end points against the constraints.
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
  /* a classifier gate from made-up SVM parameters: decision(x) = sum_i dual_i exp(-gamma |sv_i - x|^2) + b (clf.py:188-208) */
  enum { NSV = 20, P = 32 };
  static double sv[NSV * D], dual[NSV], dec[64], feas[64], wx[P * D], wl[P], step[D * D];
  static int nacc[P], nin[P];
  const double gam = 3.0, b0 = -0.2, minus_inf = -1e5;
  for (int i = 0; i < NSV * D; ++i) sv[i] = lcg(&seed);
  for (int i = 0; i < NSV; ++i) dual[i] = (i % 3 == 0 ? -1.0 : 1.0) * (0.5 + lcg(&seed));
  CHECK(bobe_gp_set_gate(gp, sv, NSV, dual, b0, gam, 0.5, minus_inf));
  CHECK(bobe_gp_gate_eval(gp, cand, 64, dec, feas));
  CHECK(bobe_gp_predict(gp, cand, 64, mean, var, 1));
  int ngated = 0;
  for (int c = 0; c < 64; ++c) {
    double want = 0.0;
    for (int i = 0; i < NSV; ++i) {
      double r2 = 0.0;
      for (int j = 0; j < D; ++j) r2 += (sv[i * D + j] - cand[c * D + j]) * (sv[i * D + j] - cand[c * D + j]);
      want += dual[i] * exp(-gam * r2);
    }
    want += b0;
    if (fabs(dec[c] - want) > 1e-12 * (1.0 + fabs(want)) || feas[c] != (dec[c] >= 0.0 ? 1.0 : 0.0)) bad = 1;
    if (feas[c] == 0.0) { /* gated: the mark for minus_inf, and the noise floor (clf_gp.py:203-204) */
      ++ngated;
      if (!(isinf(mean[c]) && mean[c] < 0.0) || var[c] != 1e-12) bad = 1;
    } else if (!isfinite(mean[c])) {
      bad = 1;
    }
  }
  /* constrained random walks above L* = the 25 % quantile of the feasible starts' means, 20 steps each */
  int nstart = 0;
  for (int c = 0; c < C && nstart < P; ++c) {
    double d1, f1, m1;
    CHECK(bobe_gp_gate_eval(gp, cand + c * D, 1, &d1, &f1));
    if (f1 == 0.0) continue;
    CHECK(bobe_gp_predict(gp, cand + c * D, 1, &m1, NULL, 1));
    for (int j = 0; j < D; ++j) wx[nstart * D + j] = cand[c * D + j];
    wl[nstart++] = m1 * ys + ym;
  }
  double lstar = wl[0];
  for (int i = 1; i < nstart; ++i) lstar = fmin(lstar, wl[i]);
  lstar -= 0.5;
  for (int i = 0; i < D * D; ++i) step[i] = 0.0;
  for (int j = 0; j < D; ++j) step[j * D + j] = 0.1;
  if (nstart == P) {
    CHECK(bobe_gp_rwalk(gp, P, wx, wl, step, lstar, 20, 4242ULL, ys, ym, nacc, nin, NULL));
    int moved = 0;
    for (int w = 0; w < P; ++w) {
      moved += nacc[w] > 0;
      if (nacc[w] > nin[w] || nin[w] > 20 || !(wl[w] > lstar)) bad = 1;
      for (int j = 0; j < D; ++j)
        if (wx[w * D + j] < 0.0 || wx[w * D + j] > 1.0) bad = 1;
    }
    CHECK(bobe_gp_gate_eval(gp, wx, P, dec, feas));
    for (int w = 0; w < P; ++w)
      if (feas[w] == 0.0) bad = 1; /* no walker ends in the infeasible region */
    if (moved == 0) bad = 1;
  } else {
    bad = 1;
  }
  CHECK(bobe_gp_set_gate(gp, NULL, 0, NULL, 0.0, 0.0, 0.5, minus_inf)); /* NULL clears */
  printf("gate: %d of 64 candidates infeasible\n", ngated);
  printf("%s: N=%d d=%d  MLL=%.6f  |grad|_inf=%.4g  argmin WIPV=%lld WIPStd=%lld  %s\n", bobe_version(), N, D, mll,
         fmax(fmax(fabs(grad[0]), fabs(grad[1])), fmax(fabs(grad[2]), fabs(grad[3]))), (long long)av, (long long)as,
         bad ? "CHECKS FAILED" : "all checks passed");
  bobe_gp_destroy(gp);
  return bad;
}
