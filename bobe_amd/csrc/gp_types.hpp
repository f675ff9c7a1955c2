// Plain types shared by the kernels and the host side of libbobe_gp.so.
#pragma once
#include <stdint.h>

namespace bobe {

constexpr int MAX_D = 32;

struct Hyper {
  double ls[MAX_D];
  double kvar;
  double noise;
  int d;
  int kern;  // 0 rbf, 1 matern-5/2
};

// problem {lo, mid, hi} of the recursive triangular inverse in 128-block units: with inv[lo:mid) and inv[mid:hi) known,
//   Tm = L[mid:hi, lo:mid) * inv[lo:mid)  (k_trtri_T),   inv[mid:hi, lo:mid) = -inv[mid:hi) * Tm  (k_trtri_R).
// `off` counts T x T tiles: a problem owns (hi-mid)(mid-lo)(128/T)^2 consecutive blocks.
struct TriProb { int lo, mid, hi, off; };

// A filler job of a panel launch (k_chol_panel<true, .>): update tile A[ti][tj] -= sum_k A[ti][k] A[tj][k]^T over the
// 64-column units [k0, k1) - k_syrk_trail's tile.  The two jobs of a workgroup run the same number of K-steps
// (workgroup-wide barriers): the plan pairs tiles of one block column; FILL_TWIN completes an odd count (computed, not stored).
enum { FILL_TWIN = 1 };
struct FillJob { int ti, tj, k0, k1, flags, pad0; };

// The classifier gate of GPwithClassifier (clf_gp.py:173-205) with the SVM-RBF decision function of clf.py:188-213:
//   decision(x) = sum_i dual[i] exp(-gamma |sv_i - x|^2) + intercept,  proba = decision >= 0,  feasible = proba >= threshold.
// svT: support vectors SoA, coordinate j of vector i at svT[j * ld + i] (unit-cube coordinates, not scaled).  n_sv = 0: no gate.
struct Gate {
  const double* svT;
  const double* dual;
  int64_t ld;
  int n_sv;
  double intercept, gamma, threshold, minus_inf;
};

}  // namespace bobe
