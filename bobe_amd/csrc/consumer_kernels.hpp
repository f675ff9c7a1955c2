// Kernels of the surrogate's consumers: HMC on the posterior mean, EI / LogEI, the classifier gate (gfx950).
// Included by gp_consumers.hip only.
#pragma once
#include "kernels_common.hpp"

namespace bobe {

// EI / LogEI pointwise scorers (BOBE/acquisition.py:21-75, 226-253, 318-330); mu, var standardised
__device__ __forceinline__ double norm_pdf(double u) { return exp(-0.5 * u * u) * 0.39894228040143267794; }
__device__ __forceinline__ double norm_cdf(double u) { return 0.5 * erfc(-u * 0.70710678118654752440); }
__device__ __forceinline__ double ei_helper(double u) { return norm_pdf(u) + u * norm_cdf(u); }
__device__ __forceinline__ double log1mexp_tfp(double x) {
  x = fabs(x);
  return (x < 0.69314718055994530942) ? log(-expm1(-x)) : log1p(-exp(-x));
}
__device__ __forceinline__ double log_ei_helper(double u) {
  const double bound = -1.0, neg_inv_sqrt_eps = -1e6;
  if (u > bound) return log(ei_helper(u));
  const double u_lower = u;
  const double u_eps = (u_lower < neg_inv_sqrt_eps) ? neg_inv_sqrt_eps : u_lower;
  const double w = log(fabs(u_eps) * erfcx(-0.70710678118654752440 * u_eps)) + 0.22579135264472743236;
  const double log_phi_u = -0.5 * (u * u + 1.83787706640934548356);
  const double second = (u > neg_inv_sqrt_eps) ? log1mexp_tfp(w) : -2.0 * log(fabs(u_lower));
  return log_phi_u + second;
}

// ---- the classifier gate of GPwithClassifier (clf_gp.py:173-205) ---------------------------------------------------
// SVM-RBF decision function of clf.py:188-213 by DIRECT DIFFERENCES, as the reference computes it:
//   diff = support_vectors - x;  norm_sq = sum_j diff_j^2;  decision = sum_i dual_i exp(-gamma norm_sq_i) + intercept.
// One 256-thread workgroup per point, ONE summation order everywhere (the batch kernel and the HMC kernels share
// gate_partial / gate_combine): thread t adds the vectors t, t + 256, ... in ascending order, lanes by chain_wave_sum, then the
// four waves ((r0 + r1) + r2) + r3, then + intercept.  A point near the boundary is therefore classified the same way by
// every entry point.  x: the point's d raw (unit-cube) coordinates, readable by every thread.
template <int DCAP>
__device__ __forceinline__ double gate_partial(const Gate& gt, const double* x, int d, int t) {
  double xr[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) xr[j] = (j < d) ? x[j] : 0.0;
  double s = 0.0;
  for (int i = t; i < gt.n_sv; i += 256) {
    double r2 = 0.0;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) {
      if (j < d) {
        const double df = gt.svT[j * gt.ld + i] - xr[j];
        r2 += df * df;
      }
    }
    s += gt.dual[i] * exp(-gt.gamma * r2);
  }
  return chain_wave_sum(s);
}
// red[4]: the four waves' sums (written by lane 0 of each wave, followed by a barrier)
__device__ __forceinline__ double gate_combine(const Gate& gt, const double* red) {
  return (((red[0] + red[1]) + red[2]) + red[3]) + gt.intercept;
}
// svm_predict_proba (clf.py:210-213) against the probability threshold (clf_gp.py:179): NaN decisions are infeasible
__device__ __forceinline__ bool gate_feasible(const Gate& gt, double decision) {
  const double proba = (decision >= 0.0) ? 1.0 : 0.0;
  return proba >= gt.threshold;
}

// decision / feasibility of C points (xq: C x d row-major, raw coordinates); optionally the gating of a prediction in
// place: mean -> -inf (the wrapper's sentinel for minus_inf, bobe_gp.h), var -> 1e-12 (clf_gp.py:189, 203), gradients -> 0
template <int DCAP>
__global__ __launch_bounds__(256) void k_gate(Gate gt, const double* __restrict__ xq, int d, double* __restrict__ decision,
                                              double* __restrict__ feasible, double* __restrict__ mean,
                                              double* __restrict__ var, double* __restrict__ dmean,
                                              double* __restrict__ dvar) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  const int64_t c = blockIdx.x;
  const double s = gate_partial<DCAP>(gt, xq + c * d, d, t);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const double dec = gate_combine(gt, red);
  const bool ok = gate_feasible(gt, dec);
  if (t == 0) {
    if (decision) decision[c] = dec;
    if (feasible) feasible[c] = ok ? 1.0 : 0.0;
    if (!ok) {
      if (mean) mean[c] = -INFINITY;
      if (var) var[c] = NOISE_FLOOR;
    }
  }
  if (!ok && t < d) {
    if (dmean) dmean[c * d + t] = 0.0;
    if (dvar) dvar[c * d + t] = 0.0;
  }
}

// ---- Hamiltonian Monte Carlo on the surrogate: L leapfrog steps of every chain in ONE launch ---------------------
// Consumer of the posterior mean (the reference's NUTS differentiates predict_mean_batched through JAX, one call per
// step and chain, samplers.py:268-288).  One workgroup = one chain.  Target on u = logit(x), x in the unit cube:
//   logp(u) = (mean(x) * ystd + ymean) / temp + sum_j [log x_j + log(1 - x_j)]          (Jacobian of the logit map)
//   g(u)    = dmean/dx * ystd / temp * x (1 - x) + (1 - 2x)
// with mean(x) = sum_n alpha_n k(x_n, x), dmean/dx_j = sum_n alpha_n G(r2) (s_nj - s_j) / ls_j (k_predict_grad's
// mean-only arithmetic).  In:  U, Pm = p0 + eps/2 * g(U)  (P x d).  Per step: u += eps * inv_mass * p; evaluate;
// p += (eps | eps/2 on the last step) * g.  Out: U, Pm (final), logp, grad, mean (physical units), X.
// Fixed reduction order (4 waves x lanes, then a fixed tree): a chain's trajectory does not depend on the batch.
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_hmc_leapfrog(const double* __restrict__ XsT, int64_t ldx, int64_t n,
                                                      const double* __restrict__ alpha, Hyper h,
                                                      double* __restrict__ U, double* __restrict__ Pm,
                                                      const double* __restrict__ inv_mass, double eps, int L,
                                                      double ystd, double ymean, double temp,
                                                      double* __restrict__ logp, double* __restrict__ grad,
                                                      double* __restrict__ mean_out, double* __restrict__ Xout, Gate gt) {
  __shared__ double u[DCAP], pm[DCAP], x[DCAP], xs[DCAP], g[DCAP], red[4][DCAP + 1], lp_s, mean_s, gred[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.x;
  const int d = h.d;
  if (t < d) {
    u[t] = U[c * d + t];
    pm[t] = Pm[c * d + t];
  }
  __syncthreads();
  for (int s = 0; s < L; ++s) {
    if (t < d) {
      const double un = u[t] + eps * inv_mass[t] * pm[t];
      u[t] = un;
      double xv = 1.0 / (1.0 + exp(-un));
      xv = xv < 1e-12 ? 1e-12 : (xv > 1.0 - 1e-12 ? 1.0 - 1e-12 : xv);
      x[t] = xv;
      xs[t] = xv / h.ls[t];
    }
    __syncthreads();
    double ms = 0.0, gm[DCAP];
#pragma unroll
    for (int j = 0; j < DCAP; ++j) gm[j] = 0.0;
    for (int64_t i = t; i < n; i += 256) {
      double df[DCAP];
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DCAP; ++j) {
        df[j] = (j < d) ? XsT[j * ldx + i] - xs[j] : 0.0;
        r2 += df[j] * df[j];
      }
      const double kv = kern_eval<KERN>(r2, h.kvar);
      const double a = alpha[i];
      const double ag = a * kern_grad_factor<KERN>(r2, h.kvar, kv);
      ms += a * kv;
#pragma unroll
      for (int j = 0; j < DCAP; ++j) gm[j] += ag * df[j];
    }
    ms = wave_sum(ms);
    if (lane == 0) red[wave][DCAP] = ms;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) {
      if (j < d) {
        const double v = wave_sum(gm[j]);
        if (lane == 0) red[wave][j] = v;
      }
    }
    // classifier gate (clf_gp.py:173-205): an infeasible point has mean = minus_inf and no mean gradient
    if (gt.n_sv > 0) {
      const double gs = gate_partial<DCAP>(gt, x, d, t);
      if (lane == 0) gred[wave] = gs;
    }
    __syncthreads();
    const bool ok = gt.n_sv > 0 ? gate_feasible(gt, gate_combine(gt, gred)) : true;
    if (t < d) {
      const double dm = ok ? (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]) / h.ls[t] : 0.0;
      const double xv = x[t];
      const double gv = dm * ystd / temp * (xv * (1.0 - xv)) + (1.0 - 2.0 * xv);
      g[t] = gv;
      pm[t] += ((s < L - 1) ? eps : 0.5 * eps) * gv;
    }
    if (t == 0) {
      const double m = ok ? (((red[0][DCAP] + red[1][DCAP]) + red[2][DCAP]) + red[3][DCAP]) * ystd + ymean : gt.minus_inf;
      double jac = 0.0;
      for (int j = 0; j < d; ++j) jac += log(x[j]) + log1p(-x[j]);
      mean_s = m;
      lp_s = m / temp + jac;
    }
    __syncthreads();
  }
  if (t < d) {
    U[c * d + t] = u[t];
    Pm[c * d + t] = pm[t];
    grad[c * d + t] = g[t];
    Xout[c * d + t] = x[t];
  }
  if (t == 0) {
    logp[c] = lp_s;
    mean_out[c] = mean_s;
  }
}

// ---- whole HMC chains on the device ---------------------------------------------------------------------------
// `niter` trajectories of every chain in ONE launch (one workgroup = one chain): momentum draw, 4-12 leapfrog steps (the
// arithmetic of k_hmc_leapfrog), Metropolis test, and - while warming up - the chain's own dual-averaging step-size
// update (Hoffman & Gelman 2014, what NumPyro's warm-up does per chain).  The host only cuts the run at the
// mass-matrix windows.  Random numbers are a counter hash (splitmix64 finaliser) of (seed, chain, iteration, index):
// a chain's path depends on nothing but its own seed, whatever the batch or the launch boundaries.
//   S     [P][3d+2]  chain state in/out: u (d), g = dlogp/du (d), x (d), logp, mean (physical units)
//   adapt [P][5]     eps, mu, hbar, log_eps_bar, m (dual averaging; only eps is read when do_adapt == 0)
//   hist  [niter - hist_from][P][d]   u after iterations >= hist_from of this launch        (may be null)
//   keep  [niter / thin][P][d+1]      x and mean after every thin-th iteration of this launch (may be null)
//   dbg   [P][d+3]   last iteration's p0 (d), L, uniform, acceptance probability             (may be null)
__device__ __forceinline__ unsigned long long hmc_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double hmc_u01(unsigned long long bits) {          // in (0, 1)
  return ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// Training points a thread of a 256-thread chain workgroup keeps in registers for the whole launch (a launch is hundreds
// of leapfrog / random-walk steps over the same points): 256 x RESIDENT rows, the rest is streamed from L2 every step.
// (One wave per SIMD: the 512 unified registers of a lane hold them; chosen as the largest counts without scratch spills.)
template <int DCAP>
struct ChainRows {
  static constexpr int HMC = DCAP == 8 ? 14 : (DCAP == 16 ? 6 : 2);
  static constexpr int WALK = DCAP == 8 ? 16 : (DCAP == 16 ? 8 : 4);
  static constexpr int STREAM = DCAP == 32 ? 1 : 2;      // streamed rows in flight per thread
};
// ... and as many more as the CU's LDS holds next to them: groups of 256 rows of d coordinates + alpha, [d + 1][rows]
// (a lane reads consecutive doubles: no bank conflicts).  Host side: the group count for n points and `resident` register rows.
constexpr int CHAIN_LDS_BYTES = 148 * 1024;                            // (of 160 KB: the kernels' static arrays stay below 10 KB)
inline int chain_lds_groups(int64_t n, int d, int resident) {
  const int64_t left = n - 256 * (int64_t)resident;
  if (left <= 0) return 0;
  const int64_t cap = CHAIN_LDS_BYTES / (8 * (int64_t)(d + 1)) / 256;
  const int64_t want = (left + 255) / 256;
  return (int)(want < cap ? want : cap);
}

template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_hmc_run(const double* __restrict__ XsT, int64_t ldx, int64_t n,
                                                 const double* __restrict__ alpha, Hyper h, int64_t P,
                                                 double* __restrict__ S, double* __restrict__ adapt,
                                                 const double* __restrict__ inv_mass, unsigned long long seed,
                                                 int64_t it0, int niter, int do_adapt, double ystd, double ymean,
                                                 double temp, int hist_from, double* __restrict__ hist, int thin,
                                                 double* __restrict__ keep, double* __restrict__ dbg, Gate gt,
                                                 int lgroups) {
  constexpr int NT = 256, NW = NT / 64;
  extern __shared__ double lrows[];            // [d + 1][256 lgroups]: training points resident in LDS (chain_lds_groups)
  __shared__ double x[DCAP], xs[DCAP], red[NW][DCAP + 1], lp_s, mean_s, gred[4], kin_s[2];
  __shared__ double u0[DCAP], g0[DCAP], x0[DCAP], lp0, mean0, eps_s;
  __shared__ int acc_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.x;
  const int d = h.d, sw = 3 * d + 2;
  double* Sc = S + c * sw;
  double* ad = adapt + c * 5;
  if (t < d) {
    u0[t] = Sc[t];
    g0[t] = Sc[d + t];
    x0[t] = Sc[2 * d + t];
  }
  if (t == 0) {
    lp0 = Sc[3 * d];
    mean0 = Sc[3 * d + 1];
    eps_s = ad[0];
  }
  const unsigned long long ckey = hmc_mix64(seed ^ hmc_mix64((unsigned long long)c));
  // The first RMAX training points of a thread (rows t, t + 256, ...) are loaded ONCE per launch - a launch is hundreds of
  // leapfrog steps, each of which would otherwise wait for the same global loads again -; points past n carry alpha = 0.
  // The next 256 lgroups rows live in LDS (every thread its own), and what fits neither is streamed from L2 in every step.
  constexpr int RMAX = ChainRows<DCAP>::HMC, UNR = ChainRows<DCAP>::STREAM;
  const int nrow = (int)((n + NT - 1) / NT);
  const int lld = NT * lgroups;
  double cx[RMAX][DCAP], ca[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int64_t i = t + NT * r;
    ca[r] = (i < n) ? alpha[i] : 0.0;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) cx[r][j] = (j < d && i < n) ? XsT[j * ldx + i] : 0.0;
  }
  for (int q = 0; q < lgroups; ++q) {
    const int64_t i = t + (int64_t)NT * (RMAX + q);
    for (int j = 0; j < d; ++j) lrows[j * lld + q * NT + t] = (i < n) ? XsT[j * ldx + i] : 0.0;
    lrows[d * lld + q * NT + t] = (i < n) ? alpha[i] : 0.0;
  }
  __syncthreads();
  // Thread t < d owns coordinate t of the trajectory (position, momentum, gradient: registers); the others only ever need
  // the point itself (x, xs in LDS).  A leapfrog step is two barriers: [every thread: its training points, the wave sums]
  // | [thread t < d: gradient, momentum, next position -> x, xs] |.  Thread 0 carries the chain's scalars and its dual-
  // averaging state in registers for the whole launch.
  const double im_r = (t < d) ? inv_mass[t] : 0.0, inv_ls = (t < d) ? 1.0 / h.ls[t] : 0.0;
  double a_eps = 0.0, a_mu = 0.0, a_hbar = 0.0, a_leb = 0.0, a_m = 0.0;
  if (t == 0) {
    a_eps = ad[0];
    a_mu = ad[1];
    a_hbar = ad[2];
    a_leb = ad[3];
    a_m = ad[4];
  }
  for (int it = 0; it < niter; ++it) {
    const unsigned long long ikey = ckey + ((unsigned long long)(it0 + it) << 12);
    const double eps = eps_s;
    double u_r = 0.0, pm_r = 0.0, p0_r = 0.0, g_r = 0.0, x_r = 0.0;
    // position update of thread t < d, and the point in cube coordinates for everybody
    auto advance = [&]() {
      u_r += eps * im_r * pm_r;
      double xv = 1.0 / (1.0 + exp(-u_r));
      xv = xv < 1e-12 ? 1e-12 : (xv > 1.0 - 1e-12 ? 1.0 - 1e-12 : xv);
      x_r = xv;
      x[t] = xv;
      xs[t] = xv * inv_ls;
    };
    if (t < d) {                                               // momentum ~ N(0, M), M = diag(1 / inv_mass)
      const double a = hmc_u01(hmc_mix64(ikey + 2 * t)), b = hmc_u01(hmc_mix64(ikey + 2 * t + 1));
      const double z = sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
      p0_r = z / sqrt(im_r);
      pm_r = p0_r + 0.5 * eps * g0[t];
      u_r = u0[t];
      advance();
    }
    const int L = 4 + (int)(hmc_mix64(ikey + 4000) % 9ull);    // (every thread: the same counter hash)
    __syncthreads();
    for (int s = 0; s < L; ++s) {
      double ms = 0.0, gm[DCAP];
#pragma unroll
      for (int j = 0; j < DCAP; ++j) gm[j] = 0.0;
      // one training point: mean and mean-gradient contributions (the difference is formed twice rather than kept)
      auto point = [&](const double (&xr)[DCAP], double a) {
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          const double df = (j < d) ? xr[j] - xs[j] : 0.0;
          r2 += df * df;
        }
        const double kv = kern_eval<KERN>(r2, h.kvar);
        const double ag = a * kern_grad_factor<KERN>(r2, h.kvar, kv);
        ms += a * kv;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) gm[j] += ag * ((j < d) ? xr[j] - xs[j] : 0.0);
      };
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < nrow) point(cx[r], ca[r]);                     // (uniform) this thread's points in registers,
      for (int q = 0; q < lgroups; ++q) {                      // in LDS,
        double xr[DCAP];
#pragma unroll
        for (int j = 0; j < DCAP; ++j) xr[j] = (j < d) ? lrows[j * lld + q * NT + t] : 0.0;
        point(xr, lrows[d * lld + q * NT + t]);
      }
      for (int64_t i = t + (int64_t)NT * (RMAX + lgroups); i < n; i += UNR * NT) {  // and streamed: ascending rows throughout
        double xr[UNR][DCAP], ar[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {                                // (all loads of the group first)
          const int64_t iq = i + q * NT;
          ar[q] = (iq < n) ? alpha[iq] : 0.0;
#pragma unroll
          for (int j = 0; j < DCAP; ++j) xr[q][j] = (j < d && iq < n) ? XsT[j * ldx + iq] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) point(xr[q], ar[q]);
      }
      ms = chain_wave_sum(ms);
      if (lane == 0) red[wave][DCAP] = ms;
      {
        const double v = wave_sum_components<DCAP>(gm, lane);
        if ((lane & (64 / DCAP - 1)) == 0) red[wave][lane / (64 / DCAP)] = v;
      }
      // classifier gate (clf_gp.py:173-205): an infeasible point has mean = minus_inf and no mean gradient - its
      // trajectory ends in a state that the Metropolis test never accepts
      if (gt.n_sv > 0) {                                       // (the gate's 256 partial sums: k_gate's order)
        const double gs = gate_partial<DCAP>(gt, x, d, t);
        if (lane == 0) gred[wave] = gs;
      }
      __syncthreads();
      const bool ok = gt.n_sv > 0 ? gate_feasible(gt, gate_combine(gt, gred)) : true;
      // (the waves' sums in wave order: ((r0 + r1) + r2) + r3)
      auto wsum = [&](int j) {
        double sres = red[0][j];
#pragma unroll
        for (int w_ = 1; w_ < NW; ++w_) sres += red[w_][j];
        return sres;
      };
      const bool last = s == L - 1;
      if (t < d) {
        const double dm = ok ? wsum(t) * inv_ls : 0.0;
        g_r = dm * ystd / temp * (x_r * (1.0 - x_r)) + (1.0 - 2.0 * x_r);
        pm_r += (last ? 0.5 * eps : eps) * g_r;
        if (!last) advance();
      }
      if (last) {
        if (wave == 0) {                                       // kinetic energies at both ends (lanes = coordinates)
          const double k0 = chain_wave_sum(p0_r * p0_r * im_r), k1 = chain_wave_sum(pm_r * pm_r * im_r);
          if (lane == 0) {
            kin_s[0] = k0;
            kin_s[1] = k1;
          }
        } else if (wave == 1) {                                // the end point's log-density
          double jl = (lane < d) ? log(x[lane]) + log1p(-x[lane]) : 0.0;
          jl = chain_wave_sum(jl);
          if (lane == 0) {
            const double m = ok ? wsum(DCAP) * ystd + ymean : gt.minus_inf;
            mean_s = m;
            lp_s = m / temp + jl;
          }
        }
      }
      __syncthreads();
    }
    if (t == 0) {                                              // Metropolis test and the chain's step-size update
      const double h0 = lp0 - 0.5 * kin_s[0], h1 = lp_s - 0.5 * kin_s[1];
      double ap = 0.0;
      if (isfinite(h1)) ap = h1 >= h0 ? 1.0 : exp(h1 - h0);
      const double r = hmc_u01(hmc_mix64(ikey + 4001));
      const int acc = r < ap;
      acc_s = acc;
      if (acc) {
        lp0 = lp_s;
        mean0 = mean_s;
      }
      if (do_adapt) {
        constexpr double t0 = 10.0, gamma = 0.05, kappa = 0.75, target = 0.8;
        const double m = a_m + 1.0;
        const double hbar = (1.0 - 1.0 / (m + t0)) * a_hbar + (target - ap) / (m + t0);
        const double le = a_mu - sqrt(m) / gamma * hbar;
        const double eta = pow(m, -kappa);
        a_hbar = hbar;
        a_leb = eta * le + (1.0 - eta) * a_leb;
        a_m = m;
        double e = exp(le);
        e = e < 1e-4 ? 1e-4 : (e > 2.0 ? 2.0 : e);
        a_eps = e;
        eps_s = e;
      }
      if (dbg && it == niter - 1) {
        double* dc = dbg + c * (d + 3);
        dc[d] = (double)L;
        dc[d + 1] = r;
        dc[d + 2] = ap;
      }
    }
    if (dbg && it == niter - 1 && t < d) dbg[c * (d + 3) + t] = p0_r;
    __syncthreads();
    if (t < d) {
      if (acc_s) {
        u0[t] = u_r;
        g0[t] = g_r;
        x0[t] = x_r;
      }
      if (hist && it >= hist_from) hist[((int64_t)(it - hist_from) * P + c) * d + t] = u0[t];
      if (keep && (it + 1) % thin == 0) keep[((int64_t)((it + 1) / thin - 1) * P + c) * (d + 1) + t] = x0[t];
    }
    if (t == 0 && keep && (it + 1) % thin == 0) keep[((int64_t)((it + 1) / thin - 1) * P + c) * (d + 1) + d] = mean0;
    // (no barrier here: the next iteration's first one comes before anything of this one is read again - u0 / g0 / x0 are
    //  read by their own thread only, x / xs are written by advance() after every reader of this iteration has passed the
    //  barrier above, eps_s and acc_s were written before it)
  }
  if (t < d) {
    Sc[t] = u0[t];
    Sc[d + t] = g0[t];
    Sc[2 * d + t] = x0[t];
  }
  if (t == 0) {
    Sc[3 * d] = lp0;
    Sc[3 * d + 1] = mean0;
    if (do_adapt) {
      ad[0] = a_eps;
      ad[2] = a_hbar;
      ad[3] = a_leb;
      ad[4] = a_m;
    }
  }
}


// ---- constrained random walks for nested sampling, whole walks on the device -----------------------------------------
// The replacement search of a nested-sampling iteration (dynesty's 'rwalk', the reference's choice: samplers.py:64, 152):
// every walker starts at a live point and takes `walks` Metropolis steps inside {mean(x) > lstar, unit cube}; a step is
// x' = x + step[d x d] z, z ~ N(0, I) (step = scale x the Cholesky factor of the live points' covariance, lower, row-major).
// One workgroup = one walker, every step's surrogate mean by the reduction of k_hmc_run (training points in registers),
// the classifier gate applied like everywhere else (an infeasible proposal has mean = minus_inf and is never accepted).
// Random numbers: the counter hash of k_hmc_run on (seed, walker, step, index).
//   X    [P][d]  in: start points, out: end points        logl [P]  in / out: physical-unit mean at the point
//   nacc [P]     accepted steps                            nin  [P]  proposals inside the cube (= surrogate calls)
//   dbg  [P][d]  the LAST proposal of every walker (tests)                                             (may be null)
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_rwalk(const double* __restrict__ XsT, int64_t ldx, int64_t n,
                                               const double* __restrict__ alpha, Hyper h, double* __restrict__ X,
                                               double* __restrict__ logl, const double* __restrict__ step, double lstar,
                                               int walks, unsigned long long seed, double ystd, double ymean,
                                               int* __restrict__ nacc, int* __restrict__ nin, double* __restrict__ dbg,
                                               Gate gt, int lgroups) {
  constexpr int NT = 256, NW = NT / 64;
  extern __shared__ double lrows[];
  __shared__ double x[DCAP], xp[DCAP], xs[DCAP], z[DCAP], red[NW], gred[4], lx, stp[DCAP * DCAP];
  __shared__ int inside_s, na_s, ni_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.x;
  const int d = h.d;
  if (t < d) x[t] = X[c * d + t];
  for (int k = t; k < d * d; k += NT) stp[k] = step[k];               // (read in every step: keep it in LDS)
  if (t == 0) {
    lx = logl[c];
    na_s = 0;
    ni_s = 0;
  }
  const unsigned long long ckey = hmc_mix64(seed ^ hmc_mix64((unsigned long long)c));
  // training points in registers, in LDS and streamed, like k_hmc_run (only the mean is needed here)
  constexpr int RMAX = ChainRows<DCAP>::WALK, UNR = ChainRows<DCAP>::STREAM;
  const int nrow = (int)((n + NT - 1) / NT);
  const int lld = NT * lgroups;
  double cx[RMAX][DCAP], ca[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int64_t i = t + NT * r;
    ca[r] = (i < n) ? alpha[i] : 0.0;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) cx[r][j] = (j < d && i < n) ? XsT[j * ldx + i] : 0.0;
  }
  for (int q = 0; q < lgroups; ++q) {
    const int64_t i = t + (int64_t)NT * (RMAX + q);
    for (int j = 0; j < d; ++j) lrows[j * lld + q * NT + t] = (i < n) ? XsT[j * ldx + i] : 0.0;
    lrows[d * lld + q * NT + t] = (i < n) ? alpha[i] : 0.0;
  }
  __syncthreads();
  for (int s = 0; s < walks; ++s) {
    const unsigned long long ikey = ckey + ((unsigned long long)s << 12);
    if (t < d) {
      const double a = hmc_u01(hmc_mix64(ikey + 2 * t)), b = hmc_u01(hmc_mix64(ikey + 2 * t + 1));
      z[t] = sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
    }
    __syncthreads();
    if (wave == 0) {                                                       // (d <= 32: the coordinates live in wave 0)
      bool in = true;
      if (t < d) {
        double v = x[t];
        for (int j = 0; j <= t; ++j) v += stp[t * d + j] * z[j];            // (lower-triangular factor)
        xp[t] = v;
        xs[t] = v / h.ls[t];
        in = (v >= 0.0) && (v <= 1.0);
      }
      const int all_in = __all(in);
      if (t == 0) {
        inside_s = all_in;
        ni_s += all_in;
      }
    }
    __syncthreads();
    if (inside_s) {                                                        // (uniform: a proposal outside costs no evaluation)
      double ms = 0.0;
      auto point = [&](const double (&xr)[DCAP], double a) {
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          const double df = (j < d) ? xr[j] - xs[j] : 0.0;
          r2 += df * df;
        }
        ms += a * kern_eval<KERN>(r2, h.kvar);
      };
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < nrow) point(cx[r], ca[r]);
      for (int q = 0; q < lgroups; ++q) {
        double xr[DCAP];
#pragma unroll
        for (int j = 0; j < DCAP; ++j) xr[j] = (j < d) ? lrows[j * lld + q * NT + t] : 0.0;
        point(xr, lrows[d * lld + q * NT + t]);
      }
      for (int64_t i = t + (int64_t)NT * (RMAX + lgroups); i < n; i += UNR * NT) {
        double xr[UNR][DCAP], ar[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {                                // (all loads of the group first)
          const int64_t iq = i + q * NT;
          ar[q] = (iq < n) ? alpha[iq] : 0.0;
#pragma unroll
          for (int j = 0; j < DCAP; ++j) xr[q][j] = (j < d && iq < n) ? XsT[j * ldx + iq] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) point(xr[q], ar[q]);
      }
      ms = chain_wave_sum(ms);
      if (lane == 0) red[wave] = ms;
      if (gt.n_sv > 0) {
        const double gs = gate_partial<DCAP>(gt, xp, d, t);
        if (lane == 0) gred[wave] = gs;
      }
      __syncthreads();
      if (t == 0) {
        double m = red[0];
#pragma unroll
        for (int w_ = 1; w_ < NW; ++w_) m += red[w_];
        m = m * ystd + ymean;
        if (gt.n_sv > 0 && !gate_feasible(gt, gate_combine(gt, gred))) m = gt.minus_inf;
        const int acc = m > lstar;
        inside_s = acc;                                                    // (reused: accepted)
        if (acc) {
          lx = m;
          ++na_s;
        }
      }
      __syncthreads();
      if (inside_s && t < d) x[t] = xp[t];
    }
    __syncthreads();
  }
  if (t < d) {
    X[c * d + t] = x[t];
    if (dbg) dbg[c * d + t] = xp[t];
  }
  if (t == 0) {
    logl[c] = lx;
    nacc[c] = na_s;
    nin[c] = ni_s;
  }
}

// mode 0: EI, 1: LogEI.  out = +EI / +logEI (the reference minimises the negative).  A mean of -inf is the gate's mark
// (k_gate): the reference's predict_single returns minus_inf there (clf_gp.py:201-204)
__global__ void k_ei(const double* __restrict__ mu, const double* __restrict__ var, int64_t n, double best_y, double zeta,
                     int mode, double* __restrict__ out, double minus_inf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = var[i];
  const double lo = mode ? 1e-18 : 1e-20;
  if (v < lo) v = lo;
  const double sigma = sqrt(v);
  double m = mu[i];
  if (m == -INFINITY) m = minus_inf;
  const double u = (m - zeta - best_y) / sigma;
  out[i] = mode ? (log_ei_helper(u) + log(sigma)) : (ei_helper(u) * sigma);
}

}  // namespace bobe
