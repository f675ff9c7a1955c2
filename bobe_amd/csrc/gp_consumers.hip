// libbobe_gp.so, consumers unit: what reads the factorised surrogate besides the sweep - Hamiltonian Monte Carlo on the
// posterior mean, EI / LogEI, the classifier gate of GPwithClassifier, GP.kernel, the device-side clone - and the
// handle's teardown.  Kernels: kernels_common.hpp, consumer_kernels.hpp.
#include "gp_handle.hpp"

#include "consumer_kernels.hpp"

using namespace bobe;

namespace bobe {
void configure_consumer_kernels() {       // (the chain kernels keep training points in up to 150 KB of LDS)
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return;
#define BIG(KE, DC)                                         \
  allow_big_lds((k_hmc_run<KE, DC>), CHAIN_LDS_BYTES);      \
  allow_big_lds((k_rwalk<KE, DC>), CHAIN_LDS_BYTES)
  BIG(0, 8); BIG(0, 16); BIG(0, 32); BIG(1, 8); BIG(1, 16); BIG(1, 32);
#undef BIG
  done[dev] = true;
}
}  // namespace bobe

// ---- classifier gate ------------------------------------------------------------------------------------------------
void bobe_gp::set_gate(const double* sv, int64_t n_sv, const double* dual, double intercept, double gamma, double threshold,
                       double minus_inf) {
  use();
  sync();
  if (!sv || n_sv <= 0) {                      // clear
    gate = Gate{nullptr, nullptr, 0, 0, 0.0, 0.0, threshold, minus_inf};
    return;
  }
  if (!dual) throw Err(BOBE_ERR_ARG, "dual_coef is NULL");
  if (n_sv > INT_MAX) throw Err(BOBE_ERR_ARG, "too many support vectors");
  // support vectors SoA (coordinate j of vector i at svT[j * n_sv + i]): a wave's lanes read consecutive vectors
  std::vector<double> host_sv((size_t)n_sv * d), svt((size_t)n_sv * d);
  const double* hs = sv;
  if (is_device_ptr(sv)) {
    HIPCHK(hipMemcpy(host_sv.data(), sv, host_sv.size() * sizeof(double), hipMemcpyDeviceToHost));
    hs = host_sv.data();
  }
  for (int64_t i = 0; i < n_sv; ++i)
    for (int j = 0; j < d; ++j) svt[(size_t)j * n_sv + i] = hs[i * d + j];
  gate_sv.ensure(svt.size() * sizeof(double));
  gate_dual.ensure((size_t)n_sv * sizeof(double));
  HIPCHK(hipMemcpy(gate_sv.p, svt.data(), svt.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(gate_dual.p, dual, (size_t)n_sv * sizeof(double),
                   is_device_ptr(dual) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  gate = Gate{gate_sv.d(), gate_dual.d(), n_sv, (int)n_sv, intercept, gamma, threshold, minus_inf};
}

void bobe_gp::gate_apply(const double* xq_dev, int64_t C, double* decision, double* feasible, double* mean, double* var,
                         double* dmean, double* dvar) {
  if (gate.n_sv <= 0) throw Err(BOBE_ERR_STATE, "no classifier gate is set (bobe_gp_set_gate)");
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
#define GT(DC) \
  hipLaunchKernelGGL((k_gate<DC>), dim3((unsigned)C), dim3(256), 0, stream, gate, xq_dev, d, decision, feasible, mean, var, dmean, dvar)
  if (dcap == 8) GT(8); else if (dcap == 16) GT(16); else GT(32);
#undef GT
  LAUNCH_CHECK();
}

void bobe_gp::gate_eval(const double* Xq, int64_t C, double* decision, double* feasible) {
  if (C <= 0) throw Err(BOBE_ERR_ARG, "C must be positive");
  use();
  const double* cin = fetch(Xq, (size_t)C * d, in_stage);
  double* d_dec = out_dev(decision, C, o_mean);
  double* d_fe = out_dev(feasible, C, o_var);
  gate_apply(cin, C, d_dec, d_fe, nullptr, nullptr, nullptr, nullptr);
  out_finish(decision, C, o_mean);
  out_finish(feasible, C, o_var);
  sync();
}

void bobe_gp::acq_ei(const double* Xq, int64_t C, double best_y, double zeta, int mode, double* out) {
  use();
  o_mean.ensure(C * sizeof(double));
  o_var.ensure(C * sizeof(double));
  // (predict_single, acquisition.py:246 / 323: gated for a GPwithClassifier)
  sweep(Xq, C, nullptr, 0, 1.0, nullptr, nullptr, o_mean.d(), o_var.d(), 1, nullptr, nullptr, nullptr, nullptr, nullptr, true);
  double* d_out = out_dev(out, C, o_wipv);
  hipLaunchKernelGGL(k_ei, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, (const double*)o_mean.d(),
                     (const double*)o_var.d(), C, best_y, zeta, mode, d_out, gate.minus_inf);
  LAUNCH_CHECK();
  out_finish(out, C, o_wipv);
  sync();
}

void bobe_gp::hmc_leapfrog(int64_t P, double* U, double* Pm, const double* inv_mass, double eps, int L, double y_std,
                           double y_mean, double temp, double* logp, double* grad, double* mean, double* X) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (P <= 0 || L < 1 || !(temp > 0.0)) throw Err(BOBE_ERR_ARG, "bad argument");
  use();
  const size_t pd = (size_t)P * d;
  // staging: [U | Pm | grad | X] (P*d each), [logp | mean] (P each), inv_mass (d)
  in_stage.ensure((4 * pd + 2 * (size_t)P + d) * sizeof(double));
  double* dU = in_stage.d();
  double* dP = dU + pd;
  double* dG = dP + pd;
  double* dX = dG + pd;
  double* dL = dX + pd;
  double* dM = dL + P;
  double* dI = dM + P;
  const bool dev = is_device_ptr(U);
  const hipMemcpyKind in = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const hipMemcpyKind out = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIPCHK(hipMemcpyAsync(dU, U, pd * sizeof(double), in, stream));
  HIPCHK(hipMemcpyAsync(dP, Pm, pd * sizeof(double), in, stream));
  HIPCHK(hipMemcpyAsync(dI, inv_mass, (size_t)d * sizeof(double), is_device_ptr(inv_mass) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        stream));
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
#define HL(KE, DC)                                                                                                   \
  hipLaunchKernelGGL((k_hmc_leapfrog<KE, DC>), dim3((unsigned)P), dim3(256), 0, stream, (const double*)XsT.d(),   \
                     Np, N, (const double*)alpha.d(), hyp, dU, dP, (const double*)dI, eps, L, y_std, y_mean, \
                     temp, dL, dG, dM, dX, gate)
  if (hyp.kern == 0) {
    if (dcap == 8) HL(0, 8); else if (dcap == 16) HL(0, 16); else HL(0, 32);
  } else {
    if (dcap == 8) HL(1, 8); else if (dcap == 16) HL(1, 16); else HL(1, 32);
  }
#undef HL
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(U, dU, pd * sizeof(double), out, stream));
  HIPCHK(hipMemcpyAsync(Pm, dP, pd * sizeof(double), out, stream));
  HIPCHK(hipMemcpyAsync(grad, dG, pd * sizeof(double), out, stream));
  HIPCHK(hipMemcpyAsync(X, dX, pd * sizeof(double), out, stream));
  HIPCHK(hipMemcpyAsync(logp, dL, (size_t)P * sizeof(double), out, stream));
  HIPCHK(hipMemcpyAsync(mean, dM, (size_t)P * sizeof(double), out, stream));
  sync();
}

void bobe_gp::hmc_run(int64_t P, double* state, double* adapt, const double* inv_mass, uint64_t seed, int64_t it0, int niter,
                      int do_adapt, double y_std, double y_mean, double temp, int hist_from, double* hist, int thin,
                      double* keep, double* dbg) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (P <= 0 || niter < 1 || it0 < 0 || !(temp > 0.0) || thin < 1 || hist_from < 0 || hist_from > niter)
    throw Err(BOBE_ERR_ARG, "bad argument");
  use();
  const size_t sw = 3 * (size_t)d + 2, ns = (size_t)P * sw, na = (size_t)P * 5;
  const size_t nh = hist ? (size_t)(niter - hist_from) * P * d : 0;
  const size_t nk = keep ? (size_t)(niter / thin) * P * (d + 1) : 0;
  const size_t nd = dbg ? (size_t)P * (d + 3) : 0;
  // staging: state | adapt | inv_mass | hist | keep | dbg
  in_stage.ensure((ns + na + d + nh + nk + nd) * sizeof(double));
  double* dS = in_stage.d();
  double* dA = dS + ns;
  double* dI = dA + na;
  double* dH = dI + d;
  double* dK = dH + nh;
  double* dD = dK + nk;
  HIPCHK(hipMemcpyAsync(dS, state, ns * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dA, adapt, na * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dI, inv_mass, (size_t)d * sizeof(double), hipMemcpyHostToDevice, stream));
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  // training points of a chain's workgroup: registers first, then LDS (chain_lds_groups), the rest streamed
#define HR(KE, DC)                                                                                                  \
  do {                                                                                                              \
    const int lg = chain_lds_groups(N, d, ChainRows<DC>::HMC);                                                      \
    hipLaunchKernelGGL((k_hmc_run<KE, DC>), dim3((unsigned)P), dim3(256), (size_t)lg * 256 * (d + 1) * sizeof(double), \
                       stream, (const double*)XsT.d(), Np, N, (const double*)alpha.d(), hyp, P, dS, dA,             \
                       (const double*)dI, (unsigned long long)seed, it0, niter, do_adapt, y_std, y_mean, temp,      \
                       hist_from, hist ? dH : nullptr, thin, keep ? dK : nullptr, dbg ? dD : nullptr, gate, lg);    \
  } while (0)
  if (hyp.kern == 0) {
    if (dcap == 8) HR(0, 8); else if (dcap == 16) HR(0, 16); else HR(0, 32);
  } else {
    if (dcap == 8) HR(1, 8); else if (dcap == 16) HR(1, 16); else HR(1, 32);
  }
#undef HR
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(state, dS, ns * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(adapt, dA, na * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (hist) HIPCHK(hipMemcpyAsync(hist, dH, nh * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (keep) HIPCHK(hipMemcpyAsync(keep, dK, nk * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (dbg) HIPCHK(hipMemcpyAsync(dbg, dD, nd * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync();
}

void bobe_gp::rwalk(int64_t P, double* Xw, double* logl, const double* step, double lstar, int walks, uint64_t seed,
                    double y_std, double y_mean, int* nacc, int* nin, double* dbg) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (P <= 0 || walks < 1) throw Err(BOBE_ERR_ARG, "bad argument");
  use();
  const size_t pd = (size_t)P * d;
  // staging: X | logl | step | dbg (doubles), then nacc | nin (ints)
  in_stage.ensure((2 * pd + (size_t)P + (size_t)d * d) * sizeof(double) + 2 * (size_t)P * sizeof(int));
  double* dX = in_stage.d();
  double* dL = dX + pd;
  double* dS = dL + P;
  double* dD = dS + (size_t)d * d;
  int* dA = reinterpret_cast<int*>(dD + pd);
  int* dN = dA + P;
  HIPCHK(hipMemcpyAsync(dX, Xw, pd * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dL, logl, (size_t)P * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(dS, step, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, stream));
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
#define RW(KE, DC)                                                                                                 \
  do {                                                                                                             \
    const int lg = chain_lds_groups(N, d, ChainRows<DC>::WALK);                                                    \
    hipLaunchKernelGGL((k_rwalk<KE, DC>), dim3((unsigned)P), dim3(256), (size_t)lg * 256 * (d + 1) * sizeof(double), \
                       stream, (const double*)XsT.d(), Np, N, (const double*)alpha.d(), hyp, dX, dL,               \
                       (const double*)dS, lstar, walks, (unsigned long long)seed, y_std, y_mean, dA, dN,           \
                       dbg ? dD : nullptr, gate, lg);                                                              \
  } while (0)
  if (hyp.kern == 0) {
    if (dcap == 8) RW(0, 8); else if (dcap == 16) RW(0, 16); else RW(0, 32);
  } else {
    if (dcap == 8) RW(1, 8); else if (dcap == 16) RW(1, 16); else RW(1, 32);
  }
#undef RW
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(Xw, dX, pd * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(logl, dL, (size_t)P * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(nacc, dA, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(nin, dN, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, stream));
  if (dbg) HIPCHK(hipMemcpyAsync(dbg, dD, pd * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync();
}

void bobe_gp::kernel_eval(const double* A, int64_t nA, const double* B, int64_t nB, const double* ls, double kvar,
                          double noise, int include_noise, double* out, bool sqdist) {
  if (nA < 1 || nB < 1) throw Err(BOBE_ERR_ARG, "empty input");
  if (include_noise && nA != nB) throw Err(BOBE_ERR_ARG, "include_noise needs a square kernel matrix (gp.py:153)");
  use();
  Hyper hk = hyp;
  if (ls) {
    for (int j = 0; j < d; ++j) hk.ls[j] = ls[j];
    hk.kvar = kvar;
    hk.noise = noise;
  }
  if (sqdist) {                                  // dist_sq: unscaled coordinates, kernel id 2 = the distance itself
    for (int j = 0; j < d; ++j) hk.ls[j] = 1.0;
    hk.kern = 2;
  }
  const int64_t pa = round_up(nA, TILE), pb = round_up(nB, TILE);
  const double* a_in = fetch(A, (size_t)nA * d, in_stage);
  const double* b_in = fetch(B, (size_t)nB * d, z_stage);
  kin_a.ensure((size_t)d * pa * sizeof(double));
  kin_b.ensure((size_t)d * pb * sizeof(double));
  kout.ensure((size_t)pa * pb * sizeof(double));
  scale(a_in, nA, pa, hk, kin_a.d(), pa);
  scale(b_in, nB, pb, hk, kin_b.d(), pb);
  kernel_matrix_cross(kin_a.d(), pa, nA, pa, kin_b.d(), pb, nB, pb, hk, kout.d(), pb);
  const bool dev = is_device_ptr(out);
  HIPCHK(hipMemcpy2DAsync(out, (size_t)nB * sizeof(double), kout.p, (size_t)pb * sizeof(double),
                          (size_t)nB * sizeof(double), (size_t)nA, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          stream));
  sync();
  if (include_noise) {
    // noise * eye(n) (gp.py:153): added on the host side of the copy for host outputs, by a tiny kernel otherwise
    if (!dev) {
      for (int64_t i = 0; i < nA; ++i) out[i * nB + i] += hk.noise;
    } else {
      std::vector<double> dg((size_t)nA);
      HIPCHK(hipMemcpy2D(dg.data(), sizeof(double), out, (size_t)(nB + 1) * sizeof(double), sizeof(double), (size_t)nA,
                         hipMemcpyDeviceToHost));
      for (auto& v : dg) v += hk.noise;
      HIPCHK(hipMemcpy2D(out, (size_t)(nB + 1) * sizeof(double), dg.data(), sizeof(double), sizeof(double), (size_t)nA,
                         hipMemcpyHostToDevice));
    }
  }
}

// GP.copy (gp.py:740-750) without leaving the device: training data, hyper-parameters and factorised state by
// device-to-device copies.  The classifier gate is not part of a GP's state_dict and is not cloned.
void bobe_gp::clone_from(bobe_gp& src) {
  if (this == &src) return;
  if (d != src.d || kern != src.kern || device != src.device)
    throw Err(BOBE_ERR_ARG, "clone needs handles of the same kernel, dimension and device");
  if (!src.have_data) throw Err(BOBE_ERR_STATE, "source holds no data");
  src.use();
  src.sync();
  sync();
  N = src.N;
  hyp = src.hyp;
  pivot_ulp = src.pivot_ulp;
  refine_kappa = src.refine_kappa;
  solve_block = src.solve_block;
  solve_panel = src.solve_panel;
  solve_chunk = src.solve_chunk;
  refine_v = src.factored && src.refine_v;      // (A is copied with its diagonal blocks as the source left them)
  if (Np != src.Np) {
    Np = src.Np;
    nb = src.nb;
    alloc_for_n();
  }
  const size_t mat = (size_t)src.Np * src.Np * sizeof(double), vec = (size_t)src.Np * sizeof(double);
  X.ensure((size_t)(src.Np + TILE) * src.d * sizeof(double));
  HIPCHK(hipMemcpyAsync(X.p, src.X.p, (size_t)src.N * src.d * sizeof(double), hipMemcpyDeviceToDevice, stream));
  HIPCHK(hipMemcpyAsync(y.p, src.y.p, vec, hipMemcpyDeviceToDevice, stream));
  if (src.factored) {
    HIPCHK(hipMemcpyAsync(XsT.p, src.XsT.p, (size_t)src.d * vec, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(A.p, src.A.p, mat, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(Linv.p, src.Linv.p, mat, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(alpha.p, src.alpha.p, vec, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(w.p, src.w.p, vec, hipMemcpyDeviceToDevice, stream));
  }
  sync();
  have_data = true;
  factored = src.factored;
  forget_z();
  not_pd = src.not_pd;
}

// everything the handle owns on the device and in pinned memory (bobe_gp_destroy)
void bobe_gp::release_all() {
  (void)hipSetDevice(device);
  if (stream) (void)hipStreamSynchronize(stream);
  DBuf* bufs[] = {&X, &y, &XsT, &XsT2, &A, &Linv, &A2, &Linv2, &Tmp, &alpha, &w, &alpha2, &w2, &part, &gpart, &res, &info,
                  &probs, &flags, &diag, &in_stage, &z_stage, &CsT, &ZsT, &kXC, &kXZ, &VZ, &WZ, &basez, &sc, &qpart, &pv,
                  &ps, &o_mean, &o_var, &o_wipv, &o_wipstd, &o_misc, &kin_a, &kin_b, &kout, &wg_ws, &gate_sv, &gate_dual, &vxc, &vxc2};
  for (DBuf* b : bufs) b->release();
  for (auto& pr : prof_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (h_res) (void)hipHostFree(h_res);
  if (h_in) (void)hipHostFree(h_in);
  h_in = nullptr;
  for (hipStream_t st : slot_streams) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  auto free_eg = [](EvalGraph& e) {
    for (int w_ = 0; w_ < 2; ++w_)
      if (e.exec[w_]) (void)hipGraphExecDestroy(e.exec[w_]);
    if (e.h_hyp) (void)hipHostFree(e.h_hyp);
    e.hyp_dev.release();
  };
  free_eg(eg);
  for (Slot* sl : slots) {
    free_eg(sl->eg);
    DBuf* sb[] = {&sl->XsT2, &sl->A2, &sl->Linv2, &sl->Tmp, &sl->alpha2, &sl->w2, &sl->part, &sl->gpart, &sl->res,
                  &sl->info, &sl->flags, &sl->diag};
    for (DBuf* b : sb) b->release();
    if (sl->h_res) (void)hipHostFree(sl->h_res);
    if (sl->ev) (void)hipEventDestroy(sl->ev);
    delete sl;
  }
  slots.clear();
  {
    DBuf* bb[] = {&bw.A, &bw.Linv, &bw.Tmp, &bw.XsT, &bw.w, &bw.alpha, &bw.part, &bw.gpart, &bw.res, &bw.info, &bw.hyp, &bw.diag};
    for (DBuf* b : bb) b->release();
    if (bw.h_hyp) (void)hipHostFree(bw.h_hyp);
    if (bw.h_res) (void)hipHostFree(bw.h_res);
  }
  for (auto& kv : chol_plans) {
    kv.second.d_jobs.release();
    kv.second.d_colk0.release();
  }
  if (ev_batch) (void)hipEventDestroy(ev_batch);
  if (own_stream && stream) (void)hipStreamDestroy(stream);
}
