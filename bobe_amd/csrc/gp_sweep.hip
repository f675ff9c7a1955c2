// libbobe_gp.so, sweep unit: batched prediction and the integrated-variance acquisition sweep, the input gradients of the
// posterior and of the WIPV / WIPStd scores, the rank-b append.  Kernels: kernels_common.hpp, sweep_kernels.hpp.
#include "gp_handle.hpp"

#include "sweep_kernels.hpp"

using namespace bobe;

namespace bobe {
void configure_sweep_kernels() {
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return;
  allow_big_lds(k_trimul, GEMM_SMEM_BYTES);
  allow_big_lds(k_trimul_t, GEMM_SMEM_BYTES);
  allow_big_lds(k_trimul_v64, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trimul_t64, GEMM64_SMEM_BYTES);
  allow_big_lds(k_blk_step, GEMM_SMEM_BYTES);
  allow_big_lds(k_cross_vv<128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_cross_vv<64>, GEMM64_SMEM_BYTES);
  done[dev] = true;
}
}  // namespace bobe

void bobe_gp::decide_refinement(double min_diag) {
  const double piv = min_diag * min_diag;
  refine_v = refine_kappa >= 0.0 && piv > 0.0 && (hyp.kvar + hyp.noise) / piv > refine_kappa;
}

void bobe_gp::solve_v(double* B, int64_t ldb, int64_t ncp, double* V, int64_t ldv, double* qp, int64_t ldq) {
  if (!refine_v) {
    hipLaunchKernelGGL(k_trimul, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)B, ldb, V, ldv, qp, ldq, (const double*)nullptr,
                       (int64_t)0, 0, (double*)nullptr, (int64_t)0);
    return;
  }
  if (!V) throw Err(BOBE_ERR_STATE, "solve_v: the blocked substitution needs a buffer for V");
  // Blocked forward substitution (sweep_kernels.hpp, k_blk_step) on two levels.  Panels of `pt` row tiles: ONE long launch
  // applies every finished row to the panel (K = all rows above it: the bulk of the flops, pt x ncp/128 tiles); inside the
  // panel, diagonal blocks of `bt` row tiles are solved one after the other, each followed by a short launch that applies it
  // to the panel's remaining rows.  Accuracy is set by bt alone (the height of the blocks that meet an explicit inverse),
  // speed by pt.  B is overwritten by the right-hand sides of the diagonal solves.
  const int bt = std::max(1, solve_block / TILE);
  const int pt = std::max(bt, (int)(solve_panel / TILE) / bt * bt);
  const double* Lf = A.d();
  const double* Li = Linv.d();
  auto step = [&](int u_r0, int u_rows, int u_k0, int u_k1, int s_r0, int s_rows) {
    hipLaunchKernelGGL(k_blk_step, dim3((unsigned)(ncp / TILE), (unsigned)(u_rows + s_rows)), dim3(256), GEMM_SMEM_BYTES,
                       stream, Lf, Li, Np, B, ldb, V, ldv, qp, ldq, u_r0, u_rows, u_k0, u_k1, s_r0, s_rows);
  };
  for (int p0 = 0; p0 < nb; p0 += pt) {
    const int p1 = std::min(nb, p0 + pt);
    if (p0 > 0) step(p0, p1 - p0, 0, p0, 0, 0);                    // panel rows -= L[., 0:p0] V[0:p0]
    for (int t0 = p0; t0 < p1; t0 += bt) {
      const int t1 = std::min(p1, t0 + bt);
      step(0, 0, 0, 0, t0, t1 - t0);                               // V[t0:t1] = inv(L_tt) B[t0:t1]
      if (t1 < p1) step(t1, p1 - t1, t0, t1, 0, 0);                // the panel's remaining rows -= L[., t0:t1] V[t0:t1]
    }
  }
}

// Z-side quantities of the sweep: ZsT, kXZ, V_Z = Linv kXZ, base_z = kself - |V_Z[:,z]|^2 and - for the score gradients
// only (need_w) - W_Z = Linv^T V_Z
void bobe_gp::prepare_z(const double* Z, int64_t M, int64_t Mp, bool need_w) {
  const bool host_z = !is_device_ptr(Z);
  const bool few = (Mp / TILE) * nb < 2 * std::max(num_cus, 1);
  auto make_w = [&]() {
    if (few)
      hipLaunchKernelGGL(k_trimul_t64, dim3((unsigned)(Mp / 64), (unsigned)(2 * nb)), dim3(256), GEMM64_SMEM_BYTES, stream,
                         (const double*)Linv.d(), Np, 2 * nb, (const double*)VZ.d(), Mp, WZ.d(), Mp);
    else
      hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(Mp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                         (const double*)Linv.d(), Np, nb, (const double*)VZ.d(), Mp, WZ.d(), Mp);
    LAUNCH_CHECK();
    wz_ready = true;
  };
  if (host_z && z_seen_m == M && std::memcmp(z_seen.data(), Z, (size_t)M * d * sizeof(double)) == 0) {
    if (need_w && !wz_ready) make_w();
    return;
  }
  forget_z();
  if (host_z) z_seen.assign(Z, Z + (size_t)M * d);
  const double* zin = fetch(Z, (size_t)M * d, z_stage);
  ZsT.ensure((size_t)d * Mp * sizeof(double));
  kXZ.ensure((size_t)Np * Mp * sizeof(double));
  VZ.ensure((size_t)Np * Mp * sizeof(double));
  WZ.ensure((size_t)Np * Mp * sizeof(double));
  basez.ensure((size_t)Mp * sizeof(double));
  qpart.ensure((size_t)2 * nb * (Mp > chunk ? Mp : chunk) * sizeof(double));   // (up to 2 nb row tiles of 64)
  scale(zin, M, Mp, hyp, ZsT.d(), Mp);
  kernel_matrix_cross(XsT.d(), Np, N, Np, ZsT.d(), Mp, M, Mp, hyp, kXZ.d(), Mp);
  // few integration points: 64 x 64 tiles (8 x 2 nb of them at M = 512) fill the chip where 4 x nb tiles of 128 x 128 do not
  if (refine_v) {
    solve_v(kXZ.d(), Mp, Mp, VZ.d(), Mp, qpart.d(), Mp);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), Mp, nb, Mp, hyp.kvar + hyp.noise, 0, basez.d(), (double*)nullptr);
  } else if (few) {
    const int nt = 2 * nb;
    hipLaunchKernelGGL(k_trimul_v64, dim3((unsigned)(Mp / 64), (unsigned)nt), dim3(256), GEMM64_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nt, (const double*)kXZ.d(), Mp, VZ.d(), Mp, qpart.d(), Mp);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), Mp, nt, Mp, hyp.kvar + hyp.noise, 0, basez.d(), (double*)nullptr);
  } else {
    hipLaunchKernelGGL(k_trimul, dim3((unsigned)(Mp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)kXZ.d(), Mp, VZ.d(), Mp, qpart.d(), Mp,
                       (const double*)nullptr, (int64_t)0, 0, (double*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), Mp, nb, Mp, hyp.kvar + hyp.noise, 0, basez.d(), (double*)nullptr);
  }
  LAUNCH_CHECK();
  if (need_w) make_w();
  if (host_z) z_seen_m = M;
}

void bobe_gp::sweep(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv,
                    double* wipstd, double* mean, double* var, int policy, int64_t* argmin_v, double* min_v,
                    int64_t* argmin_s, double* min_s, double* fantasy_out, bool gated) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (C <= 0) throw Err(BOBE_ERR_ARG, "C must be positive");
  const bool do_wip = (Z != nullptr);
  if (do_wip && M <= 0) throw Err(BOBE_ERR_ARG, "M must be positive");
  const int64_t Mp = do_wip ? round_up(M, TILE) : 0;
  const int nzt = (int)(Mp / TILE);
  const double kself = hyp.kvar + hyp.noise;
  const bool need_v = wipv || argmin_v || min_v;
  const bool need_s = wipstd || argmin_s || min_s;
  const double* cin = fetch(cand, (size_t)C * d, in_stage);
  if (do_wip) prepare_z(Z, M, Mp, false);
  // (the substitution path takes wider chunks - BOBE_SOLVE_CHUNK, speed only - unless the caller has set the chunk)
  const int64_t CH = (refine_v && solve_chunk > 0 && !chunk_set) ? solve_chunk : chunk;
  // The reference sweeps the integration points themselves (acquisition.py:394: candidates = mc_points).  Then K(X, C) and
  // V_C = L^-1 K(X, C) are what prepare_z has just made for Z: no second assembly and solve, the column sums come from V_Z in
  // the association the candidates' solve would have used (k_colsq_tile_parts: same bits as the long way).
  bool cand_is_z = false;
  if (do_wip && C == M && C <= CH && !mean && !var) {
    if (is_device_ptr(cand) || is_device_ptr(Z)) cand_is_z = (cand == Z);
    else cand_is_z = std::memcmp(cand, Z, (size_t)C * d * sizeof(double)) == 0;
  }
  // scoring runs once per super-chunk of SC candidates (bounded crossT workspace: Mp x SC doubles)
  const int64_t SC = round_up(std::min<int64_t>(C, std::max<int64_t>(CH, 65536)), CH);
  CsT.ensure((size_t)d * SC * sizeof(double));
  if (!cand_is_z) kXC.ensure((size_t)Np * CH * sizeof(double));
  sc.ensure((size_t)SC * sizeof(double));
  qpart.ensure((size_t)nb * (Mp > CH ? Mp : CH) * sizeof(double));
  part.ensure((size_t)nb * (Np > CH ? Np : CH) * sizeof(double));
  if (do_wip) pv.ensure((size_t)Mp * SC * sizeof(double));                   // crossT
  if ((do_wip || refine_v) && !cand_is_z) vxc.ensure((size_t)Np * CH * sizeof(double));      // V = Linv K(X, chunk)
  // The cross-covariance tiles of a chunk ride in the launch that solves the NEXT chunk (k_trimul), so V alternates
  // between two buffers; the last chunk of a super-chunk gets a launch of its own (k_cross_vv).  The blocked substitution of
  // an ill-conditioned factor (solve_v) is followed by a separate cross launch per chunk.
  const bool fuse_cross = do_wip && !refine_v && C > CH;
  if (fuse_cross) vxc2.ensure((size_t)Np * CH * sizeof(double));
  double* vbuf[2] = {vxc.d(), fuse_cross ? vxc2.d() : vxc.d()};
  int vsel = 0;
  struct { bool valid; const double* V; int64_t ncp; double* cross; } pend = {false, nullptr, 0, nullptr};
  auto cross_alone = [&](const double* Vc, int64_t ncp_, double* cross_out, int64_t ldvc) {
    // cross-covariances from the two solved factors (sweep_kernels.hpp, k_cross_vv): crossT[z][c] = VZ[:, z] . V[:, c]
    prof_begin(BOBE_PROF_CROSSVV);
    if ((int64_t)nzt * (ncp_ / TILE) >= 2 * std::max(num_cus, 1))
      hipLaunchKernelGGL(k_cross_vv<128>, dim3((unsigned)(ncp_ / TILE), (unsigned)nzt), dim3(256), GEMM_SMEM_BYTES, stream,
                         (const double*)VZ.d(), Mp, Vc, ldvc, Np, cross_out, SC);
    else
      hipLaunchKernelGGL(k_cross_vv<64>, dim3((unsigned)(ncp_ / 64), (unsigned)(Mp / 64)), dim3(256), GEMM64_SMEM_BYTES,
                         stream, (const double*)VZ.d(), Mp, Vc, ldvc, Np, cross_out, SC);
    prof_end(BOBE_PROF_CROSSVV);
  };
  double* d_mean = out_dev(mean, C, o_mean);
  double* d_var = out_dev(var, C, o_var);
  double* d_wipv = nullptr;
  double* d_wipstd = nullptr;
  if (do_wip) {
    if (need_v) {
      if (wipv) d_wipv = out_dev(wipv, C, o_wipv);
      else { o_wipv.ensure(C * sizeof(double)); d_wipv = o_wipv.d(); }
    }
    if (need_s) {
      if (wipstd) d_wipstd = out_dev(wipstd, C, o_wipstd);
      else { o_wipstd.ensure(C * sizeof(double)); d_wipstd = o_wipstd.d(); }
    }
  }
  double* d_fant = nullptr;
  if (fantasy_out) {   // dumped with leading dimension M (dense), C x M
    d_fant = is_device_ptr(fantasy_out) ? fantasy_out : (kout.ensure((size_t)C * M * sizeof(double)), kout.d());
  }
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  for (int64_t s0 = 0; s0 < C; s0 += SC) {
    const int64_t ns = (C - s0 < SC) ? (C - s0) : SC;
    const int64_t nsp = round_up(ns, TILE);
    scale(cin + s0 * d, ns, nsp, hyp, CsT.d(), SC);
    for (int64_t c0 = 0; c0 < ns; c0 += CH) {
      const int64_t nc = (ns - c0 < CH) ? (ns - c0) : CH;
      const int64_t ncp = round_up(nc, TILE);
      if (cand_is_z) {                // (one chunk: C <= CH) V_C = V_Z, s_c from its column sums, the cross tiles V_Z^T V_Z
        hipLaunchKernelGGL(k_colsq_tile_parts, dim3((unsigned)((ncp + 255) / 256), (unsigned)nb), dim3(256), 0, stream,
                           (const double*)VZ.d(), Mp, ncp, qpart.d(), CH);
        cross_alone(VZ.d(), ncp, pv.d() + c0, Mp);
        hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                           (const double*)qpart.d(), CH, nb, nc, kself, policy, sc.d() + c0, (double*)nullptr);
        LAUNCH_CHECK();
        continue;
      }
      // (posterior mean: the assembly leaves K(X, chunk)^T alpha per row tile on the way, k_gemv_t_part's partial sums)
      prof_begin(BOBE_PROF_KXC);
      kernel_matrix_cross(XsT.d(), Np, N, Np, CsT.d() + c0, SC, nc, ncp, hyp, kXC.d(), CH,
                          d_mean ? (const double*)alpha.d() : nullptr, d_mean ? part.d() : nullptr, CH);
      prof_end(BOBE_PROF_KXC);
      if (d_mean) {
        hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                           (const double*)part.d(), CH, nb, 0, nc, d_mean + s0 + c0);
      }
      if (do_wip || d_var) {       // (a mean-only prediction - nested sampling's likelihood calls - needs no triangular product)
        double* vcur = vbuf[vsel];
        prof_begin(BOBE_PROF_TRIMUL);
        if (fuse_cross) {
          const int ncv = (int)(ncp / TILE), ncx = pend.valid ? (int)(pend.ncp / TILE) : 0;
          const int nz = pend.valid ? nzt : 0;
          hipLaunchKernelGGL(k_trimul, dim3((unsigned)std::max(ncv, ncx), (unsigned)(nb + nz)), dim3(256), GEMM_SMEM_BYTES,
                             stream, (const double*)Linv.d(), Np, nb, (const double*)kXC.d(), CH, vcur, CH, qpart.d(), CH,
                             (const double*)VZ.d(), Mp, nz, pend.cross, SC, pend.V, CH, ncx, ncv);
          pend = {true, vcur, ncp, pv.d() + c0};
          vsel ^= 1;
        } else {
          solve_v(kXC.d(), CH, ncp, (do_wip || refine_v) ? vcur : nullptr, CH, qpart.d(), CH);
        }
        prof_end(BOBE_PROF_TRIMUL);
        if (do_wip && !fuse_cross) cross_alone(vcur, ncp, pv.d() + c0, CH);
        // s_c for the scorer, var for the caller
        hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                           (const double*)qpart.d(), CH, nb, nc, kself, policy, sc.d() + c0,
                           d_var ? d_var + s0 + c0 : nullptr);
      }
      LAUNCH_CHECK();
    }
    if (pend.valid) {                 // the super-chunk's last chunk: no next launch to ride in
      cross_alone(pend.V, pend.ncp, pend.cross, CH);
      pend.valid = false;
    }
    if (do_wip) {
      const dim3 grid((unsigned)((ns + 63) / 64));
      const size_t sm = (size_t)(d + 1) * 128 * sizeof(double);
      double* vo = d_fant ? d_fant + s0 * M : nullptr;
      prof_begin(BOBE_PROF_CROSS);
#define WS(KE, DC)                                                                                              \
  hipLaunchKernelGGL((k_wip_score<KE, DC>), grid, dim3(256), sm, stream, (const double*)pv.d(), SC,             \
                     (const double*)CsT.d(), SC, (const double*)ZsT.d(), Mp, M, (const double*)sc.d(),          \
                     (const double*)basez.d(), ns, hyp, y_std * y_std, d_wipv ? d_wipv + s0 : nullptr,          \
                     d_wipstd ? d_wipstd + s0 : nullptr, vo, M)
      if (hyp.kern == 0) {
        if (dcap == 8) WS(0, 8); else if (dcap == 16) WS(0, 16); else WS(0, 32);
      } else {
        if (dcap == 8) WS(1, 8); else if (dcap == 16) WS(1, 16); else WS(1, 32);
      }
#undef WS
      prof_end(BOBE_PROF_CROSS);
      LAUNCH_CHECK();
    }
  }
  // the classifier gate of the predict family (clf_gp.py:173-205): gated points get mean = -inf, var = 1e-12
  if (gated && gate.n_sv > 0 && (d_mean || d_var)) gate_apply(cin, C, nullptr, nullptr, d_mean, d_var, nullptr, nullptr);
  o_misc.ensure(8 * sizeof(double));
  double* m_val = o_misc.d();                                        // [0],[1]
  int64_t* m_idx = reinterpret_cast<int64_t*>(o_misc.d() + 2);       // [2],[3]
  const bool want_v = do_wip && (argmin_v || min_v);
  const bool want_s = do_wip && (argmin_s || min_s);
  if (want_v) hipLaunchKernelGGL(k_argmin, dim3(1), dim3(1024), 0, stream, (const double*)d_wipv, C, m_val, m_idx);
  if (want_s)
    hipLaunchKernelGGL(k_argmin, dim3(1), dim3(1024), 0, stream, (const double*)d_wipstd, C, m_val + 1, m_idx + 1);
  LAUNCH_CHECK();
  out_finish(mean, C, o_mean);
  out_finish(var, C, o_var);
  if (do_wip) {
    out_finish(wipv, C, o_wipv);
    out_finish(wipstd, C, o_wipstd);
  }
  if (fantasy_out && !is_device_ptr(fantasy_out))
    HIPCHK(hipMemcpyAsync(fantasy_out, d_fant, (size_t)C * M * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (want_v || want_s) {
    HIPCHK(hipMemcpyAsync(h_res, o_misc.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    const int64_t* hi = reinterpret_cast<const int64_t*>(h_res + 2);
    if (want_v) {
      if (argmin_v) *argmin_v = hi[0];
      if (min_v) *min_v = h_res[0];
    }
    if (want_s) {
      if (argmin_s) *argmin_s = hi[1];
      if (min_s) *min_s = h_res[1];
    }
  } else {
    sync();
  }
}

void bobe_gp::wip_grad(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv,
                       double* wipstd, double* dwipv, double* dwipstd) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (C <= 0 || M <= 0) throw Err(BOBE_ERR_ARG, "C and M must be positive");
  use();
  const int64_t Mp = round_up(M, TILE), CH = std::min<int64_t>(chunk, 1024);
  const double kself = hyp.kvar + hyp.noise;
  const bool few_path = C <= 16 && N <= 4096;
  // (the few-candidate path reads host coordinates through the pinned block h_in: no copy command in front of its chain)
  const bool cand_pinned = few_path && !is_device_ptr(cand);
  const double* cin = cand_pinned ? nullptr : fetch(cand, (size_t)C * d, in_stage);
  prepare_z(Z, M, Mp, true);                             // ZsT, V_Z, W_Z = K^-1 K(X,Z), base_z
  CsT.ensure((size_t)d * std::max<int64_t>(CH, chunk) * sizeof(double));
  kXC.ensure((size_t)Np * std::max<int64_t>(CH, chunk) * sizeof(double));
  pv.ensure((size_t)Np * CH * sizeof(double));           // V = Linv k_c
  ps.ensure((size_t)Np * CH * sizeof(double));           // U = K^-1 k_c
  qpart.ensure((size_t)nb * std::max<int64_t>(std::max<int64_t>(CH, chunk), Mp) * sizeof(double));
  sc.ensure((size_t)std::max<int64_t>(CH, chunk) * sizeof(double));
  double* d_v = out_dev(wipv, C, o_wipv);
  double* d_s = out_dev(wipstd, C, o_wipstd);
  double* d_dv = out_dev(dwipv, (size_t)C * d, o_mean);
  double* d_ds = out_dev(dwipstd, (size_t)C * d, o_var);
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  if (few_path) {
    double* cdev = nullptr;                        // device copy of pinned-host coordinates, written by the first stage
    if (cand_pinned) {
      if (!h_in) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_in), 16 * MAX_D * sizeof(double), hipHostMallocDefault));
      std::memcpy(h_in, cand, (size_t)C * d * sizeof(double));
      in_stage.ensure((size_t)16 * MAX_D * sizeof(double));
      cdev = in_stage.d();
    }
    const double* cfirst = cand_pinned ? (const double*)h_in : cin;   // what the first stage reads
    if (cand_pinned) cin = cdev;                                      // what the later stages read
    // A handful of candidates (the L-BFGS refinement sends one): matrix-vector stages spread over the chip instead of
    // 128-column tile passes and one workgroup per candidate (kernels.hpp, "the same for a HANDFUL of candidates").
    const int nzw = (int)(Mp / 64), nnw = (int)((N + WG_ROWS - 1) / WG_ROWS);
    const size_t n_vec = (size_t)C * Np, n_a = (size_t)C * Mp;
    wg_ws.ensure((6 * n_vec + 2 * n_a + (size_t)C * nzw * WG_ZS + (size_t)C * nnw * WG_NS) * sizeof(double));
    part.ensure((size_t)C * nb * Np * sizeof(double));
    double* kc = wg_ws.d();
    double* vv = kc + n_vec;
    double* uu = vv + n_vec;
    double* a1 = uu + n_vec;
    double* b1 = a1 + n_a;
    double* pz = b1 + n_a;
    double* pn = pz + (size_t)C * nzw * WG_ZS;
    double* t1 = pn + (size_t)C * nnw * WG_NS;     // three more vectors per candidate: the refinement step's temporaries
    double* t2 = t1 + n_vec;
    double* t3 = t2 + n_vec;
    const Hyper& h = hyp;
    const double* li = Linv.d();
#define FEW(KE, DC)                                                                                                      \
  do {                                                                                                                   \
    hipLaunchKernelGGL((k_wg_col<KE, DC>), dim3((unsigned)(Np / 256 + 1), (unsigned)C), dim3(256), 0, stream,           \
                       (const double*)XsT.d(), Np, N, Np, cfirst, h, kc, cdev);                                        \
    solve_alpha(li, vv, uu, part.d(), (int)C, 0, Np, (int64_t)nb * Np, (const double*)kc, Np);                             \
    if (refine_v) {   /* one step of iterative refinement in vector form: v += Linv (k - L v), u += Linv^T of the same */   \
      const unsigned gv_ = (unsigned)((n_vec + 255) / 256);                                                              \
      hipLaunchKernelGGL(k_gemv_lower, dim3((unsigned)(Np / 4), (unsigned)C), dim3(256), 0, stream, (const double*)A.d(), \
                         Np, Np, (const double*)vv, t1, (int64_t)0, Np, Np);                                             \
      hipLaunchKernelGGL(k_vec_axpy, dim3(gv_), dim3(256), 0, stream, t1, (const double*)kc, (const double*)t1, -1.0,    \
                         (int64_t)n_vec);                                                                                \
      solve_alpha(li, t2, t3, part.d(), (int)C, 0, Np, (int64_t)nb * Np, (const double*)t1, Np);                         \
      hipLaunchKernelGGL(k_vec_axpy, dim3(gv_), dim3(256), 0, stream, vv, (const double*)vv, (const double*)t2, 1.0,     \
                         (int64_t)n_vec);                                                                                \
      hipLaunchKernelGGL(k_vec_axpy, dim3(gv_), dim3(256), 0, stream, uu, (const double*)uu, (const double*)t3, 1.0,     \
                         (int64_t)n_vec);                                                                                \
    }                                                                                                                    \
    hipLaunchKernelGGL((k_wg_cross<KE, DC>), dim3((unsigned)nzw, (unsigned)C), dim3(256), (size_t)N * sizeof(double),     \
                       stream, (const double*)ZsT.d(), Mp, M, (const double*)VZ.d(), Mp, N, Np,                  \
                       (const double*)kc, (const double*)vv, cin, h, kself, (const double*)basez.d(), y_std * y_std,   \
                       a1, b1, Mp, pz);                                                                                  \
    hipLaunchKernelGGL((k_wg_rows<KE, DC>), dim3((unsigned)nnw, (unsigned)C), dim3(256), 0, stream,                    \
                       (const double*)XsT.d(), Np, N, Np, cin, h, (const double*)WZ.d(), Mp, Mp,                    \
                       (const double*)a1, (const double*)b1, Mp, (const double*)uu, pn);                                  \
  } while (0)
    if (h.kern == 0) {
      if (dcap == 8) FEW(0, 8); else if (dcap == 16) FEW(0, 16); else FEW(0, 32);
    } else {
      if (dcap == 8) FEW(1, 8); else if (dcap == 16) FEW(1, 16); else FEW(1, 32);
    }
#undef FEW
    const size_t n_out = (size_t)C * (2 + 2 * d);
    const bool packed = n_out <= 96 && !(wipv && is_device_ptr(wipv)) && !(wipstd && is_device_ptr(wipstd)) &&
                        !(dwipv && is_device_ptr(dwipv)) && !(dwipstd && is_device_ptr(dwipstd));
    if (packed) {                                  // the last stage writes the pinned result block itself: no copy command
      double* ob = h_res;
      hipLaunchKernelGGL(k_wg_final, dim3((unsigned)C), dim3(256), 0, stream, (const double*)pz, nzw, (const double*)pn,
                         nnw, h, M, ob, ob + C, ob + 2 * C, ob + 2 * C + C * d);
      LAUNCH_CHECK();
      sync();
      const double* hr = h_res;
      if (wipv) std::memcpy(wipv, hr, (size_t)C * sizeof(double));
      if (wipstd) std::memcpy(wipstd, hr + C, (size_t)C * sizeof(double));
      if (dwipv) std::memcpy(dwipv, hr + 2 * C, (size_t)C * d * sizeof(double));
      if (dwipstd) std::memcpy(dwipstd, hr + 2 * C + C * d, (size_t)C * d * sizeof(double));
      return;
    }
    hipLaunchKernelGGL(k_wg_final, dim3((unsigned)C), dim3(256), 0, stream, (const double*)pz, nzw, (const double*)pn,
                       nnw, h, M, d_v, d_s, d_dv, d_ds);
    LAUNCH_CHECK();
    out_finish(wipv, C, o_wipv);
    out_finish(wipstd, C, o_wipstd);
    out_finish(dwipv, (size_t)C * d, o_mean);
    out_finish(dwipstd, (size_t)C * d, o_var);
    sync();
    return;
  }
  for (int64_t c0 = 0; c0 < C; c0 += CH) {
    const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
    scale(cin + c0 * d, nc, ncp, hyp, CsT.d(), CH);
    kernel_matrix_cross(XsT.d(), Np, N, Np, CsT.d(), CH, nc, ncp, hyp, kXC.d(), CH);
    solve_v(kXC.d(), CH, ncp, pv.d(), CH, qpart.d(), CH);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), CH, nb, nc, kself, 1, sc.d(), (double*)nullptr);
    hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)pv.d(), CH, ps.d(), CH);
#define WG(KE, DC)                                                                                                   \
  hipLaunchKernelGGL((k_wip_grad<KE, DC>), dim3((unsigned)nc), dim3(256), 0, stream, (const double*)XsT.d(), Np,  \
                     N, (const double*)CsT.d(), CH, (const double*)ZsT.d(), Mp, M, (const double*)WZ.d(),  \
                     Mp, (const double*)ps.d(), CH, (const double*)VZ.d(), (const double*)pv.d(), CH,         \
                     (const double*)sc.d(), (const double*)basez.d(), hyp,                                        \
                     y_std * y_std, d_v ? d_v + c0 : nullptr, d_s ? d_s + c0 : nullptr, d_dv ? d_dv + c0 * d : nullptr, \
                     d_ds ? d_ds + c0 * d : nullptr)
    if (hyp.kern == 0) {
      if (dcap == 8) WG(0, 8); else if (dcap == 16) WG(0, 16); else WG(0, 32);
    } else {
      if (dcap == 8) WG(1, 8); else if (dcap == 16) WG(1, 16); else WG(1, 32);
    }
#undef WG
    LAUNCH_CHECK();
  }
  out_finish(wipv, C, o_wipv);
  out_finish(wipstd, C, o_wipstd);
  out_finish(dwipv, (size_t)C * d, o_mean);
  out_finish(dwipstd, (size_t)C * d, o_var);
  sync();
}

void bobe_gp::predict_grad(const double* Xq, int64_t C, double* mean, double* var, double* dmean, double* dvar) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (C <= 0) throw Err(BOBE_ERR_ARG, "C must be positive");
  use();
  const int64_t CH = std::min<int64_t>(chunk, 2048);
  const double kself = hyp.kvar + hyp.noise;
  const double* cin = fetch(Xq, (size_t)C * d, in_stage);
  if (!dvar) {
    // mean-only mode (HMC on the surrogate): scale the queries, then one kernel that walks the training points
    // and accumulates the mean and its gradient - no K(X, C), no triangular products
    CsT.ensure((size_t)d * std::max<int64_t>(CH, chunk) * sizeof(double));
    double* d_mean = out_dev(mean, C, o_mean);
    double* d_dm = out_dev(dmean, (size_t)C * d, o_wipv);
    const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
    for (int64_t c0 = 0; c0 < C; c0 += CH) {
      const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
      scale(cin + c0 * d, nc, ncp, hyp, CsT.d(), CH);
      const dim3 grid((unsigned)((nc + 63) / 64));
      const size_t sm = (size_t)(d + 1) * 128 * sizeof(double);
#define PGM(KE, DC)                                                                                                  \
  hipLaunchKernelGGL((k_predict_grad<KE, DC>), grid, dim3(256), sm, stream, (const double*)XsT.d(), Np, N,    \
                     (const double*)CsT.d(), CH, nc, (const double*)alpha.d(), (const double*)nullptr,           \
                     (int64_t)0, (const double*)nullptr, hyp, d_dm + c0 * d, (double*)nullptr,                     \
                     d_mean ? d_mean + c0 : nullptr)
      if (hyp.kern == 0) {
        if (dcap == 8) PGM(0, 8); else if (dcap == 16) PGM(0, 16); else PGM(0, 32);
      } else {
        if (dcap == 8) PGM(1, 8); else if (dcap == 16) PGM(1, 16); else PGM(1, 32);
      }
#undef PGM
      LAUNCH_CHECK();
    }
    // (classifier gate, clf_gp.py:173-205: gated points carry mean = -inf and a zero gradient)
    if (gate.n_sv > 0) gate_apply(cin, C, nullptr, nullptr, d_mean, nullptr, d_dm, nullptr);
    out_finish(mean, C, o_mean);
    out_finish(dmean, (size_t)C * d, o_wipv);
    sync();
    return;
  }
  CsT.ensure((size_t)d * std::max<int64_t>(CH, chunk) * sizeof(double));
  kXC.ensure((size_t)Np * std::max<int64_t>(CH, chunk) * sizeof(double));
  forget_z();                                      // (VZ / WZ double as this call's scratch)
  VZ.ensure((size_t)Np * CH * sizeof(double));     // V = Linv k
  WZ.ensure((size_t)Np * CH * sizeof(double));     // U = Linv^T V = K^-1 k
  qpart.ensure((size_t)nb * std::max<int64_t>(CH, chunk) * sizeof(double));
  part.ensure((size_t)nb * std::max<int64_t>(Np, std::max<int64_t>(CH, chunk)) * sizeof(double));
  sc.ensure((size_t)std::max<int64_t>(CH, chunk) * sizeof(double));
  double* d_mean = out_dev(mean, C, o_mean);
  double* d_var = out_dev(var, C, o_var);
  double* d_dm = out_dev(dmean, (size_t)C * d, o_wipv);
  double* d_dv = out_dev(dvar, (size_t)C * d, o_wipstd);
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  for (int64_t c0 = 0; c0 < C; c0 += CH) {
    const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
    scale(cin + c0 * d, nc, ncp, hyp, CsT.d(), CH);
    kernel_matrix_cross(XsT.d(), Np, N, Np, CsT.d(), CH, nc, ncp, hyp, kXC.d(), CH,
                           d_mean ? (const double*)alpha.d() : nullptr, d_mean ? part.d() : nullptr, CH);
    if (d_mean) {
      hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                         (const double*)part.d(), CH, nb, 0, nc, d_mean + c0);
    }
    solve_v(kXC.d(), CH, ncp, VZ.d(), CH, qpart.d(), CH);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), CH, nb, nc, kself, 1, sc.d(), d_var ? d_var + c0 : nullptr);
    hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)VZ.d(), CH, WZ.d(), CH);
    const dim3 grid((unsigned)((nc + 63) / 64));
    const size_t sm = (size_t)(d + 1) * 128 * sizeof(double);
#define PG(KE, DC)                                                                                                  \
  hipLaunchKernelGGL((k_predict_grad<KE, DC>), grid, dim3(256), sm, stream, (const double*)XsT.d(), Np, N,    \
                     (const double*)CsT.d(), CH, nc, (const double*)alpha.d(), (const double*)WZ.d(), CH,     \
                     (const double*)sc.d(), hyp, d_dm + c0 * d, d_dv + c0 * d)
    if (hyp.kern == 0) {
      if (dcap == 8) PG(0, 8); else if (dcap == 16) PG(0, 16); else PG(0, 32);
    } else {
      if (dcap == 8) PG(1, 8); else if (dcap == 16) PG(1, 16); else PG(1, 32);
    }
#undef PG
    LAUNCH_CHECK();
  }
  if (gate.n_sv > 0) gate_apply(cin, C, nullptr, nullptr, d_mean, d_var, d_dm, d_dv);
  out_finish(mean, C, o_mean);
  out_finish(var, C, o_var);
  out_finish(dmean, (size_t)C * d, o_wipv);
  out_finish(dvar, (size_t)C * d, o_wipstd);
  sync();
}

int bobe_gp::append(const double* X_new, int64_t b, const double* y_all) {
  if (!factored || not_pd) throw Err(BOBE_ERR_STATE, "append needs a positive-definite factorised state");
  use();
  sync();
  const int64_t N0 = N, N1 = N0 + b, Np0 = Np, Np1 = round_up(N1, TILE);
  // The handle is rebuilt in stages (X, the padded frame, y, then the new rows).  Until the last stage is through it
  // counts as holding nothing: an error on the way (out of memory in the new frame, a failed launch) leaves a handle
  // that every later call refuses ("call bobe_gp_set_data first") instead of one with N0 points' factor under N1
  // points' data; bobe_gp_set_data + bobe_gp_factor then rebuild it from scratch (what GP.update falls back to).
  factored = false;
  have_data = false;
  forget_z();
  DBuf nx, oa, ol;
  try {
  // ---- training data: X gains b rows, every y changes (the caller re-standardised them, gp.py:520-536)
  const bool xnew_on_device = is_device_ptr(X_new);
  {
    // (X keeps room for the rows up to the next multiple of 128 and one block more: a hipMalloc / hipFree pair per call
    // was a third of an append at N = 600; the new rows land behind the old ones, in stream order)
    const size_t need = (size_t)N1 * d * sizeof(double);
    if (X.bytes < need) {
      nx.ensure((size_t)(Np1 + TILE) * d * sizeof(double));
      HIPCHK(hipMemcpyAsync(nx.p, X.p, (size_t)N0 * d * sizeof(double), hipMemcpyDeviceToDevice, stream));
      sync();
      std::swap(X, nx);
      nx.release();
    }
    HIPCHK(hipMemcpyAsync(X.d() + N0 * d, X_new, (size_t)b * d * sizeof(double),
                          xnew_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  }
  // ---- a larger padded size: move L and Linv into the new [[., 0], [0, I]] frame
  if (Np1 != Np0) {
    std::swap(oa, A);
    std::swap(ol, Linv);
    Np = Np1;
    nb = (int)(Np1 / TILE);
    alloc_for_n();                      // A, Linv (fresh), scratch, probs for the new block count
    hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np1 + 255) / 256), (unsigned)Np1), dim3(256), 0, stream,
                       (const double*)oa.p, (int64_t)0, A.d(), Np1, Np1);         // identity everywhere ...
    hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np1 + 255) / 256), (unsigned)Np1), dim3(256), 0, stream,
                       (const double*)ol.p, (int64_t)0, Linv.d(), Np1, Np1);
    HIPCHK(hipMemcpy2DAsync(A.p, (size_t)Np1 * 8, oa.p, (size_t)Np0 * 8, (size_t)Np0 * 8, (size_t)Np0,
                            hipMemcpyDeviceToDevice, stream));                     // ... then the old frame on top
    HIPCHK(hipMemcpy2DAsync(Linv.p, (size_t)Np1 * 8, ol.p, (size_t)Np0 * 8, (size_t)Np0 * 8, (size_t)Np0,
                            hipMemcpyDeviceToDevice, stream));
    sync();
    oa.release();
    ol.release();
  }
  N = N0;                               // (old point count while the cross-covariances are assembled)
  HIPCHK(hipMemsetAsync(y.p, 0, (size_t)Np * sizeof(double), stream));
  HIPCHK(hipMemcpyAsync(y.p, y_all, (size_t)N1 * sizeof(double),
                        is_device_ptr(y_all) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  // ---- V = Linv K(X_old, X_new), W = Linv^T V, S = K(X_new, X_new) + noise I - V^T V
  const int64_t bp = TILE;
  kXC.ensure((size_t)Np * std::max<int64_t>(bp, chunk) * sizeof(double));
  forget_z();
  VZ.ensure((size_t)Np * bp * sizeof(double));
  WZ.ensure((size_t)Np * bp * sizeof(double));
  kin_a.ensure((size_t)d * bp * sizeof(double));
  qpart.ensure((size_t)nb * std::max<int64_t>(bp, chunk) * sizeof(double));
  o_misc.ensure((size_t)(3 * 64 * 64 + 8) * sizeof(double));
  scale(X.d(), N0, Np, hyp, XsT.d(), Np);                                    // old points only (rest 0)
  scale(X.d() + N0 * d, b, bp, hyp, kin_a.d(), bp);
  kernel_matrix_cross(XsT.d(), Np, N0, Np, kin_a.d(), bp, b, bp, hyp, kXC.d(), bp);
  solve_v(kXC.d(), bp, bp, VZ.d(), bp, nullptr, 0);
  hipLaunchKernelGGL(k_trimul_t, dim3(1, (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream, (const double*)Linv.d(), Np,
                     nb, (const double*)VZ.d(), bp, WZ.d(), bp);
  double* G = o_misc.d();                                                             // b*b Gram matrix V^T V
  hipLaunchKernelGGL(k_gram_small, dim3((unsigned)b, (unsigned)b), dim3(256), 0, stream, (const double*)VZ.d(), bp, N0,
                     (int)b, G);
  LAUNCH_CHECK();
  std::vector<double> hG((size_t)b * b), hK((size_t)b * b), hx((size_t)b * d);
  HIPCHK(hipMemcpyAsync(hG.data(), G, hG.size() * 8, hipMemcpyDeviceToHost, stream));
  if (xnew_on_device) HIPCHK(hipMemcpyAsync(hx.data(), X.d() + N0 * d, hx.size() * 8, hipMemcpyDeviceToHost, stream));
  else std::memcpy(hx.data(), X_new, hx.size() * 8);
  sync();
  // K(X_new, X_new) + noise I on the host (b <= 64 points; the kernel of gp.py:124-168 with direct differences)
  for (int64_t i = 0; i < b; ++i)
    for (int64_t j = 0; j < b; ++j) {
      double r2 = 0.0;
      for (int q = 0; q < d; ++q) {
        const double df = hx[i * d + q] / hyp.ls[q] - hx[j * d + q] / hyp.ls[q];
        r2 += df * df;
      }
      double kv;
      if (kern == 0) {
        kv = hyp.kvar * std::exp(-0.5 * r2);
      } else {
        const double dd = std::sqrt(r2 < 1e-30 ? 1e-30 : r2);
        kv = hyp.kvar * (1.0 + dd * (SQRT5 + (dd * 5.0) / 3.0)) * std::exp(-SQRT5 * dd);
      }
      hK[i * b + j] = kv + (i == j ? hyp.noise : 0.0) - hG[i * b + j];
    }
  // L22 = chol(S), L22inv by forward substitution; a pivot at or below the rank test's floor (pivot_floor; 0 at least) = the
  // appended matrix is not positive definite
  const double piv_floor = pivot_floor(hyp);
  std::vector<double> s22((size_t)2 * b * b, 0.0);
  double* L22 = s22.data();
  double* Li = s22.data() + b * b;
  bool pd = true;
  for (int64_t j = 0; j < b && pd; ++j) {
    double dj = hK[j * b + j];
    for (int64_t k = 0; k < j; ++k) dj -= L22[j * b + k] * L22[j * b + k];
    if (!(dj > 0.0) || dj < piv_floor) { pd = false; break; }
    L22[j * b + j] = std::sqrt(dj);
    for (int64_t i = j + 1; i < b; ++i) {
      double v = hK[i * b + j];
      for (int64_t k = 0; k < j; ++k) v -= L22[i * b + k] * L22[j * b + k];
      L22[i * b + j] = v / L22[j * b + j];
    }
  }
  N = N1;
  if (!pd) {                               // same outcome as the full refactorisation: NaN state, BOBE_NOT_PD
    have_data = true;
    return factor_state();
  }
  for (int64_t c = 0; c < b; ++c)
    for (int64_t i = c; i < b; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int64_t k = c; k < i; ++k) v -= L22[i * b + k] * Li[k * b + c];
      Li[i * b + c] = v / L22[i * b + i];
    }
  double* d22 = o_misc.d() + 64 * 64;
  HIPCHK(hipMemcpyAsync(d22, s22.data(), s22.size() * 8, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_append_rows, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, stream, A.d(), Linv.d(), Np,
                     N0, (int)b, (const double*)VZ.d(), (const double*)WZ.d(), bp, (const double*)d22);
  LAUNCH_CHECK();
  scale(X.d(), N1, Np, hyp, XsT.d(), Np);                                    // all points again
  solve_alpha(Linv.d(), w.d(), alpha.d(), part.d());                     // alpha = Linv^T Linv y
  decide_refinement(min_pivot_root());  // (the grown factor's smallest pivot; synchronises: s22 / hG are host temporaries)
  } catch (...) {
    nx.release();
    oa.release();
    ol.release();
    N = 0;
    Np = 0;                              // forces bobe_gp_set_data to size every buffer again
    nb = 0;
    throw;
  }
  have_data = true;
  factored = true;
  forget_z();
  not_pd = false;
  return BOBE_OK;
}
