// libbobe_gp.so, ABI unit: the extern "C" layer of include/bobe_gp.h over the member functions of struct bobe_gp
// (gp_factor.hip, gp_sweep.hip, gp_consumers.hip), the multi-GPU exchange step (a RCCL communicator owned by the library;
// librccl is opened on first use, so the library itself loads on hosts without it) and the test / bench hooks.
// Nothing throws or aborts across this boundary: every entry point returns a status and leaves its text in
// bobe_last_error().
#include "gp_handle.hpp"

#include "kernels_common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace bobe;

namespace bobe {
thread_local std::string g_err;
}

namespace {

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void k_debug_gemm(const double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ B, int64_t ldb, double* __restrict__ C,
                                                    int64_t ldc, int64_t K) {
  extern __shared__ double smem[];
  v4d acc[4][4];
  acc_zero(acc);
  gemm_tile<LA, LB>(acc, A, lda, (int64_t)blockIdx.y * TILE, B, ldb, (int64_t)blockIdx.x * TILE, 0, K, smem);
  store_tile(acc, C, ldc, (int64_t)blockIdx.y * TILE, (int64_t)blockIdx.x * TILE, 1.0, 0.0);
}

void configure_debug_kernels() {
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return;
  allow_big_lds(k_debug_gemm<0, 0>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<0, 1>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<1, 0>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<1, 1>, GEMM_SMEM_BYTES);
  done[dev] = true;
}

}  // namespace

#define API_BEGIN try {
#define API_END                      \
  }                                  \
  catch (const Err& e) {             \
    g_err = e.what();                \
    return e.code;                   \
  }                                  \
  catch (const std::exception& e) {  \
    g_err = e.what();                \
    return BOBE_ERR_HIP;             \
  }                                  \
  catch (...) {                      \
    g_err = "unknown error";         \
    return BOBE_ERR_HIP;             \
  }
#define NEED(cond, msg) \
  if (!(cond)) throw Err(BOBE_ERR_ARG, msg)

extern "C" {

const char* bobe_version(void) { return "bobe_gp 0.2.0 gfx950"; }
const char* bobe_last_error(void) { return g_err.c_str(); }

int bobe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int bobe_gp_create(bobe_gp_t** out, int device, int kernel, int d) {
  API_BEGIN
  NEED(out, "out is NULL");
  *out = nullptr;
  NEED(d >= 1 && d <= MAX_D, "d must be in [1, 32]");
  NEED(kernel == BOBE_KERNEL_RBF || kernel == BOBE_KERNEL_MATERN, "unknown kernel id");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) throw Err(BOBE_ERR_HIP, "no such HIP device (a MI355X is required; there is no CPU path)");
  bobe_gp* g = new bobe_gp();
  g->device = device;
  g->kern = kernel;
  g->d = d;
  g->hyp.d = d;
  g->hyp.kern = kernel;
  for (int j = 0; j < MAX_D; ++j) g->hyp.ls[j] = 1.0;
  g->hyp.kvar = 1.0;
  g->hyp.noise = 1e-8;
  try {
    g->use();
    g->slots.reserve(BOBE_MAX_MLL_SLOTS);      // bobe_gp_mll_wait reads it without the submit mutex: never reallocate
    HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->own_stream = true;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g->h_res), 128 * sizeof(double), hipHostMallocDefault));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    g->num_cus = prop.multiProcessorCount;
  } catch (...) {
    g->release_all();
    delete g;
    throw;
  }
  *out = g;
  return BOBE_OK;
  API_END
}

void bobe_gp_destroy(bobe_gp_t* g) {
  if (!g) return;
  g->release_all();
  delete g;
}

void* bobe_gp_get_stream(bobe_gp_t* g) { return g ? (void*)g->stream : nullptr; }

int bobe_gp_set_stream(bobe_gp_t* g, void* s) {
  API_BEGIN
  NEED(g, "gp is NULL");
  g->use();
  g->sync();
  if (g->own_stream && g->stream) HIPCHK(hipStreamDestroy(g->stream));
  g->stream = static_cast<hipStream_t>(s);
  g->own_stream = false;
  return BOBE_OK;
  API_END
}

int bobe_gp_sync(bobe_gp_t* g) {
  API_BEGIN
  NEED(g, "gp is NULL");
  g->use();
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_set_chunk(bobe_gp_t* g, int64_t chunk) {
  API_BEGIN
  NEED(g, "gp is NULL");
  const bool reset = chunk == 0;
  if (reset) chunk = 8192;
  NEED(chunk >= TILE && chunk % TILE == 0, "chunk must be a positive multiple of 128");
  g->use();
  g->sync();
  g->chunk = chunk;
  g->chunk_set = !reset;
  return BOBE_OK;
  API_END
}

int bobe_debug_solve_opts(bobe_gp_t* g, int panel, int64_t chunk) {
  API_BEGIN
  NEED(g, "gp is NULL");
  NEED(panel > 0 && panel % TILE == 0, "panel must be a positive multiple of 128");
  NEED(chunk >= 0 && chunk % TILE == 0, "chunk must be 0 or a positive multiple of 128");
  g->use();
  g->sync();
  g->solve_panel = panel;
  g->solve_chunk = chunk;
  return BOBE_OK;
  API_END
}

int64_t bobe_gp_npoints(bobe_gp_t* g) { return g ? g->N : 0; }

int bobe_gp_set_data(bobe_gp_t* g, const double* X, const double* ys, int64_t N) {
  API_BEGIN
  NEED(g && X && ys, "NULL argument");
  NEED(N >= 1, "N must be >= 1");
  g->set_data(X, ys, N);
  return BOBE_OK;
  API_END
}

int bobe_gp_set_hyper(bobe_gp_t* g, const double* ls, double kvar, double noise) {
  API_BEGIN
  NEED(g && ls, "NULL argument");
  for (int j = 0; j < g->d; ++j) g->hyp.ls[j] = ls[j];
  g->hyp.kvar = kvar;
  g->hyp.noise = noise;
  g->factored = false;
  g->forget_z();
  return BOBE_OK;
  API_END
}

int bobe_gp_factor(bobe_gp_t* g) {
  API_BEGIN
  NEED(g, "gp is NULL");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  return g->factor_state();
  API_END
}

int bobe_gp_mll(bobe_gp_t* g, const double* ls, double kvar, double* mll, double* grad) {
  API_BEGIN
  NEED(g && ls && mll, "NULL argument");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  Hyper h = g->hyp;
  for (int j = 0; j < g->d; ++j) h.ls[j] = ls[j];
  h.kvar = kvar;
  g->mll_enqueue(h, grad != nullptr);
  return g->mll_collect(mll, grad);
  API_END
}

int bobe_gp_mll_batch(bobe_gp_t* g, int64_t B, const double* ls, const double* kvar, double* mll, double* grad,
                      int* status) {
  API_BEGIN
  NEED(g && ls && kvar && mll, "NULL argument");
  NEED(B >= 0, "B must be >= 0");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  return g->mll_batch(B, ls, kvar, mll, grad, status);
  API_END
}

int bobe_gp_mll_submit(bobe_gp_t* g, int slot, const double* ls, double kvar, int want_grad) {
  API_BEGIN
  NEED(g && ls, "NULL argument");
  NEED(slot >= 0 && slot < BOBE_MAX_MLL_SLOTS, "slot out of range");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->mll_submit(slot, ls, kvar, want_grad);
  return BOBE_OK;
  API_END
}

int bobe_gp_mll_wait(bobe_gp_t* g, int slot, double* mll, double* grad) {
  API_BEGIN
  NEED(g && mll, "NULL argument");
  return g->mll_wait(slot, mll, grad);
  API_END
}

int bobe_gp_predict(bobe_gp_t* g, const double* Xq, int64_t C, double* mean, double* var, int nan_policy) {
  API_BEGIN
  NEED(g && Xq, "NULL argument");
  g->use();
  g->sweep(Xq, C, nullptr, 0, 1.0, nullptr, nullptr, mean, var, nan_policy ? 1 : 0, nullptr, nullptr, nullptr, nullptr,
           nullptr, true);
  return BOBE_OK;
  API_END
}

int bobe_gp_wip_sweep(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                      double* wipv, double* wipstd, double* mean, double* var, int64_t* argmin_v, double* min_v,
                      int64_t* argmin_s, double* min_s) {
  API_BEGIN
  NEED(g && cand && Z, "NULL argument");
  g->use();
  g->sweep(cand, C, Z, M, y_std, wipv, wipstd, mean, var, 1, argmin_v, min_v, argmin_s, min_s, nullptr);
  return BOBE_OK;
  API_END
}

int bobe_gp_fantasy_var(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                        double* out) {
  API_BEGIN
  NEED(g && cand && Z && out, "NULL argument");
  g->use();
  g->sweep(cand, C, Z, M, y_std, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, out);
  return BOBE_OK;
  API_END
}

int bobe_gp_wip_grad(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                     double* wipv, double* wipstd, double* dwipv, double* dwipstd) {
  API_BEGIN
  NEED(g && cand && Z, "NULL argument");
  g->wip_grad(cand, C, Z, M, y_std, wipv, wipstd, dwipv, dwipstd);
  return BOBE_OK;
  API_END
}

int bobe_gp_acq_ei(bobe_gp_t* g, const double* Xq, int64_t C, double best_y, double zeta, int mode, double* out) {
  API_BEGIN
  NEED(g && Xq && out, "NULL argument");
  g->acq_ei(Xq, C, best_y, zeta, mode, out);
  return BOBE_OK;
  API_END
}

int bobe_gp_predict_grad(bobe_gp_t* g, const double* Xq, int64_t C, double* mean, double* var, double* dmean,
                         double* dvar) {
  API_BEGIN
  NEED(g && Xq && dmean, "NULL argument");
  NEED(dvar || !var, "var without dvar: use bobe_gp_predict");
  g->predict_grad(Xq, C, mean, var, dmean, dvar);
  return BOBE_OK;
  API_END
}

int bobe_gp_hmc_leapfrog(bobe_gp_t* g, int64_t P, double* U, double* Pm, const double* inv_mass, double eps, int L,
                         double y_std, double y_mean, double temp, double* logp, double* grad, double* mean, double* X) {
  API_BEGIN
  NEED(g && U && Pm && inv_mass && logp && grad && mean && X, "NULL argument");
  g->hmc_leapfrog(P, U, Pm, inv_mass, eps, L, y_std, y_mean, temp, logp, grad, mean, X);
  return BOBE_OK;
  API_END
}

int bobe_gp_hmc_run(bobe_gp_t* g, int64_t P, double* state, double* adapt, const double* inv_mass, uint64_t seed,
                    int64_t it0, int niter, int do_adapt, double y_std, double y_mean, double temp, int hist_from,
                    double* hist, int thin, double* keep, double* dbg) {
  API_BEGIN
  NEED(g && state && adapt && inv_mass, "NULL argument");
  NEED(!is_device_ptr(state) && !is_device_ptr(adapt) && !is_device_ptr(inv_mass) && !is_device_ptr(hist) &&
           !is_device_ptr(keep) && !is_device_ptr(dbg),
       "bobe_gp_hmc_run takes host pointers");
  g->hmc_run(P, state, adapt, inv_mass, seed, it0, niter, do_adapt, y_std, y_mean, temp, hist_from, hist, thin, keep, dbg);
  return BOBE_OK;
  API_END
}

int bobe_gp_rwalk(bobe_gp_t* g, int64_t P, double* X, double* logl, const double* step, double lstar, int walks,
                  uint64_t seed, double y_std, double y_mean, int* n_accepted, int* n_inside, double* dbg) {
  API_BEGIN
  NEED(g && X && logl && step && n_accepted && n_inside, "NULL argument");
  NEED(!is_device_ptr(X) && !is_device_ptr(logl) && !is_device_ptr(step) && !is_device_ptr(n_accepted) &&
           !is_device_ptr(n_inside) && !is_device_ptr(dbg),
       "bobe_gp_rwalk takes host pointers");
  g->rwalk(P, X, logl, step, lstar, walks, seed, y_std, y_mean, n_accepted, n_inside, dbg);
  return BOBE_OK;
  API_END
}

int bobe_gp_set_gate(bobe_gp_t* g, const double* support_vectors, int64_t n_sv, const double* dual_coef, double intercept,
                     double gamma, double probability_threshold, double minus_inf) {
  API_BEGIN
  NEED(g, "gp is NULL");
  NEED(n_sv >= 0, "n_sv must be >= 0");
  g->set_gate(support_vectors, n_sv, dual_coef, intercept, gamma, probability_threshold, minus_inf);
  return BOBE_OK;
  API_END
}

int bobe_gp_gate_eval(bobe_gp_t* g, const double* Xq, int64_t C, double* decision, double* feasible) {
  API_BEGIN
  NEED(g && Xq, "NULL argument");
  g->gate_eval(Xq, C, decision, feasible);
  return BOBE_OK;
  API_END
}

int bobe_gp_kernel(bobe_gp_t* g, const double* A, int64_t nA, const double* B, int64_t nB, const double* ls,
                   double kvar, double noise, int include_noise, double* out) {
  API_BEGIN
  NEED(g && A && B && out, "NULL argument");
  g->kernel_eval(A, nA, B, nB, ls, kvar, noise, include_noise, out);
  return BOBE_OK;
  API_END
}

int bobe_gp_dist_sq(bobe_gp_t* g, const double* A, int64_t nA, const double* B, int64_t nB, double* out) {
  API_BEGIN
  NEED(g && A && B && out, "NULL argument");
  g->kernel_eval(A, nA, B, nB, nullptr, 0.0, 0.0, 0, out, true);
  return BOBE_OK;
  API_END
}

int bobe_gp_mll_from_k(bobe_gp_t* g, const double* K, int64_t n, const double* y, double* mll) {
  API_BEGIN
  NEED(g && K && y && mll, "NULL argument");
  NEED(n >= 1, "n must be >= 1");
  return g->mll_from_k(K, n, y, mll);
  API_END
}

int bobe_gp_chol_row_update(bobe_gp_t* g, const double* L, int64_t n, const double* k, double k_self, double* v,
                            double* diag) {
  API_BEGIN
  NEED(g && L && k && v && diag, "NULL argument");
  NEED(n >= 1, "n must be >= 1");
  g->chol_row_update(L, n, k, k_self, v, diag);
  return BOBE_OK;
  API_END
}

int bobe_gp_set_pivot_floor_ulp(bobe_gp_t* g, double ulp) {
  API_BEGIN
  NEED(g, "gp is NULL");
  NEED(ulp >= 0.0, "ulp must be >= 0 (0: the sign test alone)");      // (NaN fails too)
  g->pivot_ulp = ulp;
  return BOBE_OK;
  API_END
}

double bobe_gp_get_pivot_floor_ulp(bobe_gp_t* g) { return g ? g->pivot_ulp : -1.0; }

int bobe_gp_set_refine_kappa(bobe_gp_t* g, double kappa) {
  API_BEGIN
  NEED(g, "gp is NULL");
  NEED(kappa == kappa, "kappa is NaN");
  g->refine_kappa = kappa;
  return BOBE_OK;
  API_END
}

int bobe_gp_set_solve_block(bobe_gp_t* g, int rows) {
  API_BEGIN
  NEED(g, "gp is NULL");
  NEED(rows > 0 && rows % bobe::TILE == 0, "rows must be a positive multiple of 128");
  g->solve_block = rows;
  return BOBE_OK;
  API_END
}

int bobe_gp_get_solve_block(bobe_gp_t* g) { return g ? g->solve_block : -1; }

int bobe_gp_get_refine(bobe_gp_t* g, double* kappa, int* active) {
  API_BEGIN
  NEED(g, "gp is NULL");
  if (kappa) *kappa = g->refine_kappa;
  if (active) *active = (g->factored && g->refine_v) ? 1 : 0;
  return BOBE_OK;
  API_END
}

int bobe_gp_get_chol(bobe_gp_t* g, double* L, double* alpha) {
  API_BEGIN
  NEED(g, "gp is NULL");
  g->get_chol(L, alpha);
  return BOBE_OK;
  API_END
}

int bobe_gp_set_chol(bobe_gp_t* g, const double* L, const double* alpha) {
  API_BEGIN
  NEED(g && L && alpha, "NULL argument");
  g->set_chol(L, alpha);
  return BOBE_OK;
  API_END
}

int bobe_gp_clone_state(bobe_gp_t* dst, bobe_gp_t* src) {
  API_BEGIN
  NEED(dst && src, "NULL argument");
  dst->clone_from(*src);
  return BOBE_OK;
  API_END
}

int bobe_gp_append(bobe_gp_t* g, const double* X_new, int64_t b, const double* y_all) {
  API_BEGIN
  NEED(g && X_new && y_all, "NULL argument");
  NEED(b >= 1 && b <= 64, "b must be in [1, 64] (larger batches: bobe_gp_set_data + bobe_gp_factor)");
  if (!g->factored || g->not_pd) throw Err(BOBE_ERR_STATE, "append needs a positive-definite factorised state");
  return g->append(X_new, b, y_all);
  API_END
}

int bobe_debug_gemm(int device, int la, int lb, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double* C, int64_t ldc) {
  API_BEGIN
  NEED(M % TILE == 0 && N % TILE == 0 && K % TK == 0 && M > 0 && N > 0 && K > 0, "bad GEMM shape");
  HIPCHK(hipSetDevice(device));
  configure_debug_kernels();
  const size_t na = (size_t)(la == 0 ? M * lda : K * lda), nbb = (size_t)(lb == 0 ? N * ldb : K * ldb), nc = (size_t)M * ldc;
  DBuf da, db, dc;
  const double *pa = A, *pb = B;
  double* pc = C;
  const bool ha = !is_device_ptr(A), hb = !is_device_ptr(B), hc = !is_device_ptr(C);
  if (ha) { da.ensure(na * 8); HIPCHK(hipMemcpy(da.p, A, na * 8, hipMemcpyHostToDevice)); pa = da.d(); }
  if (hb) { db.ensure(nbb * 8); HIPCHK(hipMemcpy(db.p, B, nbb * 8, hipMemcpyHostToDevice)); pb = db.d(); }
  if (hc) { dc.ensure(nc * 8); pc = dc.d(); }
  const dim3 grid((unsigned)(N / TILE), (unsigned)(M / TILE));
#define DG(a, b) hipLaunchKernelGGL((k_debug_gemm<a, b>), grid, dim3(256), GEMM_SMEM_BYTES, 0, pa, lda, pb, ldb, pc, ldc, K)
  if (la == 0 && lb == 0) DG(0, 0); else if (la == 0) DG(0, 1); else if (lb == 0) DG(1, 0); else DG(1, 1);
#undef DG
  LAUNCH_CHECK();
  HIPCHK(hipDeviceSynchronize());
  if (hc) HIPCHK(hipMemcpy(C, dc.p, nc * 8, hipMemcpyDeviceToHost));
  da.release(); db.release(); dc.release();
  return BOBE_OK;
  API_END
}

int bobe_debug_kinv(bobe_gp_t* g, double* Kinv) {
  API_BEGIN
  NEED(g && Kinv, "NULL argument");
  g->kinv_debug(Kinv);
  return BOBE_OK;
  API_END
}

int bobe_debug_linv(bobe_gp_t* g, double* Linv) {
  API_BEGIN
  NEED(g && Linv, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  g->use();
  g->copy_out_matrix(g->Linv.d(), Linv, 1);
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_debug_time_potrf(bobe_gp_t* g, int reps, double* ms) {
  API_BEGIN
  NEED(g && ms && reps >= 1, "bad argument");
  *ms = g->time_potrf(reps);
  return BOBE_OK;
  API_END
}

int bobe_debug_time_potrf_batch(bobe_gp_t* g, int B, int reps, double* ms) {
  API_BEGIN
  NEED(g && ms && reps >= 1 && B >= 1 && B <= BOBE_MAX_MLL_SLOTS, "bad argument");
  *ms = g->time_potrf_batch(B, reps);
  return BOBE_OK;
  API_END
}

int bobe_debug_time_potrf_lockstep(bobe_gp_t* g, int B, int reps, double* ms) {
  API_BEGIN
  NEED(g && ms && reps >= 1 && B >= 1 && B <= BOBE_MAX_MLL_SLOTS, "bad argument");
  *ms = g->time_potrf_lockstep(B, reps);
  return BOBE_OK;
  API_END
}

int bobe_gp_profile_select(bobe_gp_t* g, int tag) {
  API_BEGIN
  NEED(g, "gp is NULL");
  g->use();
  g->sync();
  g->prof_tag = tag;
  g->prof_used = 0;
  return BOBE_OK;
  API_END
}

int bobe_gp_profile_read(bobe_gp_t* g, double* total_ms, int64_t* launches) {
  API_BEGIN
  NEED(g && total_ms && launches, "NULL argument");
  g->use();
  g->sync();
  double tot = 0.0;
  for (size_t i = 0; i < g->prof_used; ++i) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, g->prof_events[i].first, g->prof_events[i].second));
    tot += t;
  }
  *total_ms = tot;
  *launches = (int64_t)g->prof_used;
  g->prof_used = 0;
  return BOBE_OK;
  API_END
}

}  // extern "C"

// -------------------------------------------------------------------------------------------------
// Multi-GPU exchange step, one process per GPU: a RCCL communicator owned by the library.
// -------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  DBuf send, recv;
  double* h_recv = nullptr;      // pinned
  size_t h_recv_doubles = 0;
  void load() {
    if (lib) return;
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) throw Err(BOBE_ERR_HIP, std::string("cannot open librccl.so: ") + dlerror());
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString)
      throw Err(BOBE_ERR_HIP, "librccl.so lacks an expected entry point");
  }
  void check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Err(BOBE_ERR_HIP, std::string(what) + ": " + GetErrorString(r));
  }
  // every rank contributes n doubles; returns world*n doubles (rank-major) in pinned host memory
  const double* all_gather(const double* mine, size_t n) {
    if (!comm) throw Err(BOBE_ERR_STATE, "call bobe_mgpu_init first");
    HIPCHK(hipSetDevice(device));
    send.ensure(n * sizeof(double));
    recv.ensure((size_t)world * n * sizeof(double));
    if (h_recv_doubles < (size_t)world * n) {
      if (h_recv) (void)hipHostFree(h_recv);
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_recv), (size_t)world * n * sizeof(double), hipHostMallocDefault));
      h_recv_doubles = (size_t)world * n;
    }
    HIPCHK(hipMemcpyAsync(send.p, mine, n * sizeof(double), hipMemcpyHostToDevice, stream));
    check(AllGather(send.p, recv.p, n, ncclDouble, comm, stream), "ncclAllGather");
    HIPCHK(hipMemcpyAsync(h_recv, recv.p, (size_t)world * n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return h_recv;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// merge rule of the exchange (jnp.argmin semantics, acquisition.py:397): smallest score, ties to the lowest global
// index, a NaN score counts as minimal
void merge_pairs(const double* all, int world, int stride, int off, double* best, int64_t* best_idx) {
  bool have = false;
  double bs = 0.0;
  int64_t bi = 0;
  for (int r = 0; r < world; ++r) {
    const double s = all[(size_t)r * stride + off];
    int64_t i;
    std::memcpy(&i, &all[(size_t)r * stride + off + 1], sizeof(i));      // the index travels as raw int64 bits
    if (i < 0) continue;                                                   // a rank without candidates
    const double key = std::isnan(s) ? -INFINITY : s, bkey = std::isnan(bs) ? -INFINITY : bs;
    if (!have || key < bkey || (key == bkey && i < bi)) {
      have = true;
      bs = s;
      bi = i;
    }
  }
  *best = have ? bs : std::nan("");
  *best_idx = have ? bi : -1;
}
}  // namespace
extern "C" {

int bobe_mgpu_unique_id(char* id128) {
  API_BEGIN
  if (!id128) throw Err(BOBE_ERR_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  g_rccl.load();
  ncclUniqueId id;
  g_rccl.check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id) == BOBE_MGPU_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return BOBE_OK;
  API_END
}

int bobe_mgpu_init(const char* id128, int world, int rank, int device) {
  API_BEGIN
  if (!id128 || world < 1 || rank < 0 || rank >= world) throw Err(BOBE_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.comm) throw Err(BOBE_ERR_STATE, "already initialised: call bobe_mgpu_finalize first");
  g_rccl.load();
  HIPCHK(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  g_rccl.check(g_rccl.CommInitRank(&g_rccl.comm, world, id, rank), "ncclCommInitRank");
  g_rccl.world = world;
  g_rccl.rank = rank;
  g_rccl.device = device;
  HIPCHK(hipStreamCreateWithFlags(&g_rccl.stream, hipStreamNonBlocking));
  return BOBE_OK;
  API_END
}

int bobe_mgpu_world(void) { return g_rccl.comm ? g_rccl.world : 0; }
int bobe_mgpu_rank(void) { return g_rccl.comm ? g_rccl.rank : -1; }

void bobe_mgpu_finalize(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.comm) return;
  (void)hipSetDevice(g_rccl.device);
  (void)hipStreamSynchronize(g_rccl.stream);
  (void)g_rccl.CommDestroy(g_rccl.comm);
  g_rccl.comm = nullptr;
  (void)hipStreamDestroy(g_rccl.stream);
  g_rccl.stream = nullptr;
  g_rccl.send.release();
  g_rccl.recv.release();
  if (g_rccl.h_recv) (void)hipHostFree(g_rccl.h_recv);
  g_rccl.h_recv = nullptr;
  g_rccl.h_recv_doubles = 0;
  g_rccl.world = 1;
  g_rccl.rank = 0;
}

int bobe_mgpu_wip_sweep(bobe_gp_t* g, const double* cand, int64_t C, int64_t global_offset, const double* Z, int64_t M,
                        double y_std, double* wipv, double* wipstd, double* mean, double* var, int64_t* argmin_v,
                        double* min_v, int64_t* argmin_s, double* min_s) {
  API_BEGIN
  // (argument errors are programming errors and the same on every rank; they are raised before the collective)
  if (!g || !Z) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (C < 0 || global_offset < 0) throw Err(BOBE_ERR_ARG, "bad shard");
  if (C > 0 && !cand) throw Err(BOBE_ERR_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.comm) throw Err(BOBE_ERR_STATE, "call bobe_mgpu_init first");
  if (g->device != g_rccl.device)
    throw Err(BOBE_ERR_ARG, "the handle lives on another device than the communicator of bobe_mgpu_init");
  // A failure of the LOCAL sweep (unfactored handle, out of memory, a failed launch) must not keep this rank out of the
  // collective - the others would wait in ncclAllGather for ever.  The rank joins with a status word in its payload
  // and every rank raises after the merge.
  int64_t lv = -1, ls = -1;
  double mv = std::nan(""), msd = std::nan("");
  int local_rc = BOBE_OK;
  std::string local_msg;
  if (C > 0) {
    try {
      g->use();
      g->sweep(cand, C, Z, M, y_std, wipv, wipstd, mean, var, 1, &lv, &mv, &ls, &msd, nullptr);
      lv += global_offset;
      ls += global_offset;
    } catch (const Err& e) {
      local_rc = e.code;
      local_msg = e.what();
    } catch (const std::exception& e) {
      local_rc = BOBE_ERR_HIP;
      local_msg = e.what();
    }
    if (local_rc != BOBE_OK) {
      lv = ls = -1;                                            // a rank without a result never wins the merge
      mv = msd = std::nan("");
    }
  }
  double mine[5];
  mine[0] = mv;
  std::memcpy(&mine[1], &lv, sizeof(lv));
  mine[2] = msd;
  std::memcpy(&mine[3], &ls, sizeof(ls));
  mine[4] = (double)local_rc;
  const double* all = g_rccl.all_gather(mine, 5);            // ONE collective per acquisition: 40 B per rank
  for (int r = 0; r < g_rccl.world; ++r) {
    const int rc = (int)all[(size_t)r * 5 + 4];
    if (rc != BOBE_OK)
      throw Err(rc, "bobe_mgpu_wip_sweep: the sweep of rank " + std::to_string(r) + " failed" +
                        (r == g_rccl.rank ? ": " + local_msg : std::string(" (see that rank's bobe_last_error)")));
  }
  double bv, bs;
  int64_t iv, is;
  merge_pairs(all, g_rccl.world, 5, 0, &bv, &iv);
  merge_pairs(all, g_rccl.world, 5, 2, &bs, &is);
  if (argmin_v) *argmin_v = iv;
  if (min_v) *min_v = bv;
  if (argmin_s) *argmin_s = is;
  if (min_s) *min_s = bs;
  return BOBE_OK;
  API_END
}

int bobe_mgpu_best_fit(double mll, const double* theta, int n, double* best_mll, double* best_theta) {
  API_BEGIN
  if (!theta || !best_mll || !best_theta || n < 1 || n > 126) throw Err(BOBE_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.comm) throw Err(BOBE_ERR_STATE, "call bobe_mgpu_init first");
  // (no local compute here that could fail on one rank only: the caller passes its restart's result, NaN included)
  std::vector<double> mine((size_t)n + 1);
  mine[0] = mll;
  std::memcpy(mine.data() + 1, theta, (size_t)n * sizeof(double));
  const double* all = g_rccl.all_gather(mine.data(), (size_t)n + 1);
  int br = -1;
  double bm = -INFINITY;
  for (int r = 0; r < g_rccl.world; ++r) {                    // max by mll, a non-finite mll never wins (pool.py:322-326)
    const double m = all[(size_t)r * (n + 1)];
    if (std::isfinite(m) && (br < 0 || m > bm)) {
      br = r;
      bm = m;
    }
  }
  if (br < 0) br = 0;
  *best_mll = all[(size_t)br * (n + 1)];
  std::memcpy(best_theta, all + (size_t)br * (n + 1) + 1, (size_t)n * sizeof(double));
  return BOBE_OK;
  API_END
}

// back-to-back v_mfma_f64_16x16x4_f64 on every CU: the ceiling the GEMM-shaped kernels are priced against
static __global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters) {
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int bobe_debug_mfma_peak(int device, int waves_per_simd, double* tflops) {
  API_BEGIN
  if (!tflops || waves_per_simd < 1 || waves_per_simd > 2) throw Err(BOBE_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(device));
  const int blocks = 256 * waves_per_simd, iters = 4000;
  DBuf o;
  o.ensure((size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, o.d(), 10);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, o.d(), iters);
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  o.release();
  return BOBE_OK;
  API_END
}

}  // extern "C"

namespace {
// the cross-lane sums of kernels_common.hpp on one wave (tests): in [64][D] row-major (lane, component), D = 8 / 16 / 32;
// out [2 D]: out[j] = component j as wave_sum_components leaves it (taken from the first lane of the component's group),
// out[D + j] = chain_wave_sum of component j read back from lane 63
template <int D>
__global__ __launch_bounds__(64) void k_debug_wave_sums(const double* __restrict__ in, double* __restrict__ out) {
  const int lane = threadIdx.x;
  double v[D];
#pragma unroll
  for (int j = 0; j < D; ++j) v[j] = in[lane * D + j];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const double s = chain_wave_sum(v[j]);
    if (lane == 63) out[D + j] = s;
  }
  const double c = wave_sum_components<D>(v, lane);
  if ((lane & (64 / D - 1)) == 0) out[lane / (64 / D)] = c;
}
}  // namespace

extern "C" {
int bobe_debug_wave_sums(int device, int D, const double* in, double* out) {
  API_BEGIN
  NEED(in && out && (D == 8 || D == 16 || D == 32), "bad argument");
  HIPCHK(hipSetDevice(device));
  DBuf di, dout;
  di.ensure((size_t)64 * D * 8);
  dout.ensure((size_t)2 * D * 8);
  HIPCHK(hipMemcpy(di.p, in, (size_t)64 * D * 8, hipMemcpyHostToDevice));
  if (D == 8) hipLaunchKernelGGL(k_debug_wave_sums<8>, dim3(1), dim3(64), 0, 0, (const double*)di.d(), dout.d());
  else if (D == 16) hipLaunchKernelGGL(k_debug_wave_sums<16>, dim3(1), dim3(64), 0, 0, (const double*)di.d(), dout.d());
  else hipLaunchKernelGGL(k_debug_wave_sums<32>, dim3(1), dim3(64), 0, 0, (const double*)di.d(), dout.d());
  LAUNCH_CHECK();
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, dout.p, (size_t)2 * D * 8, hipMemcpyDeviceToHost));
  di.release();
  dout.release();
  return BOBE_OK;
  API_END
}
}  // extern "C"
