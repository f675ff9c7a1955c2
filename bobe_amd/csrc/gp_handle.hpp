// The handle behind include/bobe_gp.h (struct bobe_gp) and the host helpers shared by the translation units of
// libbobe_gp.so:
//   gp_factor.hip     K(X,X) assembly, blocked Cholesky + launch plan, triangular inverse, alpha, MLL value / gradient
//                     (single evaluation, evaluation slots, lock-step batch, graph replay), state restore
//   gp_sweep.hip      prediction / acquisition sweep, score and posterior gradients, rank-b append
//   gp_consumers.hip  HMC on the surrogate, EI / LogEI, the classifier gate, GP.kernel, device clone
//   gp_abi.hip        the extern "C" layer, the RCCL exchange step, test / bench hooks
// Host side only: buffer management, launch sequencing, host/device pointer handling.  No CPU compute path exists:
// without a HIP device every entry point fails with BOBE_ERR_HIP.
#pragma once
#include "../../include/bobe_gp.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "gemm_f64.hpp"
#include "gp_types.hpp"

namespace bobe {

struct Err : std::runtime_error {
  int code;
  Err(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

extern thread_local std::string g_err;     // text behind bobe_last_error() (gp_abi.hip)

#define HIPCHK(expr)                                                                                           \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      throw ::bobe::Err(BOBE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                          std::to_string(__LINE__) + ")");                                     \
  } while (0)

#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

inline bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  void ensure(size_t b) {
    if (b <= bytes) return;
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    bytes = 0;
    HIPCHK(hipMalloc(&p, b));
    bytes = b;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  double* d() const { return static_cast<double*>(p); }
};

// The factorisation's rank test.  A pivot (L_jj^2) below 64 ulp of the kernel matrix's diagonal k(x,x) + noise is as large as
// the rounding accumulated in its column's update at N of a few thousand: it carries no information, and neither does the
// log-determinant built on it (a too small one - the optimiser is drawn to exactly these hyper-parameters).  Such a
// factorisation counts as NOT positive definite, like one with a non-positive pivot: NaN outputs, BOBE_NOT_PD.  (LAPACK's
// dpotrf, the reference's Cholesky, tests the sign only and fails or passes on the last bit in this regime)
// The factor is per handle: bobe_gp_set_pivot_floor_ulp, default from BOBE_PIVOT_FLOOR_ULP, else 0 = the rank test is OFF and
// LAPACK's rule alone holds (a pivot <= 0 or NaN fails, nothing else: the reference's behaviour).  The BO driver opts in
// with 64 (bobe_amd/bo.py).
inline double pivot_floor(const Hyper& h, double ulp) {
  const double f = ulp * 2.220446049250313e-16 * (h.kvar + h.noise);
  return f < 0.25 ? f : 0.25;                        // (the identity padding's pivots are 1)
}
// min_diag: the smallest L_jj of a factor (k_mll_terms, res[101]); NaN counts as failed
inline bool pivots_resolved(double min_diag, double floor) { return min_diag * min_diag >= floor; }
double default_pivot_floor_ulp();                    // BOBE_PIVOT_FLOOR_ULP, else 0 (the reference's sign test alone)
double default_refine_kappa();                       // BOBE_REFINE_KAPPA, else 1e6
int default_solve_block();                           // BOBE_SOLVE_BLOCK, else 128
int default_solve_panel();                           // BOBE_SOLVE_PANEL, else 512
int64_t default_solve_chunk();                       // BOBE_SOLVE_CHUNK, else 32768

template <typename K>
void allow_big_lds(K kernel, int bytes) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

// Tuning switches, read once per process from the environment (INTEGRATION.md lists them; none changes a result bit):
//   BOBE_SYRK32_BELOW   64x64-tile count below which a single-panel trailing update takes 32x32 tiles (512)
//   BOBE_TRTRI64        128-block count below which a level of the triangular inverse takes 64x64 tiles (600)
//   BOBE_PAIR_MIN       K = 256 update pairs while B * rem^2 exceeds this (300; 0: never)
//   BOBE_LOCKSTEP_MIN_N bobe_gp_mll_batch advances its evaluations in lock step from this many points (1: always)
//   BOBE_MLL_SLOTS      evaluations in flight below that size (8)
//   BOBE_GRAPH_MAX_N    a slot replays its pipeline as a hipGraph up to this many points (2048)
//   BOBE_XCD_SHARES     0: row-major tile order on every XCD (1)
//   BOBE_FILL           deferred trailing updates in the panel launches: 0 off, 1 where they pay (potrf), 2 everywhere
//   BOBE_TRACE          print launch plans and batch timings to stderr
struct Tuning {
  int syrk32_below, trtri64_below, pair_min, lockstep_min_n, mll_slots, graph_max_n, xcd_shares, fill;
  bool mll_slots_set, trace;
};
const Tuning& tuning();

constexpr int LAUUM64_BELOW = 1200;   // lower 128-tile count below which K^-1 runs on 64x64 tiles (fixes the order of the
                                      // gradient's partial sums: a function of N only)
constexpr int FILL_NEAR = 2;          // the last panels of a block column always come from the update launches
constexpr int FILL_CHUNK = 3;         // panels per filler visit of a tile (a filler must not outlast the panel, ~28 us)
constexpr int FILL_SLACK = 16;        // caught-up work the plan accepts per deferred unit (1 / 16)
constexpr int FILL_PHASE = 1024;      // fillers ride in panel launches with B * rem^2 <= this

// Grid of an equal-work tile launch whose workgroups take their tile from xcd_share() (gemm_f64.hpp): `per` logical
// tiles per XCD, grid = 8 * per.  Small launches keep the plain order (per = 0).
struct TileGrid { int grid, per; };
inline TileGrid tile_grid(int ntiles) {
  if (!tuning().xcd_shares || ntiles < 256) return {ntiles, 0};
  const int per = (ntiles + 7) / 8;
  return {8 * per, per};
}

// per translation unit: raise the dynamic-LDS limit of its kernels (once per device)
void configure_factor_kernels();
void configure_sweep_kernels();
void configure_consumer_kernels();

struct Depth { int first, count, nblocks; };

}  // namespace bobe

struct bobe_gp {
  typedef bobe::DBuf DBuf;
  typedef bobe::Hyper Hyper;
  int device = 0;
  int kern = 0;
  int d = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t N = 0, Np = 0;
  int nb = 0;
  Hyper hyp;
  double pivot_ulp = bobe::default_pivot_floor_ulp();     // the rank test's factor (0: sign test only, as dpotrf)
  double pivot_floor(const Hyper& h) const { return bobe::pivot_floor(h, pivot_ulp); }
  bool have_data = false, factored = false, not_pd = false;
  // Where the installed factor is ill conditioned - (kvar + noise) / smallest pivot above refine_kappa (bobe_gp_set_refine_kappa;
  // default BOBE_REFINE_KAPPA, else 1e6; 0: always, negative: never) - every V = L^-1 K(X, .) is SOLVED for by a blocked
  // forward substitution (sweep_kernels.hpp, k_blk_step) instead of multiplied out with the inverse factor; the vector form
  // of bobe_gp_wip_grad's few-candidate path takes one step of iterative refinement.  Decided when a factor is installed: the
  // same bits on every rank.
  double refine_kappa = bobe::default_refine_kappa();
  bool refine_v = false;
  // solve_block: rows of the substitution's diagonal blocks (a multiple of 128; sets the accuracy).  Speed only: solve_panel
  // = rows per long update launch, solve_chunk = candidates per launch sequence of that path (0: `chunk`)
  int solve_block = bobe::default_solve_block();
  int solve_panel = bobe::default_solve_panel();
  int64_t solve_chunk = bobe::default_solve_chunk();
  void decide_refinement(double min_diag);
  // V = L^-1 B for ncp (a multiple of 128) columns: the product with the inverse factor, or - refine_v - the blocked
  // substitution, which overwrites B [Np x ldb] and needs V (V may be NULL only for the plain product); qp: the column sums
  // of squares per row tile (k_trimul's epilogue)
  void solve_v(double* B, int64_t ldb, int64_t ncp, double* V, int64_t ldv, double* qp, int64_t ldq);
  // prepare_z() keeps its results (ZsT, W_Z, base_z) while the same host Z arrives again and nothing they depend on
  // changed: an L-BFGS refinement of one acquisition point calls bobe_gp_wip_grad dozens of times with one Z
  std::vector<double> z_seen;
  int64_t z_seen_m = -1;
  bool wz_ready = false;        // W_Z = K^-1 K(X,Z) (the score GRADIENTS need it; the sweep itself does not)
  void forget_z() { z_seen_m = -1; wz_ready = false; }
  int64_t chunk = 8192;
  bool chunk_set = false;       // bobe_gp_set_chunk was called: the caller's chunk holds for the substitution path too

  DBuf X, y, XsT, XsT2, A, Linv, A2, Linv2, Tmp, alpha, w, alpha2, w2, part, gpart, res, info, probs, flags, diag;
  int num_cus = 0;
  // sweep / predict workspace
  DBuf wg_ws;     // workspace of bobe_gp_wip_grad's few-candidates path
  DBuf in_stage, z_stage, CsT, ZsT, kXC, kXZ, VZ, WZ, basez, sc, qpart, pv, ps, o_mean, o_var, o_wipv, o_wipstd,
      o_misc, kin_a, kin_b, kout, vxc, vxc2;
  std::vector<bobe::Depth> depths;
  double* h_res = nullptr;  // pinned, 128 doubles
  double* h_in = nullptr;   // pinned, 16 x MAX_D doubles: host coordinates of bobe_gp_wip_grad's few-candidate path

  // ---- classifier gate (gp_consumers.hip): support vectors SoA + dual coefficients on the device
  DBuf gate_sv, gate_dual;
  bobe::Gate gate{nullptr, nullptr, 0, 0, 0.0, 0.0, 0.5, -1e5};
  void set_gate(const double* sv, int64_t n_sv, const double* dual, double intercept, double gamma, double threshold,
                double minus_inf);
  // decision / feasibility of C device-resident query points (row-major C x d); with mean / var / dmean / dvar (device,
  // any may be null) the gated entries are overwritten: mean = -inf, var = 1e-12, gradients 0
  void gate_apply(const double* xq_dev, int64_t C, double* decision, double* feasible, double* mean, double* var,
                  double* dmean, double* dvar);
  void gate_eval(const double* Xq, int64_t C, double* decision, double* feasible);

  // ---- launch plan of a factorisation (potrf): which panel launch carries which deferred update tiles, and from which
  // panel on every block column still has to be updated by each separate update launch.  Host logic only (a function of
  // the block count, the batch width and the CU count); the tables live on the device.
  struct CholOp {
    int kind;             // 0 panel, 1 narrow update (block column `first`), 2 trailing update (block columns >= first)
    int k;                // panel: block index; updates: one past the last panel to apply (k1)
    int first;
    int tab_off, tab_cnt; // panel: filler jobs [off, off + cnt) of `jobs`; updates: offset of the column table in `colk0`
    int k0_min, k0_max;   // updates: smallest / largest first pending panel over the columns the launch touches
    bool uniform;         // updates: every column from `first` on takes part with the same first panel (no table needed)
    int last_active;      // updates: last block column that takes part
    int k0_plain;         // updates: first pending panel of the columns that are not deferred (they all share it)
  };
  struct CholPlan {
    std::vector<CholOp> ops;
    std::vector<bobe::FillJob> jobs;
    std::vector<int> colk0;
    DBuf d_jobs, d_colk0;
    int far_start = 0;    // first deferred block column (nb: none)
    int64_t deferred_units = 0, catchup_units = 0, total_units = 0;
  };
  std::map<uint64_t, CholPlan> chol_plans;
  bool fill_pays(int B) const;
  const CholPlan& chol_plan(int B, bool fill);
  void build_plans();
  // strips per workgroup of the panel launch with `rr` blocks below the diagonal block (chol_kernels.hpp, panel_workgroups)
  int panel_strips(int B, int rr) const;

  // ---- evaluation slots of bobe_gp_mll_batch / _submit: a private stream + workspace per concurrently evaluated
  // hyper-parameter vector.  Slot 0 is the handle's own (stream, XsT2, A2, ...) set; a slot is made current by
  // swapping its members in, so every pipeline stage runs unchanged on it.
  // Replayable evaluation pipeline (launch-bound sizes): the kernels of one value(+gradient) evaluation captured
  // into a hipGraph per (workspace, data generation, with/without gradient).  The hyper-parameters reach the
  // kernels through a device-resident copy that the graph's first node refreshes from pinned host memory.
  struct EvalGraph {
    hipGraphExec_t exec[2] = {nullptr, nullptr};    // [want_grad]
    std::array<const void*, 16> sig[2] = {};        // every address / size the captured kernels were given
    Hyper* h_hyp = nullptr;                         // pinned
    DBuf hyp_dev;
  };
  struct Slot {
    hipStream_t stream = nullptr;
    DBuf XsT2, A2, Linv2, Tmp, alpha2, w2, part, gpart, res, info, flags, diag;
    EvalGraph eg;
    double* h_res = nullptr;
    hipEvent_t ev = nullptr;
    bool busy = false, want_grad = false;
  };
  EvalGraph eg;                    // of the handle's own workspace (swapped with a slot's like the buffers)
  std::array<const void*, 16> eval_signature() const {
    return {XsT2.p, A2.p, Linv2.p, Tmp.p, w2.p, alpha2.p, part.p, gpart.p, res.p, info.p, X.p, y.p, probs.p,
            static_cast<const void*>(h_res), reinterpret_cast<const void*>(static_cast<uintptr_t>(N)),
            static_cast<const void*>(stream)};
  }
  void mll_enqueue_body(const Hyper& h, bool want_grad, const Hyper* hdev);
  std::vector<Slot*> slots;
  std::vector<hipStream_t> slot_streams;     // one per evaluation slot, created on first use
  const std::vector<hipStream_t>& slot_stream_set();
  hipEvent_t ev_batch = nullptr;
  bool in_slot = false;
  void swap_slot(Slot& s) {
    std::swap(stream, s.stream);
    std::swap(XsT2, s.XsT2); std::swap(A2, s.A2); std::swap(Linv2, s.Linv2); std::swap(Tmp, s.Tmp);
    std::swap(alpha2, s.alpha2); std::swap(w2, s.w2); std::swap(part, s.part); std::swap(gpart, s.gpart);
    std::swap(res, s.res); std::swap(info, s.info); std::swap(flags, s.flags); std::swap(diag, s.diag);
    std::swap(h_res, s.h_res);
    std::swap(eg, s.eg);
    in_slot = !in_slot;
  }
  // Lock-step batch workspace (bobe_gp_mll_batch; from BOBE_LOCKSTEP_MIN_N points up when that is set): the B evaluations of a batch go
  // through ONE launch sequence on the handle's stream, every kernel taking the slot from its last grid dimension;
  // slot b lives at offset b * stride of each of these contiguous buffers.
  struct BatchWs {
    int cap = 0;
    int64_t Np = 0;
    DBuf A, Linv, Tmp, XsT, w, alpha, part, gpart, res, info, hyp, diag;
    Hyper* h_hyp = nullptr;      // pinned [BOBE_MAX_MLL_SLOTS]
    double* h_res = nullptr;     // pinned [BOBE_MAX_MLL_SLOTS][128]
  } bw;
  int64_t gpart_stride() const { return (int64_t)(2 * nb) * (2 * nb + 1) / 2 * (bobe::MAX_D + 1); }
  void ensure_batch(int B);
  void mll_lockstep_enqueue(int B, const Hyper* hs, bool want_grad);
  int mll_lockstep_collect(int B, double* mll, double* grad, int* status);
  std::mutex submit_mutex;          // serialises bobe_gp_mll_submit (the slot swap is not re-entrant)
  void ensure_slots(int n);
  void mll_enqueue(const Hyper& h, bool want_grad);
  int slot_collect(Slot& sl, double* mll, double* grad);
  int mll_collect(double* mll, double* grad);
  int mll_batch(int64_t B, const double* ls, const double* kvar, double* mll, double* grad, int* status);
  void mll_submit(int slot, const double* ls, double kvar, int want_grad);
  int mll_wait(int slot, double* mll, double* grad);

  // optional per-kernel-class timing with HIP events on the handle's stream (bobe_gp_profile_*)
  int prof_tag = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  void prof_begin(int tag) {
    if (tag != prof_tag) return;
    if (prof_used == prof_events.size()) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      HIPCHK(hipEventCreate(&b));
      prof_events.emplace_back(a, b);
    }
    HIPCHK(hipEventRecord(prof_events[prof_used].first, stream));
  }
  void prof_end(int tag) {
    if (tag != prof_tag) return;
    HIPCHK(hipEventRecord(prof_events[prof_used].second, stream));
    ++prof_used;
  }

  void use() {
    HIPCHK(hipSetDevice(device));
    bobe::configure_factor_kernels();
    bobe::configure_sweep_kernels();
    bobe::configure_consumer_kernels();
  }
  void sync() { HIPCHK(hipStreamSynchronize(stream)); }

  // host or device input -> device pointer (staged through `stage` when it is host memory)
  const double* fetch(const double* p, size_t n, DBuf& stage) {
    if (bobe::is_device_ptr(p)) return p;
    stage.ensure(n * sizeof(double));
    HIPCHK(hipMemcpyAsync(stage.p, p, n * sizeof(double), hipMemcpyHostToDevice, stream));
    return stage.d();
  }
  // output: device pointer to write to (user's when it is device memory, else `stage`)
  double* out_dev(double* user, size_t n, DBuf& stage) {
    if (!user) return nullptr;
    if (bobe::is_device_ptr(user)) return user;
    stage.ensure(n * sizeof(double));
    return stage.d();
  }
  void out_finish(double* user, size_t n, DBuf& stage) {
    if (!user || bobe::is_device_ptr(user)) return;
    HIPCHK(hipMemcpyAsync(user, stage.p, n * sizeof(double), hipMemcpyDeviceToHost, stream));
  }

  // ---- gp_factor.hip
  void build_probs();
  void alloc_for_n();
  void set_data(const double* X, const double* ys, int64_t N);
  // Batched forms (B > 1): slot b of a batch works on base + b * stride of every matrix / vector it is given and
  // reads its hyper-parameters from hdev[b]; B = 1 with zero strides is the plain call.
  void scale(const double* in, int64_t n, int64_t npad, const Hyper& h, double* out, int64_t ldo,
             const Hyper* hdev = nullptr, int B = 1, int64_t bsO = 0, int* info_reset = nullptr);
  // wv / prt: also prt[row tile][column] = the tile's share of out^T wv (k_gemv_t_part's partial sums, same bits)
  void kernel_matrix_cross(const double* AT, int64_t lda, int64_t na, int64_t napad, const double* BT, int64_t ldb,
                           int64_t nbv, int64_t nbpad, const Hyper& h, double* out, int64_t ldo,
                           const double* wv = nullptr, double* prt = nullptr, int64_t ldp = 0);
  void assemble_kxx(const Hyper& h, const double* xst, double* a, const Hyper* hdev = nullptr, int B = 1,
                    int64_t bsX = 0, int64_t bsA = 0);
  void syrk(double* a, int k0, int k1, int first, int colmode, int B = 1, int64_t bsA = 0, const int* colk0 = nullptr,
            int far_col = 0, int ncols = 0);
  // defer_diag: leave the L_kk scratch blocks where they are; the trtri() that follows puts them in place (one launch less)
  void potrf(double* a, double* linv, int* info_dev, int B = 1, int64_t bsA = 0, int64_t bsL = 0, double* dg = nullptr,
             bool defer_diag = false);
  int aside_first = 1 << 30;       // set by potrf(defer_diag = true), consumed by the next trtri()
  const double* aside_dg = nullptr;
  void trtri(double* a, double* linv, double* tmp, int B = 1, int64_t bsA = 0, int64_t bsL = 0, int64_t bsT = 0);
  int lauum(const Hyper& h, const double* linv, const double* al, const double* xst, double* kinv_out, int dcap,
            const Hyper* hdev = nullptr, double* gp_out = nullptr, int B = 1, int64_t bsL = 0, int64_t bsV = 0,
            int64_t bsX = 0, int64_t bsP = 0, double* scratch = nullptr, int64_t bsS = 0);
  // wv = Linv rhs, al = Linv^T wv (rhs: y unless given; bsY: its stride per batch member)
  void solve_alpha(const double* linv, double* wv, double* al, double* prt, int B = 1, int64_t bsL = 0, int64_t bsV = 0,
                   int64_t bsP = 0, const double* rhs = nullptr, int64_t bsY = 0);
  void factor_into(const Hyper& h, double* xst, double* a, double* linv, double* wv, double* al,
                   const Hyper* hdev = nullptr);
  std::string not_pd_text(int inf, double min_diag) const;
  // gp_mll(k, train_y, num_points) / fast_update_cholesky(L, k, k_self) on caller-supplied matrices (gp.py:170-197)
  void size_workspace(int64_t n);
  int mll_from_k(const double* K, int64_t n, const double* yv, double* mll);
  void chol_row_update(const double* L, int64_t n, const double* k, double k_self, double* v, double* diag_out);
  int factor_state();                                  // bobe_gp_factor
  void copy_out_matrix(const double* src, double* dst, int lower_only);
  void get_chol(double* L, double* alpha_out);
  void set_chol(const double* L, const double* alpha_in);
  double min_pivot_root();
  void kinv_debug(double* Kinv);
  double time_potrf(int reps);
  double time_potrf_batch(int B, int reps);
  double time_potrf_lockstep(int B, int reps);
  void fill(double* p, int64_t n, double v);

  // ---- gp_sweep.hip
  void prepare_z(const double* Z, int64_t M, int64_t Mp, bool need_w);
  // gated: apply the classifier gate (when one is set) to the mean / var outputs (the predict family, not the sweep)
  void sweep(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv, double* wipstd,
             double* mean, double* var, int policy, int64_t* argmin_v, double* min_v, int64_t* argmin_s, double* min_s,
             double* fantasy_out, bool gated = false);
  void wip_grad(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv, double* wipstd,
                double* dwipv, double* dwipstd);
  void predict_grad(const double* Xq, int64_t C, double* mean, double* var, double* dmean, double* dvar);
  int append(const double* X_new, int64_t b, const double* y_all);

  // ---- gp_consumers.hip
  void acq_ei(const double* Xq, int64_t C, double best_y, double zeta, int mode, double* out);
  void hmc_leapfrog(int64_t P, double* U, double* Pm, const double* inv_mass, double eps, int L, double y_std,
                    double y_mean, double temp, double* logp, double* grad, double* mean, double* X);
  void hmc_run(int64_t P, double* state, double* adapt, const double* inv_mass, uint64_t seed, int64_t it0, int niter,
               int do_adapt, double y_std, double y_mean, double temp, int hist_from, double* hist, int thin, double* keep,
               double* dbg);
  void rwalk(int64_t P, double* Xw, double* logl, const double* step, double lstar, int walks, uint64_t seed, double y_std,
             double y_mean, int* nacc, int* nin, double* dbg);
  // sqdist: the squared distances of the rows as they are (dist_sq, gp.py:80-96) instead of kernel values
  void kernel_eval(const double* A, int64_t nA, const double* B, int64_t nB, const double* ls, double kvar, double noise,
                   int include_noise, double* out, bool sqdist = false);
  void clone_from(bobe_gp& src);
  void release_all();
};
