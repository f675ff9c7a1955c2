// libbobe_gp.so — C ABI (include/bobe_gp.h) over the gfx950 kernels in kernels.hpp.
// Host side: buffer management, launch sequencing, host/device pointer handling.  No CPU
// compute path exists here: without a HIP device every entry point fails with BOBE_ERR_HIP.
#include "../../include/bobe_gp.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "kernels.hpp"

using namespace bobe;

namespace {

thread_local std::string g_err;

struct Err : std::runtime_error {
  int code;
  Err(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      throw Err(BOBE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                  std::to_string(__LINE__) + ")");                                     \
  } while (0)

#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

bool is_device_ptr(const void* p) {
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

template <typename K>
void allow_big_lds(K kernel, int bytes) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

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

constexpr int SYRK32_BK = 128, SYRK64_BK = 16;   // BK = 128 = the whole panel: one stage, one LDS buffer
constexpr int SYRK32_SMEM = gemm_smem_doubles_exact<KC, KC, 32, 32, SYRK32_BK>() * 8 / 2;  //  66,560 B (single buffer)
constexpr int SYRK64_SMEM = gemm_smem_doubles_exact<KC, KC, 64, 64, SYRK64_BK>() * 8;      //  36,864 B -> four workgroups per CU

void configure_kernels_once() {
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return;
  allow_big_lds(k_potf2<true, false>, POTF2_SMEM_BYTES);
  allow_big_lds(k_potf2<false, false>, POTF2_SMEM_BYTES);
  allow_big_lds(k_trti_diag, POTF2_SMEM_BYTES);
  allow_big_lds(k_trsm_panel<false>, TRSM_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, false, 3>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, false, 4>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, true, 3>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, true, 4>), POTF2_SMEM_BYTES);
  allow_big_lds(k_trtri_T<128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_trtri_R<128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_syrk_trail<64, SYRK64_BK>, SYRK64_SMEM);
  allow_big_lds(k_syrk_trail<32, SYRK32_BK>, SYRK32_SMEM);
  allow_big_lds(k_trtri_T<64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trtri_R<64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 8, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 16, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 32, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 8, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 16, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 32, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trimul, GEMM_SMEM_BYTES);
  allow_big_lds(k_trimul_t, GEMM_SMEM_BYTES);
  allow_big_lds(k_trimul_v64, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trimul_t64, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 8, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 16, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 32, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 8, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 16, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 32, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<0, 0>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<0, 1>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<1, 0>, GEMM_SMEM_BYTES);
  allow_big_lds(k_debug_gemm<1, 1>, GEMM_SMEM_BYTES);
  done[dev] = true;
}

struct Depth { int first, count, nblocks; };

// tile-size switches (128-tile counts below which the 64x64-tile variant of a kernel is launched);
// overridable through the environment for tuning runs
struct Tuning { int syrk32_below, trtri64_below, lauum64_below, chol_legacy, pair_min, mll_slots, own_queues, graph_max_n, lockstep_min_n, xcd_shares, filler_iters, filler_keep, fill, fill_near, fill_chunk, fill_slack, fill_phase, fill_inv, fill_inv_chunk, sweep_overlap, panel_strips; };
const Tuning& tuning() {
  static Tuning t = [] {
    Tuning v{512, 600, 1200, 0, 300, 4, 1, 2048, 1024, 1, 0, 0, 1, 2, 3, 16, 1024, 1, 4, 0, 0};
    if (const char* e = std::getenv("BOBE_SYRK32_BELOW")) v.syrk32_below = std::atoi(e);
    if (const char* e = std::getenv("BOBE_TRTRI64")) v.trtri64_below = std::atoi(e);
    if (const char* e = std::getenv("BOBE_LAUUM64")) v.lauum64_below = std::atoi(e);
    if (const char* e = std::getenv("BOBE_CHOL_LEGACY")) v.chol_legacy = std::atoi(e);   // always potf2 / trsm / syrk launches
    if (const char* e = std::getenv("BOBE_PAIR_MIN")) v.pair_min = std::atoi(e);   // K = 256 update pairs while B*rem^2 > this (0: never)
    if (const char* e = std::getenv("BOBE_LOCKSTEP_MIN_N")) v.lockstep_min_n = std::atoi(e);
    if (const char* e = std::getenv("BOBE_MLL_SLOTS")) v.mll_slots = std::atoi(e);
    if (const char* e = std::getenv("BOBE_OWN_QUEUES")) v.own_queues = std::atoi(e);
    if (const char* e = std::getenv("BOBE_GRAPH_MAX_N")) v.graph_max_n = std::atoi(e);
    if (const char* e = std::getenv("BOBE_XCD_SHARES")) v.xcd_shares = std::atoi(e);   // 0: row-major tile order on every XCD
    // timing experiment: stand-in MFMA workgroups on the CUs a panel launch leaves empty (results unaffected)
    if (const char* e = std::getenv("BOBE_FILLER_ITERS")) v.filler_iters = std::atoi(e);
    if (const char* e = std::getenv("BOBE_FILLER_KEEP")) v.filler_keep = std::atoi(e);      // CUs left empty anyway
    // deferred trailing updates in the shadow of the panel launches (potrf): 0 = off (every update in its own launch)
    if (const char* e = std::getenv("BOBE_FILL")) v.fill = std::atoi(e);
    if (const char* e = std::getenv("BOBE_FILL_NEAR")) v.fill_near = std::max(1, std::atoi(e));   // last panels of a column: never deferred
    if (const char* e = std::getenv("BOBE_FILL_CHUNK")) v.fill_chunk = std::max(1, std::atoi(e)); // panels per filler visit of a tile
    if (const char* e = std::getenv("BOBE_FILL_PHASE")) v.fill_phase = std::atoi(e);   // fillers ride in panel launches with B rem^2 <= this
    // tiles of the triangular inverse (its diagonal blocks and the T / R stages of its recursion) as fillers too: they have
    // no deadline, so they take whatever CUs the panel launches leave; 0 = off.  Chunk: 64-column units of K per visit
    if (const char* e = std::getenv("BOBE_FILL_INV")) v.fill_inv = std::atoi(e);
    if (const char* e = std::getenv("BOBE_FILL_INV_CHUNK")) v.fill_inv_chunk = std::max(1, std::atoi(e));
    // sweep of two chunks or more: 1 = the assembly of chunk i+1 on a second stream, under the GEMM launch of chunk i.
    // Off by default: with the posterior-mean products fused into the assembly there are 75 us per chunk left to hide, and
    // the GEMM launch it runs under loses 120 us (58.3 -> 58.8 ms per cycle at the headline size)
    if (const char* e = std::getenv("BOBE_SWEEP_OVERLAP")) v.sweep_overlap = std::atoi(e);
    // 16-row strips per panel workgroup: 0 = three wherever the launch fits the chip, else four; 3 / 4 force (A/B runs)
    if (const char* e = std::getenv("BOBE_PANEL_STRIPS")) v.panel_strips = std::atoi(e);
    if (const char* e = std::getenv("BOBE_FILL_SLACK")) v.fill_slack = std::max(1, std::atoi(e)); // deferred / caught-up work the plan accepts
    return v;
  }();
  return t;
}

// Grid of an equal-work tile launch whose workgroups take their tile from xcd_share() (gemm_f64.hpp): `per` logical
// tiles per XCD, grid = 8 * per.  Small launches keep the plain order (per = 0).
struct TileGrid { int grid, per; };
TileGrid tile_grid(int ntiles) {
  if (!tuning().xcd_shares || ntiles < 256) return {ntiles, 0};
  const int per = (ntiles + 7) / 8;
  return {8 * per, per};
}

}  // namespace

struct bobe_gp {
  int device = 0;
  int kern = 0;
  int d = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t N = 0, Np = 0;
  int nb = 0;
  Hyper hyp;
  bool have_data = false, factored = false, not_pd = false;
  // prepare_z() keeps its results (ZsT, W_Z, base_z) while the same host Z arrives again and nothing they depend on
  // changed: an L-BFGS refinement of one acquisition point calls bobe_gp_wip_grad dozens of times with one Z
  std::vector<double> z_seen;
  int64_t z_seen_m = -1;
  void forget_z() { z_seen_m = -1; }
  int64_t chunk = 8192;

  DBuf X, y, XsT, XsT2, A, Linv, A2, Linv2, Tmp, alpha, w, alpha2, w2, part, gpart, res, info, probs, flags, diag;
  int num_cus = 0;
  // sweep / predict workspace
  DBuf wg_ws;     // workspace of bobe_gp_wip_grad's few-candidates path
  DBuf filler_ws; // BOBE_FILLER_ITERS experiment
  // Launch plan of a factorisation (potrf): which panel launch carries which deferred update tiles, and from which
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
    std::vector<FillJob> jobs;
    std::vector<int> colk0;
    DBuf d_jobs, d_colk0;
    int far_start = 0;    // first deferred block column (nb: none)
    int64_t deferred_units = 0, catchup_units = 0;
    // what is left of the inverse after the factorisation when part of it rode in the panel launches: per recursion depth
    // the problems whose T / R stage still has to run (TriProb lists with their own tile offsets), on the device
    bool inverse_started = false;
    struct Rest { int first_t, count_t, nblocks_t, first_r, count_r, nblocks_r, skip_t, skip_r; };
    std::vector<Rest> rest;               // [depth]
    std::vector<TriProb> rest_probs;
    std::vector<unsigned char> rest_skip;  // per list: 1 = the tile ran inside the factorisation
    DBuf d_rest, d_skip;
    int64_t inv_units = 0, inv_units_total = 0;
  };
  std::map<uint64_t, CholPlan> chol_plans;
  const CholPlan& chol_plan(int B, bool fill, bool inv = false);
  // strips per workgroup of the panel launch with `rr` blocks below the diagonal block (chol_kernels.hpp, panel_workgroups)
  int panel_strips(int B, int rr) const {
    const int ps = tuning().panel_strips;
    if (ps == 3 || ps == 4) return ps;
    return B * panel_workgroups(rr, 3) <= std::max(num_cus, 1) ? 3 : 4;
  }
  // second K(X, chunk) buffer + mean partials + events of the sweep's assembly stream (sweep())
  DBuf kXC2, part_aux;
  hipEvent_t ev_sw[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  DBuf in_stage, z_stage, CsT, ZsT, kXC, kXZ, VZ, WZ, basez, sc, qpart, pv, ps, o_mean, o_var, o_wipv, o_wipstd,
      o_misc, kin_a, kin_b, kout;
  std::vector<Depth> depths;
  double* h_res = nullptr;  // pinned, 128 doubles

  // Extra evaluation slots of bobe_gp_mll_batch: a private stream + workspace per concurrently evaluated
  // hyper-parameter vector.  Slot 0 is the handle's own (stream, XsT2, A2, ...) set; a slot is made current by
  // swapping its members in, so every pipeline stage below runs unchanged on it.
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
  // Lock-step batch workspace (bobe_gp_mll_batch from lockstep_min_n points up): the B evaluations of a batch go
  // through ONE launch sequence on the handle's stream, every kernel taking the slot from its last grid dimension;
  // slot b lives at offset b * stride of each of these contiguous buffers.
  struct BatchWs {
    int cap = 0;
    int64_t Np = 0;
    DBuf A, Linv, Tmp, XsT, w, alpha, part, gpart, res, info, hyp, diag;
    Hyper* h_hyp = nullptr;      // pinned [BOBE_MAX_MLL_SLOTS]
    double* h_res = nullptr;     // pinned [BOBE_MAX_MLL_SLOTS][128]
  } bw;
  int64_t gpart_stride() const { return (int64_t)(2 * nb) * (2 * nb + 1) / 2 * (MAX_D + 1); }
  void ensure_batch(int B);
  void mll_lockstep_enqueue(int B, const Hyper* hs, bool want_grad);
  int mll_lockstep_collect(int B, double* mll, double* grad, int* status);
  std::mutex submit_mutex;          // serialises bobe_gp_mll_submit (the slot swap is not re-entrant)
  void ensure_slots(int n);
  void mll_enqueue(const Hyper& h, bool want_grad);
  int slot_collect(Slot& sl, double* mll, double* grad);
  int mll_collect(double* mll, double* grad);

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
    configure_kernels_once();
  }
  void sync() { HIPCHK(hipStreamSynchronize(stream)); }

  // host or device input -> device pointer (staged through `stage` when it is host memory)
  const double* fetch(const double* p, size_t n, DBuf& stage) {
    if (is_device_ptr(p)) return p;
    stage.ensure(n * sizeof(double));
    HIPCHK(hipMemcpyAsync(stage.p, p, n * sizeof(double), hipMemcpyHostToDevice, stream));
    return stage.d();
  }
  // output: device pointer to write to (user's when it is device memory, else `stage`)
  double* out_dev(double* user, size_t n, DBuf& stage) {
    if (!user) return nullptr;
    if (is_device_ptr(user)) return user;
    stage.ensure(n * sizeof(double));
    return stage.d();
  }
  void out_finish(double* user, size_t n, DBuf& stage) {
    if (!user || is_device_ptr(user)) return;
    HIPCHK(hipMemcpyAsync(user, stage.p, n * sizeof(double), hipMemcpyDeviceToHost, stream));
  }

  void build_probs();
  void alloc_for_n();
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
  // tmp: the inverse's scratch matrix - part of the inverse may then ride in the panel launches; the plan that says what is
  // left of it is returned (nullptr: all of it) and goes to the trtri() that follows
  const CholPlan* potrf(double* a, double* linv, int* info_dev, int B = 1, int64_t bsA = 0, int64_t bsL = 0,
                        double* dg = nullptr, bool defer_diag = false, double* tmp = nullptr, int64_t bsT = 0);
  int aside_first = 1 << 30;       // set by potrf(defer_diag = true), consumed by the next trtri()
  const double* aside_dg = nullptr;
  void trtri(double* a, double* linv, double* tmp, int B = 1, int64_t bsA = 0, int64_t bsL = 0, int64_t bsT = 0,
             const CholPlan* rest = nullptr);
  int lauum(const Hyper& h, const double* linv, const double* al, const double* xst, double* kinv_out, int dcap,
            const Hyper* hdev = nullptr, double* gp_out = nullptr, int B = 1, int64_t bsL = 0, int64_t bsV = 0,
            int64_t bsX = 0, int64_t bsP = 0);
  void solve_alpha(const double* linv, double* wv, double* al, double* prt, int B = 1, int64_t bsL = 0, int64_t bsV = 0,
                   int64_t bsP = 0);
  void factor_into(const Hyper& h, double* xst, double* a, double* linv, double* wv, double* al,
                   const Hyper* hdev = nullptr);
  int read_info();
  void prepare_z(const double* Z, int64_t M, int64_t Mp);
  void sweep(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv, double* wipstd,
             double* mean, double* var, int policy, int64_t* argmin_v, double* min_v, int64_t* argmin_s, double* min_s,
             double* fantasy_out);
};

void bobe_gp::build_probs() {
  struct Item { int depth, lo, mid, hi; };
  std::vector<Item> items;
  struct Rec {
    static void go(std::vector<Item>& it, int depth, int lo, int hi) {
      if (hi - lo <= 1) return;
      const int mid = lo + (hi - lo) / 2;
      it.push_back({depth, lo, mid, hi});
      go(it, depth + 1, lo, mid);
      go(it, depth + 1, mid, hi);
    }
  };
  Rec::go(items, 0, 0, nb);
  int maxd = -1;
  for (auto& i : items) maxd = i.depth > maxd ? i.depth : maxd;
  std::vector<TriProb> flat;
  depths.clear();
  for (int dd = 0; dd <= maxd; ++dd) {
    Depth D{(int)flat.size(), 0, 0};
    for (auto& i : items)
      if (i.depth == dd) {
        flat.push_back({i.lo, i.mid, i.hi, D.nblocks});
        D.nblocks += (i.hi - i.mid) * (i.mid - i.lo);
        D.count++;
      }
    depths.push_back(D);
  }
  if (!flat.empty()) {
    probs.ensure(flat.size() * sizeof(TriProb));
    HIPCHK(hipMemcpy(probs.p, flat.data(), flat.size() * sizeof(TriProb), hipMemcpyHostToDevice));
  }
}

void bobe_gp::alloc_for_n() {
  const size_t mat = (size_t)Np * Np * sizeof(double);
  const size_t vec = (size_t)Np * sizeof(double);
  A.ensure(mat);
  Linv.ensure(mat);
  A2.ensure(mat);
  Linv2.ensure(mat);
  Tmp.ensure(mat);
  y.ensure(vec);
  alpha.ensure(vec);
  w.ensure(vec);
  alpha2.ensure(vec);
  w2.ensure(vec);
  XsT.ensure((size_t)d * vec);
  XsT2.ensure((size_t)d * vec);
  const int64_t pw = Np > chunk ? Np : chunk;
  part.ensure((size_t)nb * pw * sizeof(double));
  gpart.ensure((size_t)(2 * nb) * (2 * nb + 1) / 2 * (MAX_D + 1) * sizeof(double));
  res.ensure(128 * sizeof(double));
  info.ensure(sizeof(int));
  flags.ensure((size_t)nb * sizeof(int));
  diag.ensure((size_t)nb * TILE * TILE * sizeof(double));
  build_probs();
}

void bobe_gp::scale(const double* in, int64_t n, int64_t npad, const Hyper& h, double* out, int64_t ldo,
                    const Hyper* hdev, int B, int64_t bsO, int* info_reset) {
  hipLaunchKernelGGL(k_scale_coords, dim3((unsigned)((npad + 255) / 256), (unsigned)B), dim3(256), 0, stream, in, n, npad,
                     h, out, ldo, hdev, bsO, info_reset);
  LAUNCH_CHECK();
}

#define KM_LAUNCH(KE, SQ, DC, grid, ...)                                                                 \
  do {                                                                                                   \
    if (h.d == DC)                                                                                       \
      hipLaunchKernelGGL((k_kernel_matrix<KE, SQ, DC, true>), grid, dim3(256), 0, stream, __VA_ARGS__);  \
    else                                                                                                 \
      hipLaunchKernelGGL((k_kernel_matrix<KE, SQ, DC, false>), grid, dim3(256), 0, stream, __VA_ARGS__); \
  } while (0)
#define KM_DISPATCH(SQ, grid, ...)                                                                    \
  do {                                                                                                \
    const int dc_ = h.d <= 8 ? 8 : (h.d <= 16 ? 16 : 32);                                             \
    if (h.kern == 0) {                                                                                \
      if (dc_ == 8) KM_LAUNCH(0, SQ, 8, grid, __VA_ARGS__);                                           \
      else if (dc_ == 16) KM_LAUNCH(0, SQ, 16, grid, __VA_ARGS__);                                    \
      else KM_LAUNCH(0, SQ, 32, grid, __VA_ARGS__);                                                   \
    } else {                                                                                          \
      if (dc_ == 8) KM_LAUNCH(1, SQ, 8, grid, __VA_ARGS__);                                           \
      else if (dc_ == 16) KM_LAUNCH(1, SQ, 16, grid, __VA_ARGS__);                                    \
      else KM_LAUNCH(1, SQ, 32, grid, __VA_ARGS__);                                                   \
    }                                                                                                 \
  } while (0)

void bobe_gp::kernel_matrix_cross(const double* AT, int64_t lda, int64_t na, int64_t napad, const double* BT,
                                  int64_t ldb, int64_t nbv, int64_t nbpad, const Hyper& h, double* out, int64_t ldo,
                                  const double* wv, double* prt, int64_t ldp) {
  const dim3 grid((unsigned)(nbpad / TILE), (unsigned)(napad / TILE));
  KM_DISPATCH(false, grid, AT, lda, na, BT, ldb, nbv, h, out, ldo, (const Hyper*)nullptr, (int64_t)0, (int64_t)0, wv, prt,
              ldp);
  LAUNCH_CHECK();
}

void bobe_gp::assemble_kxx(const Hyper& h, const double* xst, double* a, const Hyper* hdev, int B, int64_t bsX,
                           int64_t bsA) {
  const dim3 grid((unsigned)(2 * nb * (nb + 1)), (unsigned)B);   // four workgroups per lower 128x128 tile
  prof_begin(BOBE_PROF_KXX);
  KM_DISPATCH(true, grid, xst, Np, N, xst, Np, N, h, a, Np, hdev, bsX, bsA, (const double*)nullptr, (double*)nullptr,
              (int64_t)0);
  prof_end(BOBE_PROF_KXX);
  LAUNCH_CHECK();
}
#undef KM_DISPATCH
#undef KM_LAUNCH

// Trailing update with the panels of 128-blocks [k0, k1): colmode 0 = every lower tile from 128-block `first`
// on, colmode 1 = only 128-block column `first` (rows from `first` down).  A tile's time is set by its MFMAs
// per wave (512 / 128 / 32 per 128 of K): small trailing matrices take the smallest tile that still fills the
// chip, large ones the cheapest by a rounds x tile-time estimate.  (Tile shape does not change the bits: every
// element accumulates its K range in the same order, four k per MFMA.)
// colk0 (device table, one entry per block column): the columns of the launch start at different panels (deferred
// columns, see potrf) - 64 x 64 tiles only.
void bobe_gp::syrk(double* a, int k0, int k1, int first, int colmode, int B, int64_t bsA, const int* colk0, int far_col,
                   int ncols) {
  const Tuning& tu = tuning();
  const int rem = nb - first;                 // 128-blocks in the trailing matrix
  if (rem <= 0 || (!colk0 && k1 <= k0)) return;
  const int kb = k1 - k0;
  const int n64 = 2 * rem, n32 = 4 * rem;
  // colmode 2: only the first `ncols` block columns of the trailing matrix take part (the rest is deferred)
  const int nc64 = 2 * ncols;
  const int t64 = colmode == 2 ? nc64 * n64 - nc64 * (nc64 - 1) / 2 : (colmode ? 2 * n64 - 1 : n64 * (n64 + 1) / 2);
  const int t32 = colmode ? 4 * n32 - 6 : n32 * (n32 + 1) / 2;
  // 64x64 tiles with BK = 16 (36 KB of LDS, four workgroups per CU) have the best saturated throughput of all
  // variants at every K (tools/ubench_syrk.hip); when they would leave most of the chip idle, a single-panel
  // update takes 32x32 tiles, which stage the whole K = 128 panel in one LDS buffer
  if (kb == 1 && !colk0 && colmode != 2 && B * t64 < tu.syrk32_below) {
    hipLaunchKernelGGL((k_syrk_trail<32, SYRK32_BK>), dim3(t32, B), dim3(256), SYRK32_SMEM, stream, a, Np, k0, k1, first,
                       colmode, n32, bsA, 0, (const int*)nullptr, 0, 0);
    return;
  }
  const TileGrid tg = colmode ? TileGrid{t64, 0} : tile_grid(t64);
  hipLaunchKernelGGL((k_syrk_trail<64, SYRK64_BK>), dim3(tg.grid, B), dim3(256), SYRK64_SMEM, stream, a, Np, k0, k1, first,
                     colmode, n64, bsA, tg.per, colk0, far_col, nc64);
}

// The launch plan of a factorisation of B matrices in lock step (see potrf).  Block columns >= far_start are DEFERRED:
// the update launches leave them alone until they are about to be factored (the last `fill_near` panels of a column
// always come from the update launches), and the panel launches carry their pending updates as filler workgroups on the
// CUs the panel workgroups do not occupy - whole block columns at a time, `fill_chunk` panels per visit (a filler must
// not outlast the panel, ~28 us), earliest deadline first.  far_start is the smallest column from which the fillers keep
// up (what they leave behind is caught up by the update launch that makes the column current, with a longer K range).
const bobe_gp::CholPlan& bobe_gp::chol_plan(int B, bool fill, bool inv) {
  const Tuning& tu = tuning();
  const uint64_t key = ((uint64_t)nb << 32) | ((uint64_t)B << 8) | (fill ? 1u : 0u) | (inv ? 2u : 0u);
  auto it = chol_plans.find(key);
  if (it != chol_plans.end()) return it->second;
  const int ncu = std::max(num_cus, 1);
  const int D = tu.fill_near, CH = tu.fill_chunk, ICH = tu.fill_inv_chunk;
  auto npanel = [&](int k) { return panel_workgroups(nb - 1 - k, panel_strips(B, nb - 1 - k)); };
  auto one_launch = [&](int k) { return !tu.chol_legacy && B * npanel(k) <= ncu; };
  auto tiles_of = [&](int c) { return 4 * (nb - c) - 1; };   // 64 x 64 tiles of block column c from its diagonal block down

  // ---- the inverse's recursion (build_probs) as a task graph: DIAG(b) per diagonal block, T(P) and R(P) per problem
  struct Tile { int ti, tj, kbeg, kend, kcur; };
  struct Stage {
    int depth, prob, kind;        // kind 1 = T, 2 = R
    int lo, mid, hi;
    int dep_a, dep_b;             // stages (indices) that must be complete; -1 - b: DIAG of block b
    int min_launch;               // panels that must have run: a filler of launch k reads what launches < k wrote
    std::vector<Tile> tiles;
    int left, done_launch;        // tiles not finished yet; launch that finished the stage (INT_MAX: not finished)
  };
  std::vector<Stage> stages;
  std::vector<int> node_done_stage;   // per problem (flat index over depths): index of its R stage
  std::vector<TriProb> flat;          // the problems in build_probs' order (depth-major)
  std::vector<int> flat_depth;
  if (inv) {
    struct Item { int depth, lo, mid, hi; };
    std::vector<Item> items;
    struct Rec {
      static void go(std::vector<Item>& v, int depth, int lo, int hi) {
        if (hi - lo <= 1) return;
        const int mid = lo + (hi - lo) / 2;
        v.push_back({depth, lo, mid, hi});
        go(v, depth + 1, lo, mid);
        go(v, depth + 1, mid, hi);
      }
    };
    Rec::go(items, 0, 0, nb);
    int maxd = -1;
    for (auto& i : items) maxd = std::max(maxd, i.depth);
    for (int dd = 0; dd <= maxd; ++dd)
      for (auto& i : items)
        if (i.depth == dd) {
          flat.push_back({i.lo, i.mid, i.hi, 0});
          flat_depth.push_back(dd);
        }
    // the stage that completes the inverse of block range [lo, hi): R of that problem, or DIAG(lo) for a single block
    auto range_done = [&](int lo, int hi) -> int {
      if (hi - lo == 1) return -1 - lo;
      for (size_t q = 0; q < flat.size(); ++q)
        if (flat[q].lo == lo && flat[q].hi == hi) return 2 * (int)q + 1;
      return INT_MIN;
    };
    for (size_t q = 0; q < flat.size(); ++q) {
      const TriProb& P = flat[q];
      const int rows = (P.hi - P.mid) * 2, w = (P.mid - P.lo) * 2;           // in 64 x 64 tiles
      Stage T{flat_depth[q], (int)q, 1, P.lo, P.mid, P.hi, range_done(P.lo, P.mid), INT_MIN, P.mid, {}, 0, INT_MAX};
      Stage R{flat_depth[q], (int)q, 2, P.lo, P.mid, P.hi, 2 * (int)q, range_done(P.mid, P.hi), P.hi, {}, 0, INT_MAX};
      for (int tj = 0; tj < w; ++tj)
        for (int ti = 0; ti < rows; ++ti) {
          // T: Tmp[m0][n0] = sum_{k in [n0, mid)} L[m0][k] Linv[k][n0];  R: Linv[m0][n0] = -sum_{k in [mid, m0 + 64)} Linv[m0][k] Tmp[k][n0]
          T.tiles.push_back({2 * P.mid + ti, 2 * P.lo + tj, 2 * P.lo + tj, 2 * P.mid, 2 * P.lo + tj});
          R.tiles.push_back({2 * P.mid + ti, 2 * P.lo + tj, 2 * P.mid, 2 * P.mid + ti + 1, 2 * P.mid});
        }
      T.left = R.left = (int)T.tiles.size();
      stages.push_back(T);      // index 2q
      stages.push_back(R);      // index 2q + 1
    }
  }

  typedef std::vector<std::vector<char>> TileFlags;       // [stage][tile]
  auto build = [&](int far, const TileFlags* keep, CholPlan* out, TileFlags* finished, bool do_inv) {
    std::vector<int> applied(nb, 0);
    int64_t deferred = 0, catchup = 0, inv_units = 0;
    std::vector<Stage> st = stages;                     // (fresh progress per simulation)
    std::vector<int> diag_done(nb, INT_MAX);            // launch that inverted diagonal block b
    auto dep_done = [&](int dep) -> int {               // launch after which a dependency is complete
      if (dep == INT_MIN) return -1;
      if (dep < 0) return diag_done[-1 - dep];
      return st[dep].done_launch;
    };
    auto panel = [&](int k) {
      CholOp op{0, k, k, out ? (int)out->jobs.size() : 0, 0, 0, 0, true, k, k};
      const int64_t remk = nb - 1 - k;
      int cap = one_launch(k) ? (ncu - tu.filler_keep - B * npanel(k)) / B : 0;   // filler workgroups per slot, two jobs each
      if (fill && far < nb && cap > 0 && (int64_t)B * remk * remk <= tu.fill_phase) {
        for (int c = std::max(far, k + 2); c < nb && cap > 0; ++c) {     // earliest deadline first
          const int pend = std::min(k, c - D);                           // panels < k are final; the last D are never deferred
          if (applied[c] >= pend) continue;
          const int k1 = std::min(applied[c] + CH, pend);
          const int need = (tiles_of(c) + 1) / 2;
          if (need > cap) continue;
          cap -= need;
          deferred += (int64_t)tiles_of(c) * (k1 - applied[c]);
          if (out) {
            for (int tj = 2 * c; tj <= 2 * c + 1; ++tj)
              for (int ti = tj; ti < 2 * nb; ++ti) out->jobs.push_back({0, ti, tj, 2 * applied[c], 2 * k1, 0, 0, 0});
            FillJob twin = out->jobs.back();                             // an odd count: a twin that is computed, not stored,
            twin.flags |= FILL_TWIN;                                     // keeps the two groups of a workgroup in step
            out->jobs.push_back(twin);
            op.tab_cnt += tiles_of(c) + 1;
          }
          applied[c] = k1;
        }
      }
      if (do_inv && cap > 0) {
        // diagonal blocks first (everything else waits for them), then the ready stages, deepest level first
        for (int b = 0; b < k && b < nb && cap > 0; ++b) {
          if (diag_done[b] != INT_MAX) continue;
          diag_done[b] = k;
          --cap;
          inv_units += 8;
          if (out) {
            out->jobs.push_back({3, b, b, 0, 0, one_launch(b) ? FILL_ASIDE : 0, 0, 0});
            out->jobs.push_back({4, 0, 0, 0, 0, 0, 0, 0});
            op.tab_cnt += 2;
          }
        }
        std::vector<int> order;
        for (size_t i = 0; i < st.size(); ++i) {
          Stage& S = st[i];
          if (S.left == 0 || S.min_launch > k) continue;
          if (dep_done(S.dep_a) >= k || dep_done(S.dep_b) >= k) continue;     // (INT_MAX: not done at all)
          order.push_back((int)i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return st[a].depth > st[b].depth; });
        const int launches_left = nb - k;               // panel launches from this one on (an upper bound on a tile's visits)
        for (int i : order) {
          if (cap <= 0) break;
          Stage& S = st[i];
          // This launch's visit of the unfinished tiles: at most ICH 64-column units each.  A tile is worth starting only if
          // it can still finish inside the factorisation (one visit per launch): least slack first; tiles of equal visit
          // length share a workgroup.
          struct Cand { int t, len, slack; };
          std::vector<Cand> pend;
          for (size_t t = 0; t < S.tiles.size(); ++t) {
            const Tile& tl = S.tiles[t];
            if (tl.kcur >= tl.kend || (keep && !(*keep)[i][t])) continue;
            const int need = (tl.kend - tl.kcur + ICH - 1) / ICH;
            if (need > launches_left) continue;
            pend.push_back({(int)t, std::min(ICH, tl.kend - tl.kcur), launches_left - need});
          }
          std::stable_sort(pend.begin(), pend.end(), [](const Cand& a, const Cand& b) {
            return a.slack != b.slack ? a.slack < b.slack : a.len > b.len;
          });
          std::vector<char> used(pend.size(), 0);
          for (size_t p = 0; p < pend.size() && cap > 0; ++p) {
            if (used[p]) continue;
            used[p] = 1;
            size_t q = p + 1;                             // a partner of the same visit length (the nearest in slack order)
            while (q < pend.size() && (used[q] || pend[q].len != pend[p].len)) ++q;
            const int n = q < pend.size() ? 2 : 1;
            if (n == 2) used[q] = 1;
            --cap;
            FillJob last{};
            for (int u = 0; u < n; ++u) {
              Tile& tl = S.tiles[pend[u == 0 ? p : q].t];
              const int len = pend[p].len;
              const bool first = tl.kcur == tl.kbeg, fin = tl.kcur + len == tl.kend;
              last = {S.kind, tl.ti, tl.tj, tl.kcur, tl.kcur + len,
                      (first ? FILL_FIRST : 0) | (fin && S.kind == 2 ? FILL_NEGATE : 0), 0, 0};
              if (out) out->jobs.push_back(last);
              tl.kcur += len;
              inv_units += len;
              if (fin) --S.left;
            }
            if (n == 1 && out) {
              last.flags |= FILL_TWIN;
              out->jobs.push_back(last);
            }
            if (out) op.tab_cnt += 2;
          }
          if (S.left == 0) S.done_launch = k;
        }
      }
      if (out) out->ops.push_back(op);
    };
    auto update = [&](int kind, int first, int k1, int last_near, int newest) {
      // columns taking part: the block column `first` alone (narrow) or every column from `first` on that is not deferred
      // or is within `fill_near` panels of being factored (c <= last_near)
      CholOp op{kind, k1, first, out ? (int)out->colk0.size() : 0, nb, INT_MAX, INT_MIN, true, first, k1};
      std::vector<int> tab(nb, k1);
      const int cend = kind == 1 ? first + 1 : nb;
      for (int c = first; c < cend; ++c) {
        const bool active = c < far || c <= last_near;
        if (!active) { op.uniform = false; continue; }
        tab[c] = applied[c];
        op.last_active = c;
        if (c < far) op.k0_plain = applied[c];
        op.k0_min = std::min(op.k0_min, applied[c]);
        op.k0_max = std::max(op.k0_max, applied[c]);
        if (c >= far) catchup += (int64_t)tiles_of(c) * std::max(0, (k1 - applied[c]) - newest);   // beyond the newest panel(s)
        applied[c] = k1;
      }
      if (op.k0_min != op.k0_max) op.uniform = false;
      if (op.k0_min == INT_MAX) return;                                   // nothing to do
      if (out) {
        out->colk0.insert(out->colk0.end(), tab.begin(), tab.end());
        out->ops.push_back(op);
      }
    };
    for (int k = 0; k < nb;) {
      const int rem = nb - 1 - k;
      panel(k);
      // Update-bound steps go in PAIRS: panel k, block column k+1 <- its pending panels (narrow), panel k+1, then ONE
      // trailing pass with both panels (K = 256): half the passes over the trailing matrix and a tile kernel that runs
      // 15 % faster at K = 256 than at 128, for one narrow launch more on the chain.  Same bits (every element still
      // receives panel k before panel k+1).
      if (!tu.chol_legacy && tu.pair_min > 0 && rem >= 2 && (int64_t)B * rem * rem > tu.pair_min) {
        update(1, k + 1, k + 1, nb, 1);
        panel(k + 1);
        update(2, k + 2, k + 2, k + 1 + D, 2);
        k += 2;
      } else {
        if (rem > 0) update(2, k + 1, k + 1, k + D, 1);
        k += 1;
      }
    }
    if (finished) {
      finished->assign(st.size(), std::vector<char>());
      for (size_t i = 0; i < st.size(); ++i) {
        (*finished)[i].assign(st[i].tiles.size(), 0);
        for (size_t t = 0; t < st[i].tiles.size(); ++t) (*finished)[i][t] = st[i].tiles[t].kcur >= st[i].tiles[t].kend;
      }
    }
    if (out) {
      out->far_start = far;
      out->deferred_units = deferred;
      out->catchup_units = catchup;
      out->inv_units = inv_units;
    }
    return std::make_pair(deferred, catchup);
  };
  int far = nb;
  if (fill) {
    for (int f = 1; f < nb; ++f) {
      const auto dc = build(f, nullptr, nullptr, nullptr, false);
      if (dc.first > 0 && dc.second * tu.fill_slack <= dc.first) { far = f; break; }
    }
  }
  // A TILE is either finished inside the factorisation or left to the inverse's own launches as a whole (they start a tile
  // from zero): simulate, keep the tiles that finished, simulate again with only those until nothing changes
  TileFlags keep, fin;
  if (inv) {
    build(far, nullptr, nullptr, &fin, true);
    for (int round = 0; round < 6; ++round) {
      keep = fin;
      build(far, &keep, nullptr, &fin, true);
      bool same = true;
      for (size_t i = 0; i < fin.size(); ++i)
        for (size_t t = 0; t < fin[i].size(); ++t) {
          fin[i][t] = fin[i][t] && keep[i][t];
          same = same && fin[i][t] == keep[i][t];
        }
      if (same) break;
    }
    keep = fin;
  }
  CholPlan& pl = chol_plans[key];
  TileFlags fin2;
  build(far, inv ? &keep : nullptr, &pl, inv ? &fin2 : nullptr, inv);
  if (inv) {
    // What the inverse's own launches still have to do, per depth: the problems with an unfinished tile in their T (R)
    // stage, and a mask of the tiles that are done ([tile column][tile row] per problem, at the list's tile offsets).
    // (A kept tile that did not finish after all - it cannot - is simply not masked: it runs again from zero.)
    int maxd = -1;
    for (int dd : flat_depth) maxd = std::max(maxd, dd);
    pl.rest.assign(maxd + 1, CholPlan::Rest{0, 0, 0, 0, 0, 0, 0, 0});
    for (int dd = 0; dd <= maxd; ++dd) {
      CholPlan::Rest& r = pl.rest[dd];
      for (int pass = 1; pass <= 2; ++pass) {
        (pass == 1 ? r.first_t : r.first_r) = (int)pl.rest_probs.size();
        (pass == 1 ? r.skip_t : r.skip_r) = (int)pl.rest_skip.size();
        int off = 0, cnt = 0;
        for (size_t q = 0; q < flat.size(); ++q) {
          if (flat_depth[q] != dd) continue;
          const size_t si = 2 * q + (pass - 1);
          bool all_done = true;
          for (char c : fin2[si]) all_done = all_done && c;
          if (all_done) continue;
          pl.rest_probs.push_back({flat[q].lo, flat[q].mid, flat[q].hi, off});
          // stages[si].tiles were pushed tile column by tile column, rows inside: the mask's order
          for (char c : fin2[si]) pl.rest_skip.push_back((unsigned char)(c ? 1 : 0));
          off += (flat[q].hi - flat[q].mid) * (flat[q].mid - flat[q].lo);
          ++cnt;
        }
        (pass == 1 ? r.count_t : r.count_r) = cnt;
        (pass == 1 ? r.nblocks_t : r.nblocks_r) = off;
      }
    }
    for (const Stage& S : stages)
      for (const Tile& t : S.tiles) pl.inv_units_total += t.kend - t.kbeg;
    pl.inverse_started = pl.inv_units > 0;
    if (pl.inverse_started && !pl.rest_probs.empty()) {
      pl.d_rest.ensure(pl.rest_probs.size() * sizeof(TriProb));
      HIPCHK(hipMemcpy(pl.d_rest.p, pl.rest_probs.data(), pl.rest_probs.size() * sizeof(TriProb), hipMemcpyHostToDevice));
      pl.d_skip.ensure(pl.rest_skip.size());
      HIPCHK(hipMemcpy(pl.d_skip.p, pl.rest_skip.data(), pl.rest_skip.size(), hipMemcpyHostToDevice));
    }
  }
  for (const FillJob& j : pl.jobs) {           // (the tables drive device addresses: check them on the host)
    bool ok = true;
    if (j.kind == 0) ok = j.ti >= j.tj && j.ti < 2 * nb && j.tj >= 0 && j.k0 >= 0 && j.k1 > j.k0 && j.k1 <= j.tj - (j.tj & 1);
    else if (j.kind == 1 || j.kind == 2) ok = j.ti > j.tj && j.ti < 2 * nb && j.tj >= 0 && j.k0 >= 0 && j.k1 > j.k0 && j.k1 <= 2 * nb;
    else if (j.kind == 3) ok = j.ti >= 0 && j.ti < nb;
    else ok = j.kind == 4;
    if (!ok) throw Err(BOBE_ERR_STATE, "internal error: filler job outside the matrix");
  }
  if (!pl.jobs.empty()) {
    pl.d_jobs.ensure(pl.jobs.size() * sizeof(FillJob));
    HIPCHK(hipMemcpy(pl.d_jobs.p, pl.jobs.data(), pl.jobs.size() * sizeof(FillJob), hipMemcpyHostToDevice));
  }
  bool need_tab = false;                      // (a plan without deferred columns needs no device table: its launches are
  for (const CholOp& op : pl.ops) need_tab = need_tab || (op.kind != 0 && !op.uniform);   // uniform - and it may be built
  if (need_tab) {                             //  while a slot's stream is capturing, where allocations are not allowed)
    pl.d_colk0.ensure(pl.colk0.size() * sizeof(int));
    HIPCHK(hipMemcpy(pl.d_colk0.p, pl.colk0.data(), pl.colk0.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (std::getenv("BOBE_TRACE"))
    std::fprintf(stderr, "[bobe] chol plan nb=%d B=%d fill=%d inv=%d: deferred columns from %d, %lld tile-panels in fillers, %lld caught up; "
                 "inverse: %lld of %lld tile-units inside the factorisation (%zu jobs in all)\n",
                 nb, B, (int)fill, (int)inv, pl.far_start, (long long)pl.deferred_units, (long long)pl.catchup_units,
                 (long long)pl.inv_units, (long long)pl.inv_units_total, pl.jobs.size());
  return pl;
}

// Blocked right-looking Cholesky (NB = 128) of B matrices in lock step on one stream; every launch carries the slot
// in its last grid dimension.  Per step k:
//   panel k   diagonal factor + solve of the rows below.  One launch (k_chol_panel: every 64-row workgroup factors
//             the diagonal block itself) while all B * 2 * (nb-1-k) workgroups fit on the chip at once, else the
//             k_potf2 + k_trsm_panel pair (one factorisation per slot).
//   update    A22 -= L21 L21^T on the lower tiles (k_syrk_trail).
//             Update-bound steps go in pairs: one K = 256 pass for two panels (chol_plan).
// A batch shares the latency-bound panel chain (32 x ~28 us at N = 4096, the same for 1 or 8 matrices) and gives the
// update 4-8x the tiles: 41 % of the fp64 MFMA peak with four in flight, 49 % with eight, against 18 % alone and 28 %
// for four on private streams (whose 150 KB-LDS panel kernels wait for a CU the others' update tiles keep occupied).
// The panel launches leave most of the chip empty (one 150 KB workgroup per 64 rows): updates of block columns that are
// not needed soon are DEFERRED and ride in those launches as filler workgroups (chol_plan, k_chol_panel<., true>) - not on
// an evaluation slot's private stream, where the other slots' kernels want those CUs.
// Every matrix element sees the same operation sequence in all forms (same bits).
const bobe_gp::CholPlan* bobe_gp::potrf(double* a, double* linv, int* info_dev, int B, int64_t bsA, int64_t bsL, double* dg,
                                        bool defer_diag, double* tmp, int64_t bsT) {
  const Tuning& tu = tuning();
  if (!dg) dg = diag.d();                                     // scratch for the L_kk of the panel launches
  const int64_t bsD = (int64_t)nb * TILE * TILE;
  int first_aside = nb;                                       // first step whose L_kk was left in the scratch blocks
  // Where the fillers pay (measured, profiles/r03_fill_ab.txt): a lone factorisation of 3072..4608 points - its chain
  // leaves 3/4 of the chip empty and its update launches are latency-bound (N = 4096: 1.60 -> 1.44 ms).  Smaller: the
  // updates are a few microseconds each anyway; larger, or several matrices in lock step: the panel launches have few CUs
  // to spare, a filler workgroup (one per CU, eight waves) runs the tile core at ~3/4 of its usual rate, and a launch
  // lasts as long as its slowest filler (B = 4 at N = 4096: 2.83 -> 2.85 ms; N = 8192 alone: 5.6 -> 6.2 ms).
  // BOBE_FILL=2 forces them on everywhere, BOBE_FILL=0 off.
  const bool fill = tu.fill != 0 && !in_slot && !tu.chol_legacy && tu.filler_iters == 0 &&
                    (tu.fill == 2 || (B == 1 && nb >= 24 && nb <= 36));
  // The inverse that follows has work without a deadline - its diagonal blocks and the T / R tiles of its recursion depend
  // only on block rows that are final - so the same launches carry it as well (when the caller hands over the inverse's
  // scratch matrix): what finishes inside the factorisation is skipped by trtri().  Same tiles, same K order: same bits.
  static const bool fill_in_slots = std::getenv("BOBE_FILL_SLOTS") != nullptr;      // (experiment: not on a slot by default)
  const bool inv = tmp != nullptr && tu.fill_inv != 0 && (!in_slot || (fill_in_slots && N > tu.graph_max_n)) && !tu.chol_legacy &&
                   tu.filler_iters == 0 && (tu.fill_inv == 2 || (nb >= 16 && nb <= 48));   // (larger: update-bound, nothing to hide in)
  const CholPlan& pl = chol_plan(B, fill, inv);
  const FillJob* jobs = static_cast<const FillJob*>(pl.d_jobs.p);
  const int* coltab = static_cast<const int*>(pl.d_colk0.p);
  for (const CholOp& op : pl.ops) {
    if (op.kind == 0) {                                       // (64 rows of the panel per workgroup)
      const int kk = op.k;
      const int rr = nb - 1 - kk;
      const int strips = panel_strips(B, rr);
      const int np_ = panel_workgroups(rr, strips);
      const int rows_below = rr * TILE;
      const int nv = (int)std::min<int64_t>(TILE, N - (int64_t)kk * TILE);
      if (!tu.chol_legacy && B * np_ <= std::max(num_cus, 1)) {
        first_aside = std::min(first_aside, kk);
        prof_begin(BOBE_PROF_POTF2);
        const int standin = tu.filler_iters > 0 ? std::max(0, (num_cus - tu.filler_keep - B * np_) / B) : 0;
#define PANEL_LAUNCH(FILLV, GRIDX, ...)                                                                                        \
  do {                                                                                                                         \
    if (strips == 3)                                                                                                           \
      hipLaunchKernelGGL((k_chol_panel<false, FILLV, 3>), dim3(GRIDX, B), dim3(PANEL_THREADS), POTF2_SMEM_BYTES, stream, a, Np, \
                         bsA, linv, Np, bsL, kk, np_, info_dev, nv, dg, bsD, (unsigned long long*)nullptr, __VA_ARGS__);       \
    else                                                                                                                       \
      hipLaunchKernelGGL((k_chol_panel<false, FILLV, 4>), dim3(GRIDX, B), dim3(PANEL_THREADS), POTF2_SMEM_BYTES, stream, a, Np, \
                         bsA, linv, Np, bsL, kk, np_, info_dev, nv, dg, bsD, (unsigned long long*)nullptr, __VA_ARGS__);       \
  } while (0)
        if (op.tab_cnt > 0) {
          PANEL_LAUNCH(true, np_ + op.tab_cnt / 2, jobs + op.tab_off, op.tab_cnt, 0, (double*)nullptr, rows_below, tmp, Np, bsT);
        } else if (standin > 0) {
          filler_ws.ensure((size_t)B * (np_ + standin) * PANEL_THREADS * sizeof(double));
          PANEL_LAUNCH(true, np_ + standin, (const FillJob*)nullptr, 0, tu.filler_iters, filler_ws.d(), rows_below, (double*)nullptr,
                       (int64_t)0, (int64_t)0);
        } else {
          PANEL_LAUNCH(false, np_, (const FillJob*)nullptr, 0, 0, (double*)nullptr, rows_below, (double*)nullptr, (int64_t)0,
                       (int64_t)0);
        }
#undef PANEL_LAUNCH
        prof_end(BOBE_PROF_POTF2);
      } else {
        prof_begin(BOBE_PROF_POTF2);
        hipLaunchKernelGGL((k_potf2<true, false>), dim3(B), dim3(256), POTF2_SMEM_BYTES, stream, a, Np, linv, Np, kk,
                           info_dev, (unsigned long long*)nullptr, nv, bsA, bsL);
        prof_end(BOBE_PROF_POTF2);
        if (rr > 0) {
          prof_begin(BOBE_PROF_TRSM);
          hipLaunchKernelGGL(k_trsm_panel<false>, dim3(2 * rr, B), dim3(256), TRSM_SMEM_BYTES, stream, a, Np,
                             (const double*)linv, Np, kk, (unsigned long long*)nullptr, bsA, bsL);
          prof_end(BOBE_PROF_TRSM);
        }
      }
    } else {
      prof_begin(BOBE_PROF_SYRK);
      if (op.uniform) {
        syrk(a, op.k0_min, op.k, op.first, op.kind == 1 ? 1 : 0, B, bsA);
      } else {
        // (columns before far_start share one first panel and take it as a scalar; deferred ones read the table.  A launch
        // whose active columns are few enumerates just those)
        const int ncols = op.last_active - op.first + 1;
        const bool few = ncols <= 8 && ncols < nb - op.first;
        syrk(a, op.k0_plain, op.k, op.first, few ? 2 : 0, B, bsA, coltab + op.tab_off, pl.far_start, few ? ncols : 0);
      }
      prof_end(BOBE_PROF_SYRK);
    }
  }
  // (the scratch blocks of the k_chol_panel steps; k_potf2 steps wrote in place, and come first:
  // B * npanel and rem only shrink with k)
  aside_first = 1 << 30;
  aside_dg = nullptr;
  if (first_aside < nb) {
    if (defer_diag) {
      aside_first = first_aside;
      aside_dg = dg;
    } else {
      hipLaunchKernelGGL(k_copy_diag, dim3(nb - first_aside, B), dim3(256), 0, stream, a, Np, bsA, (const double*)dg, bsD,
                         first_aside);
    }
  }
  LAUNCH_CHECK();
  return pl.inverse_started ? &pl : nullptr;
}

// Linv = L^-1: diagonal 128-blocks in one batched launch, then recursive doubling (two GEMM launches per level)
void bobe_gp::trtri(double* a, double* linv, double* tmp, int B, int64_t bsA, int64_t bsL, int64_t bsT, const CholPlan* rest) {
  const Tuning& tu = tuning();
  prof_begin(BOBE_PROF_TRTRI);
  hipLaunchKernelGGL(k_trti_diag, dim3(nb, B), dim3(256), POTF2_SMEM_BYTES, stream, a, Np, linv, Np, bsA, bsL, aside_dg,
                     (int64_t)nb * TILE * TILE, aside_first);
  aside_first = 1 << 30;
  aside_dg = nullptr;
  prof_end(BOBE_PROF_TRTRI);
  // (part of the recursion may have run inside the factorisation's panel launches: then only the stages it left)
  for (int dd = (int)depths.size() - 1; dd >= 0; --dd) {
    const Depth& D = depths[dd];
    const TriProb* pr_t = static_cast<const TriProb*>(probs.p) + D.first;
    const TriProb* pr_r = pr_t;
    int cnt_t = D.count, cnt_r = D.count, nbl_t = D.nblocks, nbl_r = D.nblocks;
    const unsigned char *sk_t = nullptr, *sk_r = nullptr;
    if (rest) {
      const CholPlan::Rest& r = rest->rest[dd];
      pr_t = static_cast<const TriProb*>(rest->d_rest.p) + r.first_t;
      pr_r = static_cast<const TriProb*>(rest->d_rest.p) + r.first_r;
      cnt_t = r.count_t; cnt_r = r.count_r; nbl_t = r.nblocks_t; nbl_r = r.nblocks_r;
      sk_t = static_cast<const unsigned char*>(rest->d_skip.p) + r.skip_t;
      sk_r = static_cast<const unsigned char*>(rest->d_skip.p) + r.skip_r;
    }
    prof_begin(BOBE_PROF_TRTRI);
    // (64x64 tiles while a level of ONE matrix has too few 128x128 tiles to fill the chip; a tile's K order is the
    // same either way.  Batches keep the per-matrix choice: four in lock step at N = 4096 take 7.0 ms per evaluation
    // round with 64x64 tiles at every level against 7.4 with 128x128 tiles at the top level)
    if (D.nblocks < tu.trtri64_below) {
      if (cnt_t > 0) {
        const TileGrid tg = tile_grid(2 * nbl_t);             // (tile pairs of complementary K: equal work)
        hipLaunchKernelGGL(k_trtri_T<64>, dim3(tg.grid, B), dim3(256), GEMM64_SMEM_BYTES, stream, a, Np,
                           (const double*)linv, Np, tmp, Np, pr_t, cnt_t, bsA, bsL, bsT, tg.per, sk_t);
      }
      if (cnt_r > 0) {
        const TileGrid tg = tile_grid(2 * nbl_r);
        hipLaunchKernelGGL(k_trtri_R<64>, dim3(tg.grid, B), dim3(256), GEMM64_SMEM_BYTES, stream, linv, Np,
                           (const double*)tmp, Np, pr_r, cnt_r, bsL, bsT, tg.per, sk_r);
      }
    } else {
      if (cnt_t > 0)
        hipLaunchKernelGGL(k_trtri_T<128>, dim3(nbl_t, B), dim3(256), GEMM_SMEM_BYTES, stream, a, Np,
                           (const double*)linv, Np, tmp, Np, pr_t, cnt_t, bsA, bsL, bsT);
      if (cnt_r > 0)
        hipLaunchKernelGGL(k_trtri_R<128>, dim3(nbl_r, B), dim3(256), GEMM_SMEM_BYTES, stream, linv, Np,
                           (const double*)tmp, Np, pr_r, cnt_r, bsL, bsT);
    }
    prof_end(BOBE_PROF_TRTRI);
  }
  LAUNCH_CHECK();
}

// K^-1 tiles fused with the gradient partial sums (optionally stores K^-1's lower tiles); returns #partials
int bobe_gp::lauum(const Hyper& h, const double* linv, const double* al, const double* xst, double* kinv_out, int dcap,
                   const Hyper* hdev, double* gp_out, int B, int64_t bsL, int64_t bsV, int64_t bsX, int64_t bsP) {
  const Tuning& tu = tuning();
  // (the tile size fixes the order of the gradient's partial sums: it depends on N only, so that an evaluation
  // returns the same bits alone, on a slot and in a batch)
  const bool small = nb * (nb + 1) / 2 < tu.lauum64_below;
  const int nt = small ? 2 * nb : nb;
  const int ntiles = nt * (nt + 1) / 2;
  double* gpo = gp_out ? gp_out : gpart.d();
#define LG(KE, DC, TT)                                                                                          \
  hipLaunchKernelGGL((k_lauum_grad<KE, DC, TT>), dim3(ntiles, B), dim3(256),                                    \
                     (TT == 128 ? GEMM_SMEM_BYTES : GEMM64_SMEM_BYTES), stream, linv, Np, Np, N, al, xst, Np, h, \
                     gpo, kinv_out, Np, hdev, bsL, bsV, bsX, bsP)
#define LGD(KE, TT)                                                                 \
  do {                                                                              \
    if (dcap == 8) LG(KE, 8, TT); else if (dcap == 16) LG(KE, 16, TT); else LG(KE, 32, TT); \
  } while (0)
  prof_begin(BOBE_PROF_LAUUM);
  if (h.kern == 0) {
    if (small) LGD(0, 64); else LGD(0, 128);
  } else {
    if (small) LGD(1, 64); else LGD(1, 128);
  }
  prof_end(BOBE_PROF_LAUUM);
#undef LGD
#undef LG
  LAUNCH_CHECK();
  return ntiles;
}

void bobe_gp::solve_alpha(const double* linv, double* wv, double* al, double* prt, int B, int64_t bsL, int64_t bsV,
                          int64_t bsP) {
  hipLaunchKernelGGL(k_gemv_lower, dim3((unsigned)(Np / 4), (unsigned)B), dim3(256), 0, stream, linv, Np, Np,
                     (const double*)y.d(), wv, bsL, bsV);
  hipLaunchKernelGGL(k_gemv_t_part, dim3((unsigned)(Np / 64), (unsigned)nb, (unsigned)B), dim3(256), 0, stream, linv, Np, 1,
                     (const double*)wv, prt, Np, bsL, bsV, bsP);
  hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((Np + 255) / 256), (unsigned)B), dim3(256), 0, stream,
                     (const double*)prt, Np, nb, 1, Np, al, bsP, bsV);
  LAUNCH_CHECK();
}

void bobe_gp::factor_into(const Hyper& h, double* xst, double* a, double* linv, double* wv, double* al,
                          const Hyper* hdev) {
  scale(X.d(), N, Np, h, xst, Np, hdev, 1, 0, static_cast<int*>(info.p));
  assemble_kxx(h, xst, a, hdev);
  const CholPlan* rest = potrf(a, linv, static_cast<int*>(info.p), 1, 0, 0, nullptr, true, Tmp.d(), 0);
  trtri(a, linv, Tmp.d(), 1, 0, 0, 0, rest);
  solve_alpha(linv, wv, al, part.d());
}

int bobe_gp::read_info() {
  int v = 0;
  HIPCHK(hipMemcpyAsync(&v, info.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  sync();
  return v;
}

void bobe_gp::ensure_slots(int n) {
  if (!ev_batch) HIPCHK(hipEventCreateWithFlags(&ev_batch, hipEventDisableTiming));
  while ((int)slots.size() < n) {
    Slot* sl = new Slot();
    slots.push_back(sl);
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sl->h_res), 128 * sizeof(double), hipHostMallocDefault));
  }
  const size_t mat = (size_t)Np * Np * sizeof(double), vec = (size_t)Np * sizeof(double);
  for (int i = 0; i < n; ++i) {
    Slot& sl = *slots[i];
    sl.A2.ensure(mat); sl.Linv2.ensure(mat); sl.Tmp.ensure(mat);
    sl.alpha2.ensure(vec); sl.w2.ensure(vec); sl.XsT2.ensure((size_t)d * vec);
    sl.part.ensure((size_t)nb * Np * sizeof(double));
    sl.gpart.ensure((size_t)(2 * nb) * (2 * nb + 1) / 2 * (MAX_D + 1) * sizeof(double));
    sl.res.ensure(128 * sizeof(double));
    sl.info.ensure(sizeof(int));
    sl.flags.ensure((size_t)nb * sizeof(int));
    sl.diag.ensure((size_t)nb * TILE * TILE * sizeof(double));
  }
}

// The slots' streams are made through the CU-mask entry point with every CU enabled: such a stream gets a
// hardware queue of its own, which plain streams (multiplexed on a few queues) do not - 34 vs 42 ms for the
// 20-evaluation fit at N = 4096.  (Real CU partitions were measured and dropped, DESIGN.md.)
const std::vector<hipStream_t>& bobe_gp::slot_stream_set() {
  std::vector<hipStream_t>& v = slot_streams;
  if (!v.empty()) return v;
  const int words = (num_cus + 31) / 32;
  for (int i = 0; i < BOBE_MAX_MLL_SLOTS; ++i) {
    hipStream_t st = nullptr;
    if (tuning().own_queues) {
      std::vector<uint32_t> mask(words, 0u);
      for (int c = 0; c < num_cus; ++c) mask[c / 32] |= (1u << (c % 32));
      if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        st = nullptr;
      }
    }
    if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    v.push_back(st);
  }
  return v;
}

// value (+ gradient) pipeline of one hyper-parameter vector on the current stream / workspace; results land in
// the pinned h_res: [0] y^T K^-1 y, [1] sum log L_ii, [2..2+d] gradient, [100] the factorisation's info word
void bobe_gp::mll_enqueue_body(const Hyper& h, bool want_grad, const Hyper* hdev) {
  if (hdev) HIPCHK(hipMemcpyAsync(eg.hyp_dev.p, eg.h_hyp, sizeof(Hyper), hipMemcpyHostToDevice, stream));
  factor_into(h, XsT2.d(), A2.d(), Linv2.d(), w2.d(), alpha2.d(), hdev);
  if (want_grad) {
    const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
    const int ntiles = lauum(h, Linv2.d(), alpha2.d(), XsT2.d(), nullptr, dcap, hdev);
    hipLaunchKernelGGL(k_mll_grad_reduce, dim3(d + 2), dim3(256), 0, stream, (const double*)gpart.d(), ntiles, dcap + 1, d,
                       dcap, res.d(), (const double*)w2.d(), (const double*)A2.d(), Np, Np, (const int*)info.p);
  } else {
    hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)w2.d(), (const double*)A2.d(), Np, Np,
                       res.d(), (int64_t)0, (int64_t)0, (int64_t)0, (const int*)info.p);
  }
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 101 * sizeof(double), hipMemcpyDeviceToHost, stream));   // [100] = info (k_mll_terms)
}

void bobe_gp::mll_enqueue(const Hyper& h, bool want_grad) {
  const Tuning& tu = tuning();
  // Up to graph_max_n points an evaluation is tens of kernels of a few microseconds each, and with several slots
  // in flight the host cannot enqueue them as fast as the GPU retires them: a slot replays its pipeline as one
  // graph (N = 64 / 512 / 2048 with four in flight: 42 / 107 / 419 us per evaluation instead of 66 / 141 / 553).
  // A lone evaluation on the handle's stream is NOT faster as a graph (125 vs 107 us at N = 64) and stays a
  // plain launch sequence; so does everything while a kernel class is being timed (events are not captured).
  if (!in_slot || N > tu.graph_max_n || prof_tag != 0) {
    mll_enqueue_body(h, want_grad, nullptr);
    return;
  }
  if (!eg.h_hyp) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&eg.h_hyp), sizeof(Hyper), hipHostMallocDefault));
  eg.hyp_dev.ensure(sizeof(Hyper));
  const int w = want_grad ? 1 : 0;
  const std::array<const void*, 16> sig = eval_signature();
  if (!eg.exec[w] || eg.sig[w] != sig) {     // first use, or N / a buffer changed since the capture
    if (eg.exec[w]) {
      (void)hipGraphExecDestroy(eg.exec[w]);
      eg.exec[w] = nullptr;
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    try {
      mll_enqueue_body(h, want_grad, static_cast<const Hyper*>(eg.hyp_dev.p));
    } catch (...) {
      (void)hipStreamEndCapture(stream, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    HIPCHK(hipStreamEndCapture(stream, &graph));
    const hipError_t ie = hipGraphInstantiate(&eg.exec[w], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) {
      eg.exec[w] = nullptr;
      HIPCHK(ie);
    }
    eg.sig[w] = sig;
  }
  *eg.h_hyp = h;            // read by the graph's first node when it executes; the caller collects before reusing it
  HIPCHK(hipGraphLaunch(eg.exec[w], stream));
}

int bobe_gp::slot_collect(Slot& sl, double* mll, double* grad) {
  // touches only the slot's own stream and pinned results: safe while another thread submits to another slot
  HIPCHK(hipStreamSynchronize(sl.stream));
  const double* hr = sl.h_res;
  int inf;
  std::memcpy(&inf, hr + 100, sizeof(int));
  if (inf != 0x7f7f7f7f) {
    *mll = std::nan("");
    if (grad)
      for (int j = 0; j <= d; ++j) grad[j] = std::nan("");
    g_err = "kernel matrix not positive definite at column " + std::to_string(inf - 1);
    return BOBE_NOT_PD;
  }
  *mll = -0.5 * hr[0] - hr[1] - 0.5 * (double)N * std::log(2.0 * M_PI);
  if (grad)
    for (int j = 0; j <= d; ++j) grad[j] = hr[2 + j];
  return BOBE_OK;
}

int bobe_gp::mll_collect(double* mll, double* grad) {
  sync();
  int inf;
  std::memcpy(&inf, h_res + 100, sizeof(int));
  if (inf != 0x7f7f7f7f) {
    *mll = std::nan("");
    if (grad)
      for (int j = 0; j <= d; ++j) grad[j] = std::nan("");
    g_err = "kernel matrix not positive definite at column " + std::to_string(inf - 1);
    return BOBE_NOT_PD;
  }
  *mll = -0.5 * h_res[0] - h_res[1] - 0.5 * (double)N * std::log(2.0 * M_PI);
  if (grad)
    for (int j = 0; j <= d; ++j) grad[j] = h_res[2 + j];
  return BOBE_OK;
}

void bobe_gp::ensure_batch(int B) {
  if (!bw.h_hyp) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bw.h_hyp), BOBE_MAX_MLL_SLOTS * sizeof(Hyper), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bw.h_res), BOBE_MAX_MLL_SLOTS * 128 * sizeof(double), hipHostMallocDefault));
  }
  const size_t mat = (size_t)Np * Np * sizeof(double), vec = (size_t)Np * sizeof(double);
  const size_t nB = (size_t)B;
  bw.A.ensure(nB * mat); bw.Linv.ensure(nB * mat); bw.Tmp.ensure(nB * mat);
  bw.XsT.ensure(nB * d * vec); bw.w.ensure(nB * vec); bw.alpha.ensure(nB * vec);
  bw.part.ensure(nB * nb * vec);
  bw.gpart.ensure(nB * (size_t)gpart_stride() * sizeof(double));
  bw.res.ensure(nB * 128 * sizeof(double));
  bw.info.ensure(BOBE_MAX_MLL_SLOTS * sizeof(int));
  bw.diag.ensure(nB * (size_t)nb * TILE * TILE * sizeof(double));
  bw.hyp.ensure(BOBE_MAX_MLL_SLOTS * sizeof(Hyper));
  bw.cap = std::max(bw.cap, B);
  bw.Np = Np;
}

// B value(+gradient) evaluations in lock step: the pipeline of mll_enqueue_body with every launch widened by the
// slot dimension.  Results land in the pinned bw.h_res[b*128 + ...] (layout of mll_enqueue_body, the info word at [100]).
void bobe_gp::mll_lockstep_enqueue(int B, const Hyper* hs, bool want_grad) {
  ensure_batch(B);
  const int64_t mat = Np * Np, vec = Np, xs = (int64_t)d * Np, prt = (int64_t)nb * Np, gps = gpart_stride();
  for (int b = 0; b < B; ++b) bw.h_hyp[b] = hs[b];
  HIPCHK(hipMemcpyAsync(bw.hyp.p, bw.h_hyp, (size_t)B * sizeof(Hyper), hipMemcpyHostToDevice, stream));
  const Hyper* hdev = static_cast<const Hyper*>(bw.hyp.p);
  int* inf = static_cast<int*>(bw.info.p);
  scale(X.d(), N, Np, hs[0], bw.XsT.d(), Np, hdev, B, xs, inf);      // (also arms the B info words)
  assemble_kxx(hs[0], bw.XsT.d(), bw.A.d(), hdev, B, xs, mat);
  const CholPlan* rest = potrf(bw.A.d(), bw.Linv.d(), inf, B, mat, mat, bw.diag.d(), true, bw.Tmp.d(), mat);
  trtri(bw.A.d(), bw.Linv.d(), bw.Tmp.d(), B, mat, mat, mat, rest);
  solve_alpha(bw.Linv.d(), bw.w.d(), bw.alpha.d(), bw.part.d(), B, mat, vec, prt);
  // (the info word of slot b rides in res[b * 128 + 100]: one copy brings everything to the host)
  if (want_grad) {
    const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
    const int ntiles = lauum(hs[0], bw.Linv.d(), bw.alpha.d(), bw.XsT.d(), nullptr, dcap, hdev, bw.gpart.d(), B, mat, vec,
                             xs, gps);
    hipLaunchKernelGGL(k_mll_grad_reduce, dim3(d + 2, B), dim3(256), 0, stream, (const double*)bw.gpart.d(), ntiles, dcap + 1,
                       d, dcap, bw.res.d(), (const double*)bw.w.d(), (const double*)bw.A.d(), Np, Np, (const int*)inf, gps,
                       (int64_t)128, vec, mat);
  } else {
    hipLaunchKernelGGL(k_mll_terms, dim3(B), dim3(256), 0, stream, (const double*)bw.w.d(), (const double*)bw.A.d(), Np, Np,
                       bw.res.d(), vec, mat, (int64_t)128, (const int*)inf);
  }
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(bw.h_res, bw.res.p, (size_t)B * 128 * sizeof(double), hipMemcpyDeviceToHost, stream));
}

int bobe_gp::mll_lockstep_collect(int B, double* mll, double* grad, int* status) {
  sync();
  int worst = BOBE_OK;
  for (int b = 0; b < B; ++b) {
    const double* hr = bw.h_res + (size_t)b * 128;
    double* gb = grad ? grad + (size_t)b * (d + 1) : nullptr;
    int st = BOBE_OK;
    int inf_b;
    std::memcpy(&inf_b, hr + 100, sizeof(int));
    if (inf_b != 0x7f7f7f7f) {
      mll[b] = std::nan("");
      if (gb)
        for (int j = 0; j <= d; ++j) gb[j] = std::nan("");
      g_err = "kernel matrix not positive definite at column " + std::to_string(inf_b - 1);
      st = BOBE_NOT_PD;
      worst = st;
    } else {
      mll[b] = -0.5 * hr[0] - hr[1] - 0.5 * (double)N * std::log(2.0 * M_PI);
      if (gb)
        for (int j = 0; j <= d; ++j) gb[j] = hr[2 + j];
    }
    if (status) status[b] = st;
  }
  return worst;
}

// Z-side quantities of the sweep: ZsT, kXZ, V_Z = Linv kXZ, base_z = kself - |V_Z[:,z]|^2, W_Z = Linv^T V_Z
void bobe_gp::prepare_z(const double* Z, int64_t M, int64_t Mp) {
  const bool host_z = !is_device_ptr(Z);
  if (host_z && z_seen_m == M && std::memcmp(z_seen.data(), Z, (size_t)M * d * sizeof(double)) == 0) return;
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
  if ((Mp / TILE) * nb < 2 * std::max(num_cus, 1)) {
    const int nt = 2 * nb;
    hipLaunchKernelGGL(k_trimul_v64, dim3((unsigned)(Mp / 64), (unsigned)nt), dim3(256), GEMM64_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nt, (const double*)kXZ.d(), Mp, VZ.d(), Mp, qpart.d(), Mp);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), Mp, nt, Mp, hyp.kvar + hyp.noise, 0, basez.d(), (double*)nullptr);
    hipLaunchKernelGGL(k_trimul_t64, dim3((unsigned)(Mp / 64), (unsigned)nt), dim3(256), GEMM64_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nt, (const double*)VZ.d(), Mp, WZ.d(), Mp);
  } else {
    hipLaunchKernelGGL(k_trimul, dim3((unsigned)(Mp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)kXZ.d(), Mp, VZ.d(), Mp, qpart.d(), Mp,
                       (const double*)nullptr, (int64_t)0, 0, (double*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, stream,
                       (const double*)qpart.d(), Mp, nb, Mp, hyp.kvar + hyp.noise, 0, basez.d(), (double*)nullptr);
    hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(Mp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, stream,
                       (const double*)Linv.d(), Np, nb, (const double*)VZ.d(), Mp, WZ.d(), Mp);
  }
  LAUNCH_CHECK();
  if (host_z) z_seen_m = M;
}

void bobe_gp::sweep(const double* cand, int64_t C, const double* Z, int64_t M, double y_std, double* wipv,
                    double* wipstd, double* mean, double* var, int policy, int64_t* argmin_v, double* min_v,
                    int64_t* argmin_s, double* min_s, double* fantasy_out) {
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
  if (do_wip) prepare_z(Z, M, Mp);
  const int64_t CH = chunk;
  // scoring runs once per super-chunk of SC candidates (bounded crossT workspace: Mp x SC doubles)
  const int64_t SC = round_up(std::min<int64_t>(C, std::max<int64_t>(CH, 65536)), CH);
  CsT.ensure((size_t)d * SC * sizeof(double));
  kXC.ensure((size_t)Np * CH * sizeof(double));
  sc.ensure((size_t)SC * sizeof(double));
  qpart.ensure((size_t)nb * (Mp > CH ? Mp : CH) * sizeof(double));
  part.ensure((size_t)nb * (Np > CH ? Np : CH) * sizeof(double));
  if (do_wip) pv.ensure((size_t)Mp * SC * sizeof(double));   // crossT
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
  // BOBE_SWEEP_OVERLAP=1 (experiment, off by default), two chunks or more: the VALU-bound front of a chunk (K(X, chunk)
  // with the posterior-mean products) runs on a second stream under the MFMA-bound GEMM launch of the chunk before it; the
  // GEMM launches themselves stay in order on the handle's stream.  K(X, chunk) is double-buffered; events: [0] inputs ready, [1] scoring done with CsT,
  // [2 + b] buffer b assembled, [4 + b] buffer b consumed.  Same kernels on the same data: same bits.
  const bool overlap = tuning().sweep_overlap != 0 && C > CH;
  hipStream_t aux = nullptr;
  double* kxc[2] = {kXC.d(), kXC.d()};
  if (overlap) {
    aux = slot_stream_set()[0];
    for (hipEvent_t& e : ev_sw)
      if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    kXC2.ensure((size_t)Np * CH * sizeof(double));
    part_aux.ensure((size_t)nb * CH * sizeof(double));
    kxc[1] = kXC2.d();
    HIPCHK(hipEventRecord(ev_sw[0], stream));          // (after the candidates' upload and everything the factor needs)
    HIPCHK(hipStreamWaitEvent(aux, ev_sw[0], 0));
  }
  struct StreamSwap {                                  // the launch helpers use the member `stream`
    hipStream_t& a;
    hipStream_t& b;
    bool on;
    StreamSwap(hipStream_t& a_, hipStream_t& b_, bool on_) : a(a_), b(b_), on(on_) { if (on) std::swap(a, b); }
    ~StreamSwap() { if (on) std::swap(a, b); }
  };
  int64_t ci = 0;                                      // chunk counter (buffer parity)
  for (int64_t s0 = 0; s0 < C; s0 += SC) {
    const int64_t ns = (C - s0 < SC) ? (C - s0) : SC;
    const int64_t nsp = round_up(ns, TILE);
    {
      StreamSwap sw(stream, aux, overlap);
      if (overlap && s0 > 0) HIPCHK(hipStreamWaitEvent(stream, ev_sw[1], 0));   // the scorer read the previous CsT
      scale(cin + s0 * d, ns, nsp, hyp, CsT.d(), SC);
    }
    for (int64_t c0 = 0; c0 < ns; c0 += CH, ++ci) {
      const int64_t nc = (ns - c0 < CH) ? (ns - c0) : CH;
      const int64_t ncp = round_up(nc, TILE);
      const int b = overlap ? (int)(ci & 1) : 0;
      {
        StreamSwap sw(stream, aux, overlap);
        if (overlap && ci >= 2) HIPCHK(hipStreamWaitEvent(stream, ev_sw[4 + b], 0));
        // (posterior mean: the assembly leaves K(X, chunk)^T alpha per row tile on the way, k_gemv_t_part's partial sums)
        double* pm = overlap ? part_aux.d() : part.d();
        kernel_matrix_cross(XsT.d(), Np, N, Np, CsT.d() + c0, SC, nc, ncp, hyp, kxc[b], CH,
                            d_mean ? (const double*)alpha.d() : nullptr, d_mean ? pm : nullptr, CH);
        if (d_mean) {
          hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                             (const double*)pm, CH, nb, 0, nc, d_mean + s0 + c0);
        }
        if (overlap) HIPCHK(hipEventRecord(ev_sw[2 + b], stream));
      }
      if (overlap) HIPCHK(hipStreamWaitEvent(stream, ev_sw[2 + b], 0));
      prof_begin(BOBE_PROF_TRIMUL);
      hipLaunchKernelGGL(k_trimul, dim3((unsigned)(ncp / TILE), (unsigned)(nb + nzt)), dim3(256), GEMM_SMEM_BYTES,
                         stream, (const double*)Linv.d(), Np, nb, (const double*)kxc[b], CH, (double*)nullptr,
                         (int64_t)0, qpart.d(), CH, (const double*)WZ.d(), Mp, nzt, do_wip ? pv.d() + c0 : nullptr, SC);
      prof_end(BOBE_PROF_TRIMUL);
      if (overlap) HIPCHK(hipEventRecord(ev_sw[4 + b], stream));
      // s_c for the scorer, var for the caller
      hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream,
                         (const double*)qpart.d(), CH, nb, nc, kself, policy, sc.d() + c0,
                         d_var ? d_var + s0 + c0 : nullptr);
      LAUNCH_CHECK();
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
    if (overlap && s0 + SC < C) HIPCHK(hipEventRecord(ev_sw[1], stream));
  }
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

// -------------------------------------------------------------------------------------------------
// extern "C"
// -------------------------------------------------------------------------------------------------
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

extern "C" {

const char* bobe_version(void) { return "bobe_gp 0.1.0 gfx950"; }
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
  if (!out) throw Err(BOBE_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (d < 1 || d > MAX_D) throw Err(BOBE_ERR_ARG, "d must be in [1, 32]");
  if (kernel != BOBE_KERNEL_RBF && kernel != BOBE_KERNEL_MATERN) throw Err(BOBE_ERR_ARG, "unknown kernel id");
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
    {
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, device));
      g->num_cus = prop.multiProcessorCount;
    }
  } catch (...) {
    delete g;
    throw;
  }
  *out = g;
  return BOBE_OK;
  API_END
}

void bobe_gp_destroy(bobe_gp_t* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  DBuf* bufs[] = {&g->X, &g->y, &g->XsT, &g->XsT2, &g->A, &g->Linv, &g->A2, &g->Linv2, &g->Tmp, &g->alpha, &g->w,
                  &g->alpha2, &g->w2, &g->part, &g->gpart, &g->res, &g->info, &g->probs, &g->flags, &g->diag, &g->in_stage, &g->z_stage,
                  &g->CsT, &g->ZsT, &g->kXC, &g->kXZ, &g->VZ, &g->WZ, &g->basez, &g->sc, &g->qpart, &g->pv, &g->ps,
                  &g->o_mean, &g->o_var, &g->o_wipv, &g->o_wipstd, &g->o_misc, &g->kin_a, &g->kin_b, &g->kout, &g->wg_ws, &g->filler_ws,
                  &g->kXC2, &g->part_aux};
  for (DBuf* b : bufs) b->release();
  for (auto& pr : g->prof_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (g->h_res) (void)hipHostFree(g->h_res);
  for (hipStream_t st : g->slot_streams) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  auto free_eg = [](bobe_gp::EvalGraph& e) {
    for (int w = 0; w < 2; ++w)
      if (e.exec[w]) (void)hipGraphExecDestroy(e.exec[w]);
    if (e.h_hyp) (void)hipHostFree(e.h_hyp);
    e.hyp_dev.release();
  };
  free_eg(g->eg);
  for (bobe_gp::Slot* sl : g->slots) {
    free_eg(sl->eg);
    DBuf* sb[] = {&sl->XsT2, &sl->A2, &sl->Linv2, &sl->Tmp, &sl->alpha2, &sl->w2, &sl->part, &sl->gpart, &sl->res,
                  &sl->info, &sl->flags, &sl->diag};
    for (DBuf* b : sb) b->release();
    if (sl->h_res) (void)hipHostFree(sl->h_res);
    if (sl->ev) (void)hipEventDestroy(sl->ev);
    delete sl;
  }
  {
    DBuf* bb[] = {&g->bw.A, &g->bw.Linv, &g->bw.Tmp, &g->bw.XsT, &g->bw.w, &g->bw.alpha, &g->bw.part, &g->bw.gpart,
                  &g->bw.res, &g->bw.info, &g->bw.hyp, &g->bw.diag};
    for (DBuf* b : bb) b->release();
    if (g->bw.h_hyp) (void)hipHostFree(g->bw.h_hyp);
    if (g->bw.h_res) (void)hipHostFree(g->bw.h_res);
  }
  for (auto& kv : g->chol_plans) {
    kv.second.d_jobs.release();
    kv.second.d_colk0.release();
    kv.second.d_rest.release();
    kv.second.d_skip.release();
  }
  if (g->ev_batch) (void)hipEventDestroy(g->ev_batch);
  for (hipEvent_t e : g->ev_sw)
    if (e) (void)hipEventDestroy(e);
  if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

void* bobe_gp_get_stream(bobe_gp_t* g) { return g ? (void*)g->stream : nullptr; }

int bobe_gp_set_stream(bobe_gp_t* g, void* s) {
  API_BEGIN
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
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
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
  g->use();
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_set_chunk(bobe_gp_t* g, int64_t chunk) {
  API_BEGIN
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
  if (chunk == 0) chunk = 8192;
  if (chunk < TILE || chunk % TILE) throw Err(BOBE_ERR_ARG, "chunk must be a positive multiple of 128");
  g->use();
  g->sync();
  g->chunk = chunk;
  return BOBE_OK;
  API_END
}

int64_t bobe_gp_npoints(bobe_gp_t* g) { return g ? g->N : 0; }

int bobe_gp_set_data(bobe_gp_t* g, const double* X, const double* ys, int64_t N) {
  API_BEGIN
  if (!g || !X || !ys) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (N < 1) throw Err(BOBE_ERR_ARG, "N must be >= 1");
  g->use();
  g->sync();
  const int64_t Np = round_up(N, TILE);
  g->N = N;
  if (Np != g->Np) {
    g->Np = Np;
    g->nb = (int)(Np / TILE);
    g->alloc_for_n();
  }
  g->X.ensure((size_t)N * g->d * sizeof(double));
  HIPCHK(hipMemcpyAsync(g->X.p, X, (size_t)N * g->d * sizeof(double),
                        is_device_ptr(X) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemsetAsync(g->y.p, 0, (size_t)Np * sizeof(double), g->stream));
  HIPCHK(hipMemcpyAsync(g->y.p, ys, (size_t)N * sizeof(double),
                        is_device_ptr(ys) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
  g->sync();
  g->have_data = true;
  g->factored = false;
  g->forget_z();
  return BOBE_OK;
  API_END
}

int bobe_gp_set_hyper(bobe_gp_t* g, const double* ls, double kvar, double noise) {
  API_BEGIN
  if (!g || !ls) throw Err(BOBE_ERR_ARG, "NULL argument");
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
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  g->factor_into(g->hyp, g->XsT.d(), g->A.d(), g->Linv.d(), g->w.d(), g->alpha.d());
  const int inf = g->read_info();
  g->factored = true;
  g->forget_z();
  g->not_pd = (inf != 0x7f7f7f7f);
  if (g->not_pd) {
    const double nan = std::nan("");
    const int64_t nn = g->Np * g->Np;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, g->stream, g->A.d(), nn, nan);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, g->stream, g->Linv.d(), nn, nan);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((g->Np + 255) / 256)), dim3(256), 0, g->stream, g->alpha.d(), g->Np, nan);
    LAUNCH_CHECK();
    g->sync();
    g_err = "kernel matrix not positive definite at column " + std::to_string(inf - 1);
    return BOBE_NOT_PD;
  }
  return BOBE_OK;
  API_END
}

int bobe_gp_mll(bobe_gp_t* g, const double* ls, double kvar, double* mll, double* grad) {
  API_BEGIN
  if (!g || !ls || !mll) throw Err(BOBE_ERR_ARG, "NULL argument");
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
  if (!g || !ls || !kvar || !mll) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (B < 0) throw Err(BOBE_ERR_ARG, "B must be >= 0");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  const int d = g->d;
  int worst = BOBE_OK;
  if (B >= 2 && g->N >= tuning().lockstep_min_n) {      // (kernel-class event timing works there too: one stream, no capture)
    // GPU-bound sizes: the evaluations advance in lock step through one batched launch sequence
    for (int64_t b0 = 0; b0 < B; b0 += BOBE_MAX_MLL_SLOTS) {
      const int nbat = (int)std::min<int64_t>(BOBE_MAX_MLL_SLOTS, B - b0);
      Hyper hs[BOBE_MAX_MLL_SLOTS];
      for (int i = 0; i < nbat; ++i) {
        hs[i] = g->hyp;
        for (int j = 0; j < d; ++j) hs[i].ls[j] = ls[(b0 + i) * d + j];
        hs[i].kvar = kvar[b0 + i];
      }
      g->mll_lockstep_enqueue(nbat, hs, grad != nullptr);
      const int st = g->mll_lockstep_collect(nbat, mll + b0, grad ? grad + b0 * (d + 1) : nullptr,
                                             status ? status + b0 : nullptr);
      if (st != BOBE_OK) worst = st;
    }
    return worst;
  }
  // (this path serves batches below lockstep_min_n points: kernels of a few workgroups each, where eight evaluations in
  //  flight beat four - an 8-restart fit at N = 400 / 900: 59.5 / 93.7 against 66.5 / 102.3 ms; BOBE_MLL_SLOTS overrides)
  static const bool slots_from_env = std::getenv("BOBE_MLL_SLOTS") != nullptr;
  const int width = std::max(1, std::min<int>(slots_from_env ? tuning().mll_slots : BOBE_MAX_MLL_SLOTS, BOBE_MAX_MLL_SLOTS));
  for (int64_t b0 = 0; b0 < B; b0 += width) {
    const int nbat = (int)std::min<int64_t>(width, B - b0);
    static const bool trace = std::getenv("BOBE_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    if (nbat == 1) {   // a lone evaluation owns the whole chip on the handle's stream
      Hyper h = g->hyp;
      for (int j = 0; j < d; ++j) h.ls[j] = ls[b0 * d + j];
      h.kvar = kvar[b0];
      g->mll_enqueue(h, grad != nullptr);
    } else {
      g->ensure_slots(nbat);
      for (int i = 0; i < nbat; ++i)
        if (g->slots[i]->busy)
          throw Err(BOBE_ERR_STATE, "an evaluation submitted with bobe_gp_mll_submit is still in flight on a slot this batch needs");
      const std::vector<hipStream_t>& sts = g->slot_stream_set();
      // the batch streams start after everything already queued on the handle's stream (data uploads)
      HIPCHK(hipEventRecord(g->ev_batch, g->stream));
      for (int i = 0; i < nbat; ++i) {
        Hyper h = g->hyp;
        for (int j = 0; j < d; ++j) h.ls[j] = ls[(b0 + i) * d + j];
        h.kvar = kvar[b0 + i];
        bobe_gp::Slot& sl = *g->slots[i];
        sl.stream = sts[i];
        HIPCHK(hipStreamWaitEvent(sl.stream, g->ev_batch, 0));
        g->swap_slot(sl);
        try {
          g->mll_enqueue(h, grad != nullptr);
        } catch (...) {
              g->swap_slot(sl);
          throw;
        }
          g->swap_slot(sl);
      }
    }
    const auto t_enq = std::chrono::steady_clock::now();
    for (int i = 0; i < nbat; ++i) {
      double* gi = grad ? grad + (b0 + i) * (d + 1) : nullptr;
      int st;
      if (nbat == 1) {
        st = g->mll_collect(mll + b0, gi);
      } else {
        bobe_gp::Slot& sl = *g->slots[i];
        g->swap_slot(sl);
        try {
          st = g->mll_collect(mll + b0 + i, gi);
        } catch (...) {
          g->swap_slot(sl);
          throw;
        }
        g->swap_slot(sl);
      }
      if (status) status[b0 + i] = st;
      if (st != BOBE_OK) worst = st;
    }
    if (trace) {
      const auto t_end = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[bobe] mll_batch B=%d: enqueue %.3f ms, total %.3f ms\n", nbat,
                   std::chrono::duration<double, std::milli>(t_enq - t_start).count(),
                   std::chrono::duration<double, std::milli>(t_end - t_start).count());
    }
  }
  return worst;
  API_END
}

int bobe_gp_mll_submit(bobe_gp_t* g, int slot, const double* ls, double kvar, int want_grad) {
  API_BEGIN
  if (!g || !ls) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (slot < 0 || slot >= BOBE_MAX_MLL_SLOTS) throw Err(BOBE_ERR_ARG, "slot out of range");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  std::lock_guard<std::mutex> lock(g->submit_mutex);
  g->use();
  g->ensure_slots(slot + 1);
  const std::vector<hipStream_t>& sts = g->slot_stream_set();
  Hyper h = g->hyp;
  for (int j = 0; j < g->d; ++j) h.ls[j] = ls[j];
  h.kvar = kvar;
  bobe_gp::Slot& sl = *g->slots[slot];
  if (sl.busy)      // its pinned inputs / workspace / results are still in use by the evaluation not yet collected
    throw Err(BOBE_ERR_STATE, "slot already has an evaluation in flight: call bobe_gp_mll_wait first");
  sl.stream = sts[slot];
  // ordered after whatever is queued on the handle's stream (data uploads); the event is private to the slot
  if (!sl.ev) HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(sl.ev, g->stream));
  HIPCHK(hipStreamWaitEvent(sl.stream, sl.ev, 0));
  g->swap_slot(sl);
  try {
    g->mll_enqueue(h, want_grad != 0);
  } catch (...) {
    g->swap_slot(sl);
    throw;
  }
  g->swap_slot(sl);
  sl.busy = true;
  sl.want_grad = want_grad != 0;
  return BOBE_OK;
  API_END
}

int bobe_gp_mll_wait(bobe_gp_t* g, int slot, double* mll, double* grad) {
  API_BEGIN
  if (!g || !mll) throw Err(BOBE_ERR_ARG, "NULL argument");
  bobe_gp::Slot* slp = nullptr;
  {
    std::lock_guard<std::mutex> lock(g->submit_mutex);    // (the slot table grows under this mutex)
    if (slot < 0 || slot >= (int)g->slots.size() || !g->slots[slot]->busy)
      throw Err(BOBE_ERR_STATE, "no evaluation was submitted to this slot");
    slp = g->slots[slot];
  }
  HIPCHK(hipSetDevice(g->device));
  bobe_gp::Slot& sl = *slp;
  struct Release {                                        // the slot is free again once its stream has drained
    bobe_gp::Slot& s;
    ~Release() { s.busy = false; }
  } release{sl};
  return g->slot_collect(sl, mll, sl.want_grad ? grad : nullptr);
  API_END
}

int bobe_gp_predict(bobe_gp_t* g, const double* Xq, int64_t C, double* mean, double* var, int nan_policy) {
  API_BEGIN
  if (!g || !Xq) throw Err(BOBE_ERR_ARG, "NULL argument");
  g->use();
  g->sweep(Xq, C, nullptr, 0, 1.0, nullptr, nullptr, mean, var, nan_policy ? 1 : 0, nullptr, nullptr, nullptr, nullptr,
           nullptr);
  return BOBE_OK;
  API_END
}

int bobe_gp_wip_sweep(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                      double* wipv, double* wipstd, double* mean, double* var, int64_t* argmin_v, double* min_v,
                      int64_t* argmin_s, double* min_s) {
  API_BEGIN
  if (!g || !cand || !Z) throw Err(BOBE_ERR_ARG, "NULL argument");
  g->use();
  g->sweep(cand, C, Z, M, y_std, wipv, wipstd, mean, var, 1, argmin_v, min_v, argmin_s, min_s, nullptr);
  return BOBE_OK;
  API_END
}

int bobe_gp_fantasy_var(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                        double* out) {
  API_BEGIN
  if (!g || !cand || !Z || !out) throw Err(BOBE_ERR_ARG, "NULL argument");
  g->use();
  g->sweep(cand, C, Z, M, y_std, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, out);
  return BOBE_OK;
  API_END
}

int bobe_gp_wip_grad(bobe_gp_t* g, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                     double* wipv, double* wipstd, double* dwipv, double* dwipstd) {
  API_BEGIN
  if (!g || !cand || !Z) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (C <= 0 || M <= 0) throw Err(BOBE_ERR_ARG, "C and M must be positive");
  g->use();
  const int d = g->d, nb = g->nb;
  const int64_t Np = g->Np, Mp = round_up(M, TILE), CH = std::min<int64_t>(g->chunk, 1024);
  const double kself = g->hyp.kvar + g->hyp.noise;
  const double* cin = g->fetch(cand, (size_t)C * d, g->in_stage);
  g->prepare_z(Z, M, Mp);                                   // ZsT, W_Z = K^-1 K(X,Z), base_z
  g->CsT.ensure((size_t)d * std::max<int64_t>(CH, g->chunk) * sizeof(double));
  g->kXC.ensure((size_t)Np * std::max<int64_t>(CH, g->chunk) * sizeof(double));
  g->pv.ensure((size_t)Np * CH * sizeof(double));           // V = Linv k_c
  g->ps.ensure((size_t)Np * CH * sizeof(double));           // U = K^-1 k_c
  g->qpart.ensure((size_t)nb * std::max<int64_t>(std::max<int64_t>(CH, g->chunk), Mp) * sizeof(double));
  g->sc.ensure((size_t)std::max<int64_t>(CH, g->chunk) * sizeof(double));
  double* d_v = g->out_dev(wipv, C, g->o_wipv);
  double* d_s = g->out_dev(wipstd, C, g->o_wipstd);
  double* d_dv = g->out_dev(dwipv, (size_t)C * d, g->o_mean);
  double* d_ds = g->out_dev(dwipstd, (size_t)C * d, g->o_var);
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  static const bool few_path = [] { const char* e = std::getenv("BOBE_WIPG_FEW"); return !e || std::atoi(e) != 0; }();
  if (few_path && C <= 16 && g->N <= 4096) {
    // A handful of candidates (the L-BFGS refinement sends one): matrix-vector stages spread over the chip instead of
    // 128-column tile passes and one workgroup per candidate (kernels.hpp, "the same for a HANDFUL of candidates").
    const int64_t N = g->N;
    const int nzw = (int)(Mp / 64), nnw = (int)((N + 63) / 64);
    const size_t n_vec = (size_t)C * Np, n_a = (size_t)C * Mp;
    g->wg_ws.ensure((3 * n_vec + 2 * n_a + (size_t)C * nzw * WG_ZS + (size_t)C * nnw * WG_NS) * sizeof(double));
    g->part.ensure((size_t)C * nb * Np * sizeof(double));
    double* kc = g->wg_ws.d();
    double* vv = kc + n_vec;
    double* uu = vv + n_vec;
    double* a1 = uu + n_vec;
    double* b1 = a1 + n_a;
    double* pz = b1 + n_a;
    double* pn = pz + (size_t)C * nzw * WG_ZS;
    const Hyper& h = g->hyp;
    const double* li = g->Linv.d();
#define FEW(KE, DC)                                                                                                      \
  do {                                                                                                                   \
    hipLaunchKernelGGL((k_wg_col<KE, DC>), dim3((unsigned)(Np / 256 + 1), (unsigned)C), dim3(256), 0, g->stream,           \
                       (const double*)g->XsT.d(), Np, N, Np, cin, h, kc);                                                 \
    hipLaunchKernelGGL(k_gemv_lower, dim3((unsigned)(Np / 4), (unsigned)C), dim3(256), 0, g->stream, li, Np, Np,          \
                       (const double*)kc, vv, (int64_t)0, Np, Np);                                                        \
    hipLaunchKernelGGL(k_gemv_t_part, dim3((unsigned)(Np / 64), (unsigned)nb, (unsigned)C), dim3(256), 0, g->stream, li,  \
                       Np, 1, (const double*)vv, g->part.d(), Np, (int64_t)0, Np, (int64_t)nb * Np);                      \
    hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((Np + 255) / 256), (unsigned)C), dim3(256), 0, g->stream,          \
                       (const double*)g->part.d(), Np, nb, 1, Np, uu, (int64_t)nb * Np, Np);                              \
    hipLaunchKernelGGL((k_wg_cross<KE, DC>), dim3((unsigned)nzw, (unsigned)C), dim3(256), (size_t)N * sizeof(double),     \
                       g->stream, (const double*)g->ZsT.d(), Mp, M, (const double*)g->WZ.d(), Mp, N, Np,                  \
                       (const double*)kc, (const double*)vv, cin, h, kself, (const double*)g->basez.d(), y_std * y_std,   \
                       a1, b1, Mp, pz);                                                                                  \
    hipLaunchKernelGGL((k_wg_rows<KE, DC>), dim3((unsigned)nnw, (unsigned)C), dim3(256), 0, g->stream,                    \
                       (const double*)g->XsT.d(), Np, N, Np, cin, h, (const double*)g->WZ.d(), Mp, Mp,                    \
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
    if (packed) {                                  // one copy through the pinned result block instead of four
      g->o_wipv.ensure(n_out * sizeof(double));
      double* ob = g->o_wipv.d();
      hipLaunchKernelGGL(k_wg_final, dim3((unsigned)C), dim3(64), 0, g->stream, (const double*)pz, nzw, (const double*)pn,
                         nnw, h, M, ob, ob + C, ob + 2 * C, ob + 2 * C + C * d);
      LAUNCH_CHECK();
      HIPCHK(hipMemcpyAsync(g->h_res, ob, n_out * sizeof(double), hipMemcpyDeviceToHost, g->stream));
      g->sync();
      const double* hr = g->h_res;
      if (wipv) std::memcpy(wipv, hr, (size_t)C * sizeof(double));
      if (wipstd) std::memcpy(wipstd, hr + C, (size_t)C * sizeof(double));
      if (dwipv) std::memcpy(dwipv, hr + 2 * C, (size_t)C * d * sizeof(double));
      if (dwipstd) std::memcpy(dwipstd, hr + 2 * C + C * d, (size_t)C * d * sizeof(double));
      return BOBE_OK;
    }
    hipLaunchKernelGGL(k_wg_final, dim3((unsigned)C), dim3(64), 0, g->stream, (const double*)pz, nzw, (const double*)pn,
                       nnw, h, M, d_v, d_s, d_dv, d_ds);
    LAUNCH_CHECK();
    g->out_finish(wipv, C, g->o_wipv);
    g->out_finish(wipstd, C, g->o_wipstd);
    g->out_finish(dwipv, (size_t)C * d, g->o_mean);
    g->out_finish(dwipstd, (size_t)C * d, g->o_var);
    g->sync();
    return BOBE_OK;
  }
  for (int64_t c0 = 0; c0 < C; c0 += CH) {
    const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
    g->scale(cin + c0 * d, nc, ncp, g->hyp, g->CsT.d(), CH);
    g->kernel_matrix_cross(g->XsT.d(), Np, g->N, Np, g->CsT.d(), CH, nc, ncp, g->hyp, g->kXC.d(), CH);
    hipLaunchKernelGGL(k_trimul, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream,
                       (const double*)g->Linv.d(), Np, nb, (const double*)g->kXC.d(), CH, g->pv.d(), CH, g->qpart.d(), CH,
                       (const double*)nullptr, (int64_t)0, 0, (double*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, g->stream,
                       (const double*)g->qpart.d(), CH, nb, nc, kself, 1, g->sc.d(), (double*)nullptr);
    hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream,
                       (const double*)g->Linv.d(), Np, nb, (const double*)g->pv.d(), CH, g->ps.d(), CH);
#define WG(KE, DC)                                                                                                   \
  hipLaunchKernelGGL((k_wip_grad<KE, DC>), dim3((unsigned)nc), dim3(256), 0, g->stream, (const double*)g->XsT.d(), Np,  \
                     g->N, (const double*)g->CsT.d(), CH, (const double*)g->ZsT.d(), Mp, M, (const double*)g->WZ.d(),  \
                     Mp, (const double*)g->ps.d(), CH, (const double*)g->sc.d(), (const double*)g->basez.d(), g->hyp,  \
                     y_std * y_std, d_v ? d_v + c0 : nullptr, d_s ? d_s + c0 : nullptr, d_dv ? d_dv + c0 * d : nullptr, \
                     d_ds ? d_ds + c0 * d : nullptr)
    if (g->hyp.kern == 0) {
      if (dcap == 8) WG(0, 8); else if (dcap == 16) WG(0, 16); else WG(0, 32);
    } else {
      if (dcap == 8) WG(1, 8); else if (dcap == 16) WG(1, 16); else WG(1, 32);
    }
#undef WG
    LAUNCH_CHECK();
  }
  g->out_finish(wipv, C, g->o_wipv);
  g->out_finish(wipstd, C, g->o_wipstd);
  g->out_finish(dwipv, (size_t)C * d, g->o_mean);
  g->out_finish(dwipstd, (size_t)C * d, g->o_var);
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_acq_ei(bobe_gp_t* g, const double* Xq, int64_t C, double best_y, double zeta, int mode, double* out) {
  API_BEGIN
  if (!g || !Xq || !out) throw Err(BOBE_ERR_ARG, "NULL argument");
  g->use();
  g->o_mean.ensure(C * sizeof(double));
  g->o_var.ensure(C * sizeof(double));
  g->sweep(Xq, C, nullptr, 0, 1.0, nullptr, nullptr, g->o_mean.d(), g->o_var.d(), 1, nullptr, nullptr, nullptr, nullptr,
           nullptr);
  double* d_out = g->out_dev(out, C, g->o_wipv);
  hipLaunchKernelGGL(k_ei, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, g->stream, (const double*)g->o_mean.d(),
                     (const double*)g->o_var.d(), C, best_y, zeta, mode, d_out);
  LAUNCH_CHECK();
  g->out_finish(out, C, g->o_wipv);
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_predict_grad(bobe_gp_t* g, const double* Xq, int64_t C, double* mean, double* var, double* dmean,
                         double* dvar) {
  API_BEGIN
  if (!g || !Xq || !dmean) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!dvar && var) throw Err(BOBE_ERR_ARG, "var without dvar: use bobe_gp_predict");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (C <= 0) throw Err(BOBE_ERR_ARG, "C must be positive");
  g->use();
  const int d = g->d, nb = g->nb;
  const int64_t Np = g->Np, CH = std::min<int64_t>(g->chunk, 2048);
  const double kself = g->hyp.kvar + g->hyp.noise;
  const double* cin = g->fetch(Xq, (size_t)C * d, g->in_stage);
  if (!dvar) {
    // mean-only mode (HMC on the surrogate): scale the queries, then one kernel that walks the training points
    // and accumulates the mean and its gradient - no K(X, C), no triangular products
    g->CsT.ensure((size_t)d * std::max<int64_t>(CH, g->chunk) * sizeof(double));
    double* d_mean = g->out_dev(mean, C, g->o_mean);
    double* d_dm = g->out_dev(dmean, (size_t)C * d, g->o_wipv);
    const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
    for (int64_t c0 = 0; c0 < C; c0 += CH) {
      const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
      g->scale(cin + c0 * d, nc, ncp, g->hyp, g->CsT.d(), CH);
      const dim3 grid((unsigned)((nc + 63) / 64));
      const size_t sm = (size_t)(d + 1) * 128 * sizeof(double);
#define PGM(KE, DC)                                                                                                  \
  hipLaunchKernelGGL((k_predict_grad<KE, DC>), grid, dim3(256), sm, g->stream, (const double*)g->XsT.d(), Np, g->N,    \
                     (const double*)g->CsT.d(), CH, nc, (const double*)g->alpha.d(), (const double*)nullptr,           \
                     (int64_t)0, (const double*)nullptr, g->hyp, d_dm + c0 * d, (double*)nullptr,                     \
                     d_mean ? d_mean + c0 : nullptr)
      if (g->hyp.kern == 0) {
        if (dcap == 8) PGM(0, 8); else if (dcap == 16) PGM(0, 16); else PGM(0, 32);
      } else {
        if (dcap == 8) PGM(1, 8); else if (dcap == 16) PGM(1, 16); else PGM(1, 32);
      }
#undef PGM
      LAUNCH_CHECK();
    }
    g->out_finish(mean, C, g->o_mean);
    g->out_finish(dmean, (size_t)C * d, g->o_wipv);
    g->sync();
    return BOBE_OK;
  }
  g->CsT.ensure((size_t)d * std::max<int64_t>(CH, g->chunk) * sizeof(double));
  g->kXC.ensure((size_t)Np * std::max<int64_t>(CH, g->chunk) * sizeof(double));
  g->forget_z();                                      // (VZ / WZ double as this call's scratch)
  g->VZ.ensure((size_t)Np * CH * sizeof(double));     // V = Linv k
  g->WZ.ensure((size_t)Np * CH * sizeof(double));     // U = Linv^T V = K^-1 k
  g->qpart.ensure((size_t)nb * std::max<int64_t>(CH, g->chunk) * sizeof(double));
  g->part.ensure((size_t)nb * std::max<int64_t>(Np, std::max<int64_t>(CH, g->chunk)) * sizeof(double));
  g->sc.ensure((size_t)std::max<int64_t>(CH, g->chunk) * sizeof(double));
  double* d_mean = g->out_dev(mean, C, g->o_mean);
  double* d_var = g->out_dev(var, C, g->o_var);
  double* d_dm = g->out_dev(dmean, (size_t)C * d, g->o_wipv);
  double* d_dv = g->out_dev(dvar, (size_t)C * d, g->o_wipstd);
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
  for (int64_t c0 = 0; c0 < C; c0 += CH) {
    const int64_t nc = std::min<int64_t>(CH, C - c0), ncp = round_up(nc, TILE);
    g->scale(cin + c0 * d, nc, ncp, g->hyp, g->CsT.d(), CH);
    g->kernel_matrix_cross(g->XsT.d(), Np, g->N, Np, g->CsT.d(), CH, nc, ncp, g->hyp, g->kXC.d(), CH,
                           d_mean ? (const double*)g->alpha.d() : nullptr, d_mean ? g->part.d() : nullptr, CH);
    if (d_mean) {
      hipLaunchKernelGGL(k_colsum_parts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, g->stream,
                         (const double*)g->part.d(), CH, nb, 0, nc, d_mean + c0);
    }
    hipLaunchKernelGGL(k_trimul, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream,
                       (const double*)g->Linv.d(), Np, nb, (const double*)g->kXC.d(), CH, g->VZ.d(), CH, g->qpart.d(), CH,
                       (const double*)nullptr, (int64_t)0, 0, (double*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_predict_finalize, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, g->stream,
                       (const double*)g->qpart.d(), CH, nb, nc, kself, 1, g->sc.d(), d_var ? d_var + c0 : nullptr);
    hipLaunchKernelGGL(k_trimul_t, dim3((unsigned)(ncp / TILE), (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream,
                       (const double*)g->Linv.d(), Np, nb, (const double*)g->VZ.d(), CH, g->WZ.d(), CH);
    const dim3 grid((unsigned)((nc + 63) / 64));
    const size_t sm = (size_t)(d + 1) * 128 * sizeof(double);
#define PG(KE, DC)                                                                                                  \
  hipLaunchKernelGGL((k_predict_grad<KE, DC>), grid, dim3(256), sm, g->stream, (const double*)g->XsT.d(), Np, g->N,    \
                     (const double*)g->CsT.d(), CH, nc, (const double*)g->alpha.d(), (const double*)g->WZ.d(), CH,     \
                     (const double*)g->sc.d(), g->hyp, d_dm + c0 * d, d_dv + c0 * d)
    if (g->hyp.kern == 0) {
      if (dcap == 8) PG(0, 8); else if (dcap == 16) PG(0, 16); else PG(0, 32);
    } else {
      if (dcap == 8) PG(1, 8); else if (dcap == 16) PG(1, 16); else PG(1, 32);
    }
#undef PG
    LAUNCH_CHECK();
  }
  g->out_finish(mean, C, g->o_mean);
  g->out_finish(var, C, g->o_var);
  g->out_finish(dmean, (size_t)C * d, g->o_wipv);
  g->out_finish(dvar, (size_t)C * d, g->o_wipstd);
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_hmc_leapfrog(bobe_gp_t* g, int64_t P, double* U, double* Pm, const double* inv_mass, double eps, int L,
                         double y_std, double y_mean, double temp, double* logp, double* grad, double* mean, double* X) {
  API_BEGIN
  if (!g || !U || !Pm || !inv_mass || !logp || !grad || !mean || !X) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (P <= 0 || L < 1 || !(temp > 0.0)) throw Err(BOBE_ERR_ARG, "bad argument");
  g->use();
  const int d = g->d;
  const size_t pd = (size_t)P * d;
  // staging: [U | Pm | grad | X] (P*d each), [logp | mean] (P each), inv_mass (d)
  g->in_stage.ensure((4 * pd + 2 * (size_t)P + d) * sizeof(double));
  double* dU = g->in_stage.d();
  double* dP = dU + pd;
  double* dG = dP + pd;
  double* dX = dG + pd;
  double* dL = dX + pd;
  double* dM = dL + P;
  double* dI = dM + P;
  const bool dev = is_device_ptr(U);
  const hipMemcpyKind in = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const hipMemcpyKind out = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIPCHK(hipMemcpyAsync(dU, U, pd * sizeof(double), in, g->stream));
  HIPCHK(hipMemcpyAsync(dP, Pm, pd * sizeof(double), in, g->stream));
  HIPCHK(hipMemcpyAsync(dI, inv_mass, (size_t)d * sizeof(double), is_device_ptr(inv_mass) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        g->stream));
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
#define HL(KE, DC)                                                                                                   \
  hipLaunchKernelGGL((k_hmc_leapfrog<KE, DC>), dim3((unsigned)P), dim3(256), 0, g->stream, (const double*)g->XsT.d(),   \
                     g->Np, g->N, (const double*)g->alpha.d(), g->hyp, dU, dP, (const double*)dI, eps, L, y_std, y_mean, \
                     temp, dL, dG, dM, dX)
  if (g->hyp.kern == 0) {
    if (dcap == 8) HL(0, 8); else if (dcap == 16) HL(0, 16); else HL(0, 32);
  } else {
    if (dcap == 8) HL(1, 8); else if (dcap == 16) HL(1, 16); else HL(1, 32);
  }
#undef HL
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(U, dU, pd * sizeof(double), out, g->stream));
  HIPCHK(hipMemcpyAsync(Pm, dP, pd * sizeof(double), out, g->stream));
  HIPCHK(hipMemcpyAsync(grad, dG, pd * sizeof(double), out, g->stream));
  HIPCHK(hipMemcpyAsync(X, dX, pd * sizeof(double), out, g->stream));
  HIPCHK(hipMemcpyAsync(logp, dL, (size_t)P * sizeof(double), out, g->stream));
  HIPCHK(hipMemcpyAsync(mean, dM, (size_t)P * sizeof(double), out, g->stream));
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_hmc_run(bobe_gp_t* g, int64_t P, double* state, double* adapt, const double* inv_mass, uint64_t seed,
                    int64_t it0, int niter, int do_adapt, double y_std, double y_mean, double temp, int hist_from,
                    double* hist, int thin, double* keep, double* dbg) {
  API_BEGIN
  if (!g || !state || !adapt || !inv_mass) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  if (P <= 0 || niter < 1 || it0 < 0 || !(temp > 0.0) || thin < 1 || hist_from < 0 || hist_from > niter)
    throw Err(BOBE_ERR_ARG, "bad argument");
  g->use();
  const int d = g->d;
  const size_t sw = 3 * (size_t)d + 2, ns = (size_t)P * sw, na = (size_t)P * 5;
  const size_t nh = hist ? (size_t)(niter - hist_from) * P * d : 0;
  const size_t nk = keep ? (size_t)(niter / thin) * P * (d + 1) : 0;
  const size_t nd = dbg ? (size_t)P * (d + 3) : 0;
  // staging: state | adapt | inv_mass | hist | keep | dbg
  g->in_stage.ensure((ns + na + d + nh + nk + nd) * sizeof(double));
  double* dS = g->in_stage.d();
  double* dA = dS + ns;
  double* dI = dA + na;
  double* dH = dI + d;
  double* dK = dH + nh;
  double* dD = dK + nk;
  HIPCHK(hipMemcpyAsync(dS, state, ns * sizeof(double), hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(dA, adapt, na * sizeof(double), hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(dI, inv_mass, (size_t)d * sizeof(double), hipMemcpyHostToDevice, g->stream));
  const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
#define HR(KE, DC)                                                                                                  \
  hipLaunchKernelGGL((k_hmc_run<KE, DC>), dim3((unsigned)P), dim3(256), 0, g->stream, (const double*)g->XsT.d(), g->Np, \
                     g->N, (const double*)g->alpha.d(), g->hyp, P, dS, dA, (const double*)dI,                       \
                     (unsigned long long)seed, it0, niter, do_adapt, y_std, y_mean, temp, hist_from,                \
                     hist ? dH : nullptr, thin, keep ? dK : nullptr, dbg ? dD : nullptr)
  if (g->hyp.kern == 0) {
    if (dcap == 8) HR(0, 8); else if (dcap == 16) HR(0, 16); else HR(0, 32);
  } else {
    if (dcap == 8) HR(1, 8); else if (dcap == 16) HR(1, 16); else HR(1, 32);
  }
#undef HR
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(state, dS, ns * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipMemcpyAsync(adapt, dA, na * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  if (hist) HIPCHK(hipMemcpyAsync(hist, dH, nh * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  if (keep) HIPCHK(hipMemcpyAsync(keep, dK, nk * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  if (dbg) HIPCHK(hipMemcpyAsync(dbg, dD, nd * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_kernel(bobe_gp_t* g, const double* A, int64_t nA, const double* B, int64_t nB, const double* ls,
                   double kvar, double noise, int include_noise, double* out) {
  API_BEGIN
  if (!g || !A || !B || !out) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (nA < 1 || nB < 1) throw Err(BOBE_ERR_ARG, "empty input");
  if (include_noise && nA != nB) throw Err(BOBE_ERR_ARG, "include_noise needs a square kernel matrix (gp.py:153)");
  g->use();
  const int d = g->d;
  Hyper hk = g->hyp;
  if (ls) {
    for (int j = 0; j < d; ++j) hk.ls[j] = ls[j];
    hk.kvar = kvar;
    hk.noise = noise;
  }
  const int64_t pa = round_up(nA, TILE), pb = round_up(nB, TILE);
  const double* a_in = g->fetch(A, (size_t)nA * d, g->in_stage);
  const double* b_in = g->fetch(B, (size_t)nB * d, g->z_stage);
  g->kin_a.ensure((size_t)d * pa * sizeof(double));
  g->kin_b.ensure((size_t)d * pb * sizeof(double));
  g->kout.ensure((size_t)pa * pb * sizeof(double));
  g->scale(a_in, nA, pa, hk, g->kin_a.d(), pa);
  g->scale(b_in, nB, pb, hk, g->kin_b.d(), pb);
  g->kernel_matrix_cross(g->kin_a.d(), pa, nA, pa, g->kin_b.d(), pb, nB, pb, hk, g->kout.d(), pb);
  const bool dev = is_device_ptr(out);
  HIPCHK(hipMemcpy2DAsync(out, (size_t)nB * sizeof(double), g->kout.p, (size_t)pb * sizeof(double),
                          (size_t)nB * sizeof(double), (size_t)nA, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          g->stream));
  g->sync();
  if (include_noise) {
    // noise * eye(n) (gp.py:153): added on the host side of the copy for host outputs, by a tiny kernel otherwise
    if (!dev) {
      for (int64_t i = 0; i < nA; ++i) out[i * nB + i] += hk.noise;
    } else {
      std::vector<double> diag((size_t)nA);
      HIPCHK(hipMemcpy2D(diag.data(), sizeof(double), out, (size_t)(nB + 1) * sizeof(double), sizeof(double), (size_t)nA,
                         hipMemcpyDeviceToHost));
      for (auto& v : diag) v += hk.noise;
      HIPCHK(hipMemcpy2D(out, (size_t)(nB + 1) * sizeof(double), diag.data(), sizeof(double), sizeof(double), (size_t)nA,
                         hipMemcpyHostToDevice));
    }
  }
  return BOBE_OK;
  API_END
}

static void copy_out_matrix(bobe_gp* g, const double* src, double* dst, int lower_only) {
  const int64_t N = g->N;
  double* d_out = g->out_dev(dst, (size_t)N * N, g->kout);
  hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, g->stream, src, g->Np, d_out,
                     N, N, N, lower_only);
  LAUNCH_CHECK();
  g->out_finish(dst, (size_t)N * N, g->kout);
}

int bobe_gp_get_chol(bobe_gp_t* g, double* L, double* alpha) {
  API_BEGIN
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  g->use();
  if (L) copy_out_matrix(g, g->A.d(), L, g->not_pd ? 0 : 1);   // not PD: all-NaN, like jnp.linalg.cholesky
  if (alpha)
    HIPCHK(hipMemcpyAsync(alpha, g->alpha.p, (size_t)g->N * sizeof(double),
                          is_device_ptr(alpha) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g->stream));
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_set_chol(bobe_gp_t* g, const double* L, const double* alpha) {
  API_BEGIN
  if (!g || !L || !alpha) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  const int64_t N = g->N, Np = g->Np;
  const double* l_in = g->fetch(L, (size_t)N * N, g->kout);
  hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, g->stream, l_in,
                     N, g->A.d(), Np, Np);
  HIPCHK(hipMemsetAsync(g->alpha.p, 0, (size_t)Np * sizeof(double), g->stream));
  HIPCHK(hipMemcpyAsync(g->alpha.p, alpha, (size_t)N * sizeof(double),
                        is_device_ptr(alpha) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
  g->scale(g->X.d(), N, Np, g->hyp, g->XsT.d(), Np);
  for (int k = 0; k < g->nb; ++k)
    hipLaunchKernelGGL((k_potf2<false, false>), dim3(1), dim3(256), POTF2_SMEM_BYTES, g->stream, g->A.d(), Np, g->Linv.d(), Np,
                       k, static_cast<int*>(g->info.p), (unsigned long long*)nullptr);
  LAUNCH_CHECK();
  g->trtri(g->A.d(), g->Linv.d(), g->Tmp.d());
  g->sync();
  g->factored = true;
  g->forget_z();
  g->not_pd = false;
  return BOBE_OK;
  API_END
}

int bobe_gp_clone_state(bobe_gp_t* dst, bobe_gp_t* src) {
  API_BEGIN
  if (!dst || !src) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (dst == src) return BOBE_OK;
  if (dst->d != src->d || dst->kern != src->kern || dst->device != src->device)
    throw Err(BOBE_ERR_ARG, "clone needs handles of the same kernel, dimension and device");
  if (!src->have_data) throw Err(BOBE_ERR_STATE, "source holds no data");
  src->use();
  src->sync();
  dst->sync();
  dst->N = src->N;
  dst->hyp = src->hyp;
  if (dst->Np != src->Np) {
    dst->Np = src->Np;
    dst->nb = src->nb;
    dst->alloc_for_n();
  }
  const size_t mat = (size_t)src->Np * src->Np * sizeof(double), vec = (size_t)src->Np * sizeof(double);
  dst->X.ensure((size_t)src->N * src->d * sizeof(double));
  HIPCHK(hipMemcpyAsync(dst->X.p, src->X.p, (size_t)src->N * src->d * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
  HIPCHK(hipMemcpyAsync(dst->y.p, src->y.p, vec, hipMemcpyDeviceToDevice, dst->stream));
  if (src->factored) {
    HIPCHK(hipMemcpyAsync(dst->XsT.p, src->XsT.p, (size_t)src->d * vec, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->A.p, src->A.p, mat, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->Linv.p, src->Linv.p, mat, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->alpha.p, src->alpha.p, vec, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->w.p, src->w.p, vec, hipMemcpyDeviceToDevice, dst->stream));
  }
  dst->sync();
  dst->have_data = true;
  dst->factored = src->factored;
  dst->forget_z();
  dst->not_pd = src->not_pd;
  return BOBE_OK;
  API_END
}

int bobe_gp_append(bobe_gp_t* g, const double* X_new, int64_t b, const double* y_all) {
  API_BEGIN
  if (!g || !X_new || !y_all) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (b < 1 || b > 64) throw Err(BOBE_ERR_ARG, "b must be in [1, 64] (larger batches: bobe_gp_set_data + bobe_gp_factor)");
  if (!g->factored || g->not_pd) throw Err(BOBE_ERR_STATE, "append needs a positive-definite factorised state");
  g->use();
  g->sync();
  const int d = g->d;
  const int64_t N0 = g->N, N1 = N0 + b, Np0 = g->Np, Np1 = round_up(N1, TILE);
  // The handle is rebuilt in stages (X, the padded frame, y, then the new rows).  Until the last stage is through it
  // counts as holding nothing: an error on the way (out of memory in the new frame, a failed launch) leaves a handle
  // that every later call refuses ("call bobe_gp_set_data first") instead of one with N0 points' factor under N1
  // points' data; bobe_gp_set_data + bobe_gp_factor then rebuild it from scratch (what GP.update falls back to).
  g->factored = false;
  g->have_data = false;
  g->forget_z();
  DBuf nx, oa, ol;
  try {
  // ---- training data: X gains b rows, every y changes (the caller re-standardised them, gp.py:520-536)
  {
    nx.ensure((size_t)N1 * d * sizeof(double));
    HIPCHK(hipMemcpyAsync(nx.p, g->X.p, (size_t)N0 * d * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(static_cast<double*>(nx.p) + N0 * d, X_new, (size_t)b * d * sizeof(double),
                          is_device_ptr(X_new) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
    g->sync();
    std::swap(g->X, nx);
    nx.release();
  }
  // ---- a larger padded size: move L and Linv into the new [[., 0], [0, I]] frame
  if (Np1 != Np0) {
    std::swap(oa, g->A);
    std::swap(ol, g->Linv);
    g->Np = Np1;
    g->nb = (int)(Np1 / TILE);
    g->alloc_for_n();                      // A, Linv (fresh), scratch, probs for the new block count
    hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np1 + 255) / 256), (unsigned)Np1), dim3(256), 0, g->stream,
                       (const double*)oa.p, (int64_t)0, g->A.d(), Np1, Np1);         // identity everywhere ...
    hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np1 + 255) / 256), (unsigned)Np1), dim3(256), 0, g->stream,
                       (const double*)ol.p, (int64_t)0, g->Linv.d(), Np1, Np1);
    HIPCHK(hipMemcpy2DAsync(g->A.p, (size_t)Np1 * 8, oa.p, (size_t)Np0 * 8, (size_t)Np0 * 8, (size_t)Np0,
                            hipMemcpyDeviceToDevice, g->stream));                     // ... then the old frame on top
    HIPCHK(hipMemcpy2DAsync(g->Linv.p, (size_t)Np1 * 8, ol.p, (size_t)Np0 * 8, (size_t)Np0 * 8, (size_t)Np0,
                            hipMemcpyDeviceToDevice, g->stream));
    g->sync();
    oa.release();
    ol.release();
  }
  const int64_t Np = g->Np;
  const int nb = g->nb;
  g->N = N0;                               // (old point count while the cross-covariances are assembled)
  HIPCHK(hipMemsetAsync(g->y.p, 0, (size_t)Np * sizeof(double), g->stream));
  HIPCHK(hipMemcpyAsync(g->y.p, y_all, (size_t)N1 * sizeof(double),
                        is_device_ptr(y_all) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
  // ---- V = Linv K(X_old, X_new), W = Linv^T V, S = K(X_new, X_new) + noise I - V^T V
  const int64_t bp = TILE;
  g->kXC.ensure((size_t)Np * std::max<int64_t>(bp, g->chunk) * sizeof(double));
  g->forget_z();
  g->VZ.ensure((size_t)Np * bp * sizeof(double));
  g->WZ.ensure((size_t)Np * bp * sizeof(double));
  g->kin_a.ensure((size_t)d * bp * sizeof(double));
  g->qpart.ensure((size_t)nb * std::max<int64_t>(bp, g->chunk) * sizeof(double));
  g->o_misc.ensure((size_t)(3 * 64 * 64 + 8) * sizeof(double));
  g->scale(g->X.d(), N0, Np, g->hyp, g->XsT.d(), Np);                                    // old points only (rest 0)
  g->scale(g->X.d() + N0 * d, b, bp, g->hyp, g->kin_a.d(), bp);
  g->kernel_matrix_cross(g->XsT.d(), Np, N0, Np, g->kin_a.d(), bp, b, bp, g->hyp, g->kXC.d(), bp);
  hipLaunchKernelGGL(k_trimul, dim3(1, (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream, (const double*)g->Linv.d(), Np,
                     nb, (const double*)g->kXC.d(), bp, g->VZ.d(), bp, (double*)nullptr, (int64_t)0, (const double*)nullptr,
                     (int64_t)0, 0, (double*)nullptr, (int64_t)0);
  hipLaunchKernelGGL(k_trimul_t, dim3(1, (unsigned)nb), dim3(256), GEMM_SMEM_BYTES, g->stream, (const double*)g->Linv.d(), Np,
                     nb, (const double*)g->VZ.d(), bp, g->WZ.d(), bp);
  double* G = g->o_misc.d();                                                             // b*b Gram matrix V^T V
  hipLaunchKernelGGL(k_gram_small, dim3((unsigned)b, (unsigned)b), dim3(256), 0, g->stream, (const double*)g->VZ.d(), bp, N0,
                     (int)b, G);
  LAUNCH_CHECK();
  std::vector<double> hG((size_t)b * b), hK((size_t)b * b), hx((size_t)b * d);
  HIPCHK(hipMemcpyAsync(hG.data(), G, hG.size() * 8, hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipMemcpyAsync(hx.data(), g->X.d() + N0 * d, hx.size() * 8, hipMemcpyDeviceToHost, g->stream));
  g->sync();
  // K(X_new, X_new) + noise I on the host (b <= 64 points; the kernel of gp.py:124-168 with direct differences)
  for (int64_t i = 0; i < b; ++i)
    for (int64_t j = 0; j < b; ++j) {
      double r2 = 0.0;
      for (int q = 0; q < d; ++q) {
        const double df = hx[i * d + q] / g->hyp.ls[q] - hx[j * d + q] / g->hyp.ls[q];
        r2 += df * df;
      }
      double kv;
      if (g->kern == 0) {
        kv = g->hyp.kvar * std::exp(-0.5 * r2);
      } else {
        const double dd = std::sqrt(r2 < 1e-30 ? 1e-30 : r2);
        kv = g->hyp.kvar * (1.0 + dd * (SQRT5 + (dd * 5.0) / 3.0)) * std::exp(-SQRT5 * dd);
      }
      hK[i * b + j] = kv + (i == j ? g->hyp.noise : 0.0) - hG[i * b + j];
    }
  // L22 = chol(S), L22inv by forward substitution; a non-positive pivot = the appended matrix is not positive definite
  std::vector<double> s22((size_t)2 * b * b, 0.0);
  double* L22 = s22.data();
  double* Li = s22.data() + b * b;
  bool pd = true;
  for (int64_t j = 0; j < b && pd; ++j) {
    double dj = hK[j * b + j];
    for (int64_t k = 0; k < j; ++k) dj -= L22[j * b + k] * L22[j * b + k];
    if (!(dj > 0.0)) { pd = false; break; }
    L22[j * b + j] = std::sqrt(dj);
    for (int64_t i = j + 1; i < b; ++i) {
      double v = hK[i * b + j];
      for (int64_t k = 0; k < j; ++k) v -= L22[i * b + k] * L22[j * b + k];
      L22[i * b + j] = v / L22[j * b + j];
    }
  }
  g->N = N1;
  if (!pd) {                               // same outcome as the full refactorisation: NaN state, BOBE_NOT_PD
    g->have_data = true;
    return bobe_gp_factor(g);
  }
  for (int64_t c = 0; c < b; ++c)
    for (int64_t i = c; i < b; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int64_t k = c; k < i; ++k) v -= L22[i * b + k] * Li[k * b + c];
      Li[i * b + c] = v / L22[i * b + i];
    }
  double* d22 = g->o_misc.d() + 64 * 64;
  HIPCHK(hipMemcpyAsync(d22, s22.data(), s22.size() * 8, hipMemcpyHostToDevice, g->stream));
  hipLaunchKernelGGL(k_append_rows, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, g->stream, g->A.d(), g->Linv.d(), Np,
                     N0, (int)b, (const double*)g->VZ.d(), (const double*)g->WZ.d(), bp, (const double*)d22);
  LAUNCH_CHECK();
  g->scale(g->X.d(), N1, Np, g->hyp, g->XsT.d(), Np);                                    // all points again
  g->solve_alpha(g->Linv.d(), g->w.d(), g->alpha.d(), g->part.d());                     // alpha = Linv^T Linv y
  g->sync();                               // (s22 / hG are host temporaries of this call)
  } catch (...) {
    nx.release();
    oa.release();
    ol.release();
    g->N = 0;
    g->Np = 0;                              // forces bobe_gp_set_data to size every buffer again
    g->nb = 0;
    throw;
  }
  g->have_data = true;
  g->factored = true;
  g->forget_z();
  g->not_pd = false;
  return BOBE_OK;
  API_END
}

int bobe_debug_gemm(int device, int la, int lb, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double* C, int64_t ldc) {
  API_BEGIN
  if (M % TILE || N % TILE || K % TK || M <= 0 || N <= 0 || K <= 0) throw Err(BOBE_ERR_ARG, "bad GEMM shape");
  HIPCHK(hipSetDevice(device));
  configure_kernels_once();
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
  if (!g || !Kinv) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  g->use();
  g->lauum(g->hyp, g->Linv.d(), g->alpha.d(), g->XsT.d(), g->Tmp.d(), 32);
  // symmetrise on the host side of the copy
  const int64_t N = g->N;
  std::vector<double> full((size_t)N * N);
  HIPCHK(hipMemcpy2DAsync(full.data(), (size_t)N * 8, g->Tmp.p, (size_t)g->Np * 8, (size_t)N * 8, (size_t)N,
                          hipMemcpyDeviceToHost, g->stream));
  g->sync();
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = i + 1; j < N; ++j) full[i * N + j] = full[j * N + i];
  if (is_device_ptr(Kinv)) HIPCHK(hipMemcpy(Kinv, full.data(), full.size() * 8, hipMemcpyHostToDevice));
  else std::memcpy(Kinv, full.data(), full.size() * 8);
  return BOBE_OK;
  API_END
}

int bobe_debug_linv(bobe_gp_t* g, double* Linv) {
  API_BEGIN
  if (!g || !Linv) throw Err(BOBE_ERR_ARG, "NULL argument");
  if (!g->factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  g->use();
  copy_out_matrix(g, g->Linv.d(), Linv, 1);
  g->sync();
  return BOBE_OK;
  API_END
}

int bobe_gp_profile_select(bobe_gp_t* g, int tag) {
  API_BEGIN
  if (!g) throw Err(BOBE_ERR_ARG, "gp is NULL");
  g->use();
  g->sync();
  g->prof_tag = tag;
  g->prof_used = 0;
  return BOBE_OK;
  API_END
}

int bobe_gp_profile_read(bobe_gp_t* g, double* total_ms, int64_t* launches) {
  API_BEGIN
  if (!g || !total_ms || !launches) throw Err(BOBE_ERR_ARG, "NULL argument");
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

// -------------------------------------------------------------------------------------------------
// Multi-GPU exchange step, one process per GPU: a RCCL communicator owned by the library.  librccl is
// opened on first use (dlopen), so libbobe_gp.so itself loads on hosts without it.
// -------------------------------------------------------------------------------------------------
}  // extern "C"
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
__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters) {
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

int bobe_debug_time_potrf(bobe_gp_t* g, int reps, double* ms) {
  API_BEGIN
  if (!g || !ms || reps < 1) throw Err(BOBE_ERR_ARG, "bad argument");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  double total = 0.0;
  g->scale(g->X.d(), g->N, g->Np, g->hyp, g->XsT2.d(), g->Np);
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed)
    g->assemble_kxx(g->hyp, g->XsT2.d(), g->A2.d());
    HIPCHK(hipMemsetAsync(g->info.p, 0x7f, sizeof(int), g->stream));
    HIPCHK(hipEventRecord(e0, g->stream));
    g->potrf(g->A2.d(), g->Linv2.d(), static_cast<int*>(g->info.p));
    HIPCHK(hipEventRecord(e1, g->stream));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 0) total += t;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = total / reps;
  return BOBE_OK;
  API_END
}

// B factorisations in flight at once, one evaluation slot each (the state of the fit's concurrent restarts):
// device time from the first to the last factorisation kernel, averaged over reps; *ms is for all B together
int bobe_debug_time_potrf_batch(bobe_gp_t* g, int B, int reps, double* ms) {
  API_BEGIN
  if (!g || !ms || reps < 1 || B < 1 || B > BOBE_MAX_MLL_SLOTS) throw Err(BOBE_ERR_ARG, "bad argument");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  g->ensure_slots(B);
  const std::vector<hipStream_t>& sts = g->slot_stream_set();
  hipEvent_t e0, e1;
  std::vector<hipEvent_t> done(B);
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (auto& e : done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  double total = 0.0;
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed)
    for (int i = 0; i < B; ++i) {      // K(X,X) of every slot, on the handle's stream
      bobe_gp::Slot& sl = *g->slots[i];
      g->scale(g->X.d(), g->N, g->Np, g->hyp, sl.XsT2.d(), g->Np);
      g->assemble_kxx(g->hyp, sl.XsT2.d(), sl.A2.d());
      HIPCHK(hipMemsetAsync(sl.info.p, 0x7f, sizeof(int), g->stream));
    }
    HIPCHK(hipEventRecord(e0, g->stream));
    for (int i = 0; i < B; ++i) {
      bobe_gp::Slot& sl = *g->slots[i];
      sl.stream = sts[i];
      HIPCHK(hipStreamWaitEvent(sl.stream, e0, 0));
      g->swap_slot(sl);
      try {
        g->potrf(g->A2.d(), g->Linv2.d(), static_cast<int*>(g->info.p));
      } catch (...) {
        g->swap_slot(sl);
        throw;
      }
      g->swap_slot(sl);
      HIPCHK(hipEventRecord(done[i], sl.stream));
      HIPCHK(hipStreamWaitEvent(g->stream, done[i], 0));
    }
    HIPCHK(hipEventRecord(e1, g->stream));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 0) total += t;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (auto& e : done) (void)hipEventDestroy(e);
  *ms = total / reps;
  return BOBE_OK;
  API_END
}

// B factorisations advancing in lock step through one batched launch sequence (the fit's restarts from
// lockstep_min_n points up): device time of the whole batch, averaged over reps
int bobe_debug_time_potrf_lockstep(bobe_gp_t* g, int B, int reps, double* ms) {
  API_BEGIN
  if (!g || !ms || reps < 1 || B < 1 || B > BOBE_MAX_MLL_SLOTS) throw Err(BOBE_ERR_ARG, "bad argument");
  if (!g->have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  g->use();
  g->ensure_batch(B);
  const int64_t mat = g->Np * g->Np, xs = (int64_t)g->d * g->Np;
  for (int b = 0; b < B; ++b) g->bw.h_hyp[b] = g->hyp;
  HIPCHK(hipMemcpyAsync(g->bw.hyp.p, g->bw.h_hyp, (size_t)B * sizeof(Hyper), hipMemcpyHostToDevice, g->stream));
  const Hyper* hdev = static_cast<const Hyper*>(g->bw.hyp.p);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  double total = 0.0;
  g->scale(g->X.d(), g->N, g->Np, g->hyp, g->bw.XsT.d(), g->Np, hdev, B, xs);
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed: first touch of the workspace, clocks)
    g->assemble_kxx(g->hyp, g->bw.XsT.d(), g->bw.A.d(), hdev, B, xs, mat);
    HIPCHK(hipMemsetAsync(g->bw.info.p, 0x7f, (size_t)B * sizeof(int), g->stream));
    HIPCHK(hipEventRecord(e0, g->stream));
    g->potrf(g->bw.A.d(), g->bw.Linv.d(), static_cast<int*>(g->bw.info.p), B, mat, mat, g->bw.diag.d());
    HIPCHK(hipEventRecord(e1, g->stream));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 0) total += t;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = total / reps;
  return BOBE_OK;
  API_END
}

}  // extern "C"
