// libbobe_gp.so, factorisation unit: K(X,X) assembly, the blocked Cholesky and its launch plan, the triangular inverse,
// alpha, the marginal likelihood and its gradient (single evaluation, evaluation slots, lock-step batch, graph replay),
// restore of a given factor.  Kernels: kernels_common.hpp, chol_kernels.hpp.
#include "gp_handle.hpp"

#include <chrono>

#include "chol_kernels.hpp"

using namespace bobe;

namespace bobe {

constexpr int SYRK32_BK = 128, SYRK64_BK = 16;   // BK = 128 = the whole panel: one stage, one LDS buffer
constexpr int SYRK32_SMEM = gemm_smem_doubles_exact<KC, KC, 32, 32, SYRK32_BK>() * 8 / 2;  //  66,560 B (single buffer)
constexpr int SYRK64_SMEM = gemm_smem_doubles_exact<KC, KC, 64, 64, SYRK64_BK>() * 8;      //  36,864 B -> four workgroups per CU

void configure_factor_kernels() {
  static bool done[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return;
  allow_big_lds(k_potf2<true>, POTF2_SMEM_BYTES);
  allow_big_lds(k_potf2<false>, POTF2_SMEM_BYTES);
  allow_big_lds(k_trti_diag, POTF2_SMEM_BYTES);
  allow_big_lds(k_trsm_panel, TRSM_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, 3>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<false, 4>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<true, 3>), POTF2_SMEM_BYTES);
  allow_big_lds((k_chol_panel<true, 4>), POTF2_SMEM_BYTES);
  allow_big_lds(k_trtri_T<128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_trtri_R<128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_syrk_trail<64, SYRK64_BK>, SYRK64_SMEM);
  allow_big_lds(k_syrk_trail<32, SYRK32_BK>, SYRK32_SMEM);
  allow_big_lds(k_trtri_T<64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trtri_R<64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_trtri_T<32>, GEMM32_SMEM_BYTES);
  allow_big_lds(k_trtri_R<32>, GEMM32_SMEM_BYTES);
  allow_big_lds(k_lauum_tiles32, GEMM32_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 8, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 16, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 32, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 8, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 16, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 32, 64>, GEMM64_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 8, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 16, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<0, 32, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 8, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 16, 128>, GEMM_SMEM_BYTES);
  allow_big_lds(k_lauum_grad<1, 32, 128>, GEMM_SMEM_BYTES);
  done[dev] = true;
}

const Tuning& tuning() {
  static Tuning t = [] {
    Tuning v{512, 600, 300, 1, BOBE_MAX_MLL_SLOTS, 2048, 1, 1, false, false};
    auto geti = [](const char* name, int& dst) {
      const char* e = std::getenv(name);
      if (e) dst = std::atoi(e);
      return e != nullptr;
    };
    geti("BOBE_SYRK32_BELOW", v.syrk32_below);
    geti("BOBE_TRTRI64", v.trtri64_below);
    geti("BOBE_PAIR_MIN", v.pair_min);
    geti("BOBE_LOCKSTEP_MIN_N", v.lockstep_min_n);
    v.mll_slots_set = geti("BOBE_MLL_SLOTS", v.mll_slots);
    geti("BOBE_GRAPH_MAX_N", v.graph_max_n);
    geti("BOBE_XCD_SHARES", v.xcd_shares);
    geti("BOBE_FILL", v.fill);
    v.trace = std::getenv("BOBE_TRACE") != nullptr;
    return v;
  }();
  return t;
}

double default_refine_kappa() {
  static const double v = [] {
    const char* e = std::getenv("BOBE_REFINE_KAPPA");
    return e ? std::atof(e) : 1e6;
  }();
  return v;
}

int default_solve_block() {
  static const int v = [] {
    const char* e = std::getenv("BOBE_SOLVE_BLOCK");
    const int b = e ? std::atoi(e) : TILE;
    return b > 0 ? (b + TILE - 1) / TILE * TILE : TILE;
  }();
  return v;
}

int default_solve_panel() {
  static const int v = [] {
    const char* e = std::getenv("BOBE_SOLVE_PANEL");
    const int b = e ? std::atoi(e) : 512;
    return b > 0 ? (b + TILE - 1) / TILE * TILE : 512;
  }();
  return v;
}

int64_t default_solve_chunk() {
  static const int64_t v = [] {
    const char* e = std::getenv("BOBE_SOLVE_CHUNK");
    const int64_t c = e ? std::atoll(e) : 32768;       // (K(X, chunk) and V: 1 GiB each at N = 4096)
    return c > 0 ? (c + TILE - 1) / TILE * TILE : 0;
  }();
  return v;
}

double default_pivot_floor_ulp() {
  static const double v = [] {
    const char* e = std::getenv("BOBE_PIVOT_FLOOR_ULP");
    if (!e) return 0.0;
    const double u = std::atof(e);
    return u >= 0.0 ? u : 0.0;
  }();
  return v;
}

}  // namespace bobe

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
  build_plans();
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
    } else if (!SQ && h.kern == 2) {   /* dist_sq: cross form only */                                 \
      if (dc_ == 8) KM_LAUNCH(2, false, 8, grid, __VA_ARGS__);                                        \
      else if (dc_ == 16) KM_LAUNCH(2, false, 16, grid, __VA_ARGS__);                                 \
      else KM_LAUNCH(2, false, 32, grid, __VA_ARGS__);                                                \
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
  // variants at every K (profiles/r02_a_ubench_update_variants.txt); when they would leave most of the chip idle, a single-panel
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

int bobe_gp::panel_strips(int B, int rr) const {
  // three 16-row strips per panel workgroup wherever the launch still fits the chip (a third more workgroups), else four
  return B * panel_workgroups(rr, 3) <= std::max(num_cus, 1) ? 3 : 4;
}

// Where deferred updates in the panel launches pay.  Measured on 256 CUs (profiles/r04_fill_rule.txt: a lone
// factorisation, fillers forced on / off): N = 2048 1.008, 2560 0.990, 3072 1.017, 3584 0.989, 3840 0.974, 4096 0.965,
// 4352 0.966, 4608 1.040, 4992 1.136, 5120 1.140, 6144 1.157, 8192 1.066.  They pay in a narrow band: the matrix must have
// enough block columns for the update launches to be worth replacing (below ~28 they last a few microseconds each and the
// difference is inside the noise), and the FIRST panel launch - the widest - must leave about two thirds of the chip free
// (from 36 block columns up it occupies 94+ of 256 CUs, whole block columns of update tiles no longer fit beside the chain,
// a filler workgroup runs the tile core at ~3/4 of its usual rate and every launch lasts as long as its slowest filler).
// Hence: nb >= 28 and B * (panel workgroups of step 0) <= 35 % of the CUs - on 256 CUs a lone factorisation of 28 ... 34
// block columns (N = 3457 ... 4352), never a lock-step batch (B >= 2 fails the second test for every nb >= 28).  Never on
// an evaluation slot's private stream, where the other slots' kernels want those CUs.  BOBE_FILL=2 forces the fillers on
// everywhere, BOBE_FILL=0 off.
bool bobe_gp::fill_pays(int B) const {
  const int f = tuning().fill;
  if (f == 0 || in_slot) return false;
  if (f == 2) return true;
  return nb >= 28 && 20 * B * panel_workgroups(nb - 1, panel_strips(B, nb - 1)) <= 7 * std::max(num_cus, 1);
}

// The launch plan of a factorisation of B matrices in lock step (see potrf).  Block columns >= far_start are DEFERRED:
// the update launches leave them alone until they are about to be factored (the last FILL_NEAR panels of a column
// always come from the update launches), and the panel launches carry their pending updates as filler workgroups on the
// CUs the panel workgroups do not occupy - whole block columns at a time, FILL_CHUNK panels per visit (a filler must
// not outlast the panel, ~28 us), earliest deadline first.  far_start is the smallest column from which the fillers keep
// up (what they leave behind is caught up by the update launch that makes the column current, with a longer K range).
const bobe_gp::CholPlan& bobe_gp::chol_plan(int B, bool fill) {
  const Tuning& tu = tuning();
  const uint64_t key = ((uint64_t)nb << 32) | ((uint64_t)B << 8) | (fill ? 1u : 0u);
  auto it = chol_plans.find(key);
  if (it != chol_plans.end()) return it->second;
  const int ncu = std::max(num_cus, 1);
  constexpr int D = FILL_NEAR, CH = FILL_CHUNK;
  auto npanel = [&](int k) { return panel_workgroups(nb - 1 - k, panel_strips(B, nb - 1 - k)); };
  auto one_launch = [&](int k) { return B * npanel(k) <= ncu; };
  auto tiles_of = [&](int c) { return 4 * (nb - c) - 1; };   // 64 x 64 tiles of block column c from its diagonal block down

  auto build = [&](int far, CholPlan* out) {
    std::vector<int> applied(nb, 0);
    int64_t deferred = 0, catchup = 0, total = 0;
    auto panel = [&](int k) {
      CholOp op{0, k, k, out ? (int)out->jobs.size() : 0, 0, 0, 0, true, k, k};
      const int64_t remk = nb - 1 - k;
      int cap = one_launch(k) ? (ncu - B * npanel(k)) / B : 0;   // filler workgroups per slot, two jobs each
      if (fill && far < nb && cap > 0 && (int64_t)B * remk * remk <= FILL_PHASE) {
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
              for (int ti = tj; ti < 2 * nb; ++ti) out->jobs.push_back({ti, tj, 2 * applied[c], 2 * k1, 0, 0});
            FillJob twin = out->jobs.back();                             // an odd count: a twin that is computed, not stored,
            twin.flags |= FILL_TWIN;                                     // keeps the two groups of a workgroup in step
            out->jobs.push_back(twin);
            op.tab_cnt += tiles_of(c) + 1;
          }
          applied[c] = k1;
        }
      }
      if (out) out->ops.push_back(op);
    };
    auto update = [&](int kind, int first, int k1, int last_near, int newest) {
      // columns taking part: the block column `first` alone (narrow) or every column from `first` on that is not deferred
      // or is within FILL_NEAR panels of being factored (c <= last_near)
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
        total += (int64_t)tiles_of(c) * (k1 - applied[c]);
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
      if (tu.pair_min > 0 && rem >= 2 && (int64_t)B * rem * rem > tu.pair_min) {
        update(1, k + 1, k + 1, nb, 1);
        panel(k + 1);
        update(2, k + 2, k + 2, k + 1 + D, 2);
        k += 2;
      } else {
        if (rem > 0) update(2, k + 1, k + 1, k + D, 1);
        k += 1;
      }
    }
    if (out) {
      out->far_start = far;
      out->deferred_units = deferred;
      out->catchup_units = catchup;
      out->total_units = total + deferred;
    }
    return std::make_pair(deferred, catchup);
  };
  int far = nb;
  if (fill) {
    for (int f = 1; f < nb; ++f) {
      const auto dc = build(f, nullptr);
      if (dc.first > 0 && dc.second * FILL_SLACK <= dc.first) { far = f; break; }
    }
  }
  CholPlan& pl = chol_plans[key];
  build(far, &pl);
  for (const FillJob& j : pl.jobs)             // (the tables drive device addresses: check them on the host)
    if (!(j.ti >= j.tj && j.ti < 2 * nb && j.tj >= 0 && j.k0 >= 0 && j.k1 > j.k0 && j.k1 <= j.tj - (j.tj & 1)))
      throw Err(BOBE_ERR_STATE, "internal error: filler job outside the matrix");
  if (!pl.jobs.empty()) {
    pl.d_jobs.ensure(pl.jobs.size() * sizeof(FillJob));
    HIPCHK(hipMemcpy(pl.d_jobs.p, pl.jobs.data(), pl.jobs.size() * sizeof(FillJob), hipMemcpyHostToDevice));
  }
  bool need_tab = false;                      // (a plan without deferred columns needs no device table: its launches are uniform)
  for (const CholOp& op : pl.ops) need_tab = need_tab || (op.kind != 0 && !op.uniform);
  if (need_tab) {
    pl.d_colk0.ensure(pl.colk0.size() * sizeof(int));
    HIPCHK(hipMemcpy(pl.d_colk0.p, pl.colk0.data(), pl.colk0.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (tu.trace)
    std::fprintf(stderr, "[bobe] chol plan nb=%d B=%d fill=%d: deferred columns from %d, %lld of %lld tile-panels in fillers, "
                 "%lld caught up (%zu jobs)\n", nb, B, (int)fill, pl.far_start, (long long)pl.deferred_units,
                 (long long)pl.total_units, (long long)pl.catchup_units, pl.jobs.size());
  return pl;
}

// The plans the evaluation paths will ask for, built where allocation is allowed (a plan uploads its tables with
// hipMalloc + a synchronous copy: not on the evaluation path, and never while a slot's stream is capturing a graph)
void bobe_gp::build_plans() {
  if (nb <= 0) return;
  (void)chol_plan(1, false);
  if (!in_slot && fill_pays(1)) (void)chol_plan(1, true);
}

// Blocked right-looking Cholesky (NB = 128) of B matrices in lock step on one stream; every launch carries the slot
// in its last grid dimension.  Per step k:
//   panel k   diagonal factor + solve of the rows below.  One launch (k_chol_panel: every 48- or 64-row workgroup factors
//             the diagonal block itself) while all B * npanel workgroups fit on the chip at once, else the
//             k_potf2 + k_trsm_panel pair (one factorisation per slot).
//   update    A22 -= L21 L21^T on the lower tiles (k_syrk_trail).
//             Update-bound steps go in pairs: one K = 256 pass for two panels (chol_plan).
// A batch shares the latency-bound panel chain (32 x ~27 us at N = 4096, the same for 1 or 8 matrices) and gives the
// update 4-8x the tiles: 44 % of the fp64 MFMA peak with four in flight, 50 % with eight, against 21 % alone and 28 %
// for four on private streams (whose 150 KB-LDS panel kernels wait for a CU the others' update tiles keep occupied).
// The panel launches leave most of the chip empty (one 150 KB workgroup per 48 / 64 rows): where that pays (fill_pays),
// updates of block columns that are not needed soon are DEFERRED and ride in those launches as filler workgroups
// (chol_plan, k_chol_panel<true, .>).  Every matrix element sees the same operation sequence in all forms (same bits).
void bobe_gp::potrf(double* a, double* linv, int* info_dev, int B, int64_t bsA, int64_t bsL, double* dg, bool defer_diag) {
  if (!dg) dg = diag.d();                                     // scratch for the L_kk of the panel launches
  const int64_t bsD = (int64_t)nb * TILE * TILE;
  int first_aside = nb;                                       // first step whose L_kk was left in the scratch blocks
  const CholPlan& pl = chol_plan(B, fill_pays(B));
  const FillJob* jobs = static_cast<const FillJob*>(pl.d_jobs.p);
  const int* coltab = static_cast<const int*>(pl.d_colk0.p);
  for (const CholOp& op : pl.ops) {
    if (op.kind == 0) {
      const int kk = op.k;
      const int rr = nb - 1 - kk;
      const int strips = panel_strips(B, rr);
      const int np_ = panel_workgroups(rr, strips);
      const int rows_below = rr * TILE;
      const int nv = (int)std::min<int64_t>(TILE, N - (int64_t)kk * TILE);
      if (B * np_ <= std::max(num_cus, 1)) {
        first_aside = std::min(first_aside, kk);
        prof_begin(BOBE_PROF_POTF2);
#define PANEL_LAUNCH(FILLV, GRIDX, ...)                                                                                        \
  do {                                                                                                                         \
    if (strips == 3)                                                                                                           \
      hipLaunchKernelGGL((k_chol_panel<FILLV, 3>), dim3(GRIDX, B), dim3(PANEL_THREADS), POTF2_SMEM_BYTES, stream, a, Np, bsA, \
                         linv, Np, bsL, kk, np_, info_dev, nv, dg, bsD, __VA_ARGS__);                                          \
    else                                                                                                                       \
      hipLaunchKernelGGL((k_chol_panel<FILLV, 4>), dim3(GRIDX, B), dim3(PANEL_THREADS), POTF2_SMEM_BYTES, stream, a, Np, bsA, \
                         linv, Np, bsL, kk, np_, info_dev, nv, dg, bsD, __VA_ARGS__);                                          \
  } while (0)
        if (op.tab_cnt > 0) PANEL_LAUNCH(true, np_ + op.tab_cnt / 2, jobs + op.tab_off, op.tab_cnt, rows_below);
        else PANEL_LAUNCH(false, np_, (const FillJob*)nullptr, 0, rows_below);
#undef PANEL_LAUNCH
        prof_end(BOBE_PROF_POTF2);
      } else {
        prof_begin(BOBE_PROF_POTF2);
        hipLaunchKernelGGL(k_potf2<true>, dim3(B), dim3(256), POTF2_SMEM_BYTES, stream, a, Np, linv, Np, kk, info_dev, nv, bsA,
                           bsL);
        prof_end(BOBE_PROF_POTF2);
        if (rr > 0) {
          prof_begin(BOBE_PROF_TRSM);
          hipLaunchKernelGGL(k_trsm_panel, dim3(2 * rr, B), dim3(256), TRSM_SMEM_BYTES, stream, a, Np, (const double*)linv, Np,
                             kk, bsA, bsL);
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
}

// Linv = L^-1: diagonal 128-blocks in one batched launch, then recursive doubling (two GEMM launches per level)
void bobe_gp::trtri(double* a, double* linv, double* tmp, int B, int64_t bsA, int64_t bsL, int64_t bsT) {
  const Tuning& tu = tuning();
  prof_begin(BOBE_PROF_TRTRI);
  hipLaunchKernelGGL(k_trti_diag, dim3(nb, B), dim3(256), POTF2_SMEM_BYTES, stream, a, Np, linv, Np, bsA, bsL, aside_dg,
                     (int64_t)nb * TILE * TILE, aside_first);
  aside_first = 1 << 30;
  aside_dg = nullptr;
  prof_end(BOBE_PROF_TRTRI);
  for (int dd = (int)depths.size() - 1; dd >= 0; --dd) {
    const Depth& D = depths[dd];
    const TriProb* pr = static_cast<const TriProb*>(probs.p) + D.first;
    prof_begin(BOBE_PROF_TRTRI);
    // (64x64 tiles while a level of ONE matrix has too few 128x128 tiles to fill the chip; a tile's K order is the
    // same either way.  Batches keep the per-matrix choice: four in lock step at N = 4096 take 7.0 ms per evaluation
    // round with 64x64 tiles at every level against 7.4 with 128x128 tiles at the top level)
    if (B * 2 * D.nblocks <= std::max(num_cus, 1)) {
      // A level with fewer 64 x 64 tile pairs than CUs (N <= 1024 in a batch of four, the deep levels of larger
      // matrices): 32 x 32 tiles - four times the workgroups, a quarter of the work each.  Such a launch lasts as long as
      // ONE workgroup's K loop, and a wave issues a 16 x 16 x 4 fp64 MFMA every 64 cycles at best: a quarter of the MFMAs
      // per wave and K-step is a shorter launch.  Tile shape does not change an element's accumulation order (same bits).
      const TileGrid tg = tile_grid(8 * D.nblocks);
      hipLaunchKernelGGL(k_trtri_T<32>, dim3(tg.grid, B), dim3(256), GEMM32_SMEM_BYTES, stream, a, Np,
                         (const double*)linv, Np, tmp, Np, pr, D.count, bsA, bsL, bsT, tg.per);
      hipLaunchKernelGGL(k_trtri_R<32>, dim3(tg.grid, B), dim3(256), GEMM32_SMEM_BYTES, stream, linv, Np,
                         (const double*)tmp, Np, pr, D.count, bsL, bsT, tg.per);
    } else if (D.nblocks < tu.trtri64_below) {
      const TileGrid tg = tile_grid(2 * D.nblocks);             // (tile pairs of complementary K: equal work)
      hipLaunchKernelGGL(k_trtri_T<64>, dim3(tg.grid, B), dim3(256), GEMM64_SMEM_BYTES, stream, a, Np,
                         (const double*)linv, Np, tmp, Np, pr, D.count, bsA, bsL, bsT, tg.per);
      hipLaunchKernelGGL(k_trtri_R<64>, dim3(tg.grid, B), dim3(256), GEMM64_SMEM_BYTES, stream, linv, Np,
                         (const double*)tmp, Np, pr, D.count, bsL, bsT, tg.per);
    } else {
      hipLaunchKernelGGL(k_trtri_T<128>, dim3(D.nblocks, B), dim3(256), GEMM_SMEM_BYTES, stream, a, Np,
                         (const double*)linv, Np, tmp, Np, pr, D.count, bsA, bsL, bsT);
      hipLaunchKernelGGL(k_trtri_R<128>, dim3(D.nblocks, B), dim3(256), GEMM_SMEM_BYTES, stream, linv, Np,
                         (const double*)tmp, Np, pr, D.count, bsL, bsT);
    }
    prof_end(BOBE_PROF_TRTRI);
  }
  LAUNCH_CHECK();
}

// K^-1 tiles fused with the gradient partial sums (optionally stores K^-1's lower tiles); returns #partials.
// scratch (an Np x Np matrix per slot, stride bsS): small launches - fewer 64 x 64 tiles than four per CU - form K^-1 on
// 32 x 32 tiles into it first (k_lauum_tiles) and run the gradient epilogue from there: N = 1024 in a batch of four,
// 77 -> 36 us (the fused launch is as long as its longest tile, K = 1024 on one CU).  Same partial sums, same bits.
int bobe_gp::lauum(const Hyper& h, const double* linv, const double* al, const double* xst, double* kinv_out, int dcap,
                   const Hyper* hdev, double* gp_out, int B, int64_t bsL, int64_t bsV, int64_t bsX, int64_t bsP,
                   double* scratch, int64_t bsS) {
  // (the tile size fixes the order of the gradient's partial sums: it depends on N only, so that an evaluation
  // returns the same bits alone, on a slot and in a batch)
  const bool small = nb * (nb + 1) / 2 < LAUUM64_BELOW;
  const int nt = small ? 2 * nb : nb;
  const int ntiles = nt * (nt + 1) / 2;
  double* gpo = gp_out ? gp_out : gpart.d();
  const bool split = small && scratch && !kinv_out && B * ntiles < 4 * std::max(num_cus, 1);
  double* kio = split ? scratch : kinv_out;
  const int64_t bsK = split ? bsS : 0;
  prof_begin(BOBE_PROF_LAUUM);
  if (split) {
    hipLaunchKernelGGL(k_lauum_tiles32, dim3(4 * ntiles * B), dim3(256), GEMM32_SMEM_BYTES, stream, linv, Np, Np, scratch, Np,
                       bsL, bsS, B);
  }
#define LG(KE, DC, TT)                                                                                          \
  hipLaunchKernelGGL((k_lauum_grad<KE, DC, TT>), dim3(ntiles * B), dim3(256),                                   \
                     (TT == 128 ? GEMM_SMEM_BYTES : GEMM64_SMEM_BYTES), stream, linv, Np, Np, N, al, xst, Np, h, \
                     gpo, kio, Np, hdev, bsL, bsV, bsX, bsP, B, split ? 1 : 0, bsK)
#define LGD(KE, TT)                                                                 \
  do {                                                                              \
    if (dcap == 8) LG(KE, 8, TT); else if (dcap == 16) LG(KE, 16, TT); else LG(KE, 32, TT); \
  } while (0)
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
                          int64_t bsP, const double* rhs, int64_t bsY) {
  hipLaunchKernelGGL(k_gemv_lower, dim3((unsigned)(Np / 4), (unsigned)B), dim3(256), 0, stream, linv, Np, Np,
                     rhs ? rhs : (const double*)y.d(), wv, bsL, bsV, rhs ? bsY : (int64_t)0);
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
  potrf(a, linv, static_cast<int*>(info.p), 1, 0, 0, nullptr, true);
  trtri(a, linv, Tmp.d());
  solve_alpha(linv, wv, al, part.d());
}

// the text of a BOBE_NOT_PD status: a non-positive pivot (the factorisation's info word), or pivots below pivot_floor
std::string bobe_gp::not_pd_text(int inf, double min_diag) const {
  if (inf != 0x7f7f7f7f) return "kernel matrix not positive definite at column " + std::to_string(inf - 1);
  char buf[200];
  if (min_diag != min_diag)
    std::snprintf(buf, sizeof buf, "kernel matrix not positive definite: NaN pivot");
  else
    std::snprintf(buf, sizeof buf,
                  "kernel matrix numerically singular: smallest pivot %.3g is below %g ulp of its diagonal (rank test)",
                  min_diag * min_diag, pivot_ulp);
  return buf;
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
    std::vector<uint32_t> mask(words, 0u);
    for (int c = 0; c < num_cus; ++c) mask[c / 32] |= (1u << (c % 32));
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) {
      (void)hipGetLastError();
      st = nullptr;
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
    const int ntiles = lauum(h, Linv2.d(), alpha2.d(), XsT2.d(), nullptr, dcap, hdev, nullptr, 1, 0, 0, 0, 0, Tmp.d(), 0);
    hipLaunchKernelGGL(k_mll_grad_reduce, dim3(d + 2), dim3(256), 0, stream, (const double*)gpart.d(), ntiles, dcap + 1, d,
                       dcap, res.d(), (const double*)w2.d(), (const double*)A2.d(), Np, Np, (const int*)info.p);
  } else {
    hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)w2.d(), (const double*)A2.d(), Np, Np,
                       res.d(), (int64_t)0, (int64_t)0, (int64_t)0, (const int*)info.p);
  }
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 102 * sizeof(double), hipMemcpyDeviceToHost, stream));   // [100] = info, [101] = min L_jj
}

void bobe_gp::mll_enqueue(const Hyper& h, bool want_grad) {
  const Tuning& tu = tuning();
  h_res[110] = pivot_floor(h);              // (host side of the pinned block, beyond what the device copy writes: read at collect)
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
  if (inf != 0x7f7f7f7f || !pivots_resolved(hr[101], hr[110])) {
    *mll = std::nan("");
    if (grad)
      for (int j = 0; j <= d; ++j) grad[j] = std::nan("");
    g_err = not_pd_text(inf, hr[101]);
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
  if (inf != 0x7f7f7f7f || !pivots_resolved(h_res[101], h_res[110])) {
    *mll = std::nan("");
    if (grad)
      for (int j = 0; j <= d; ++j) grad[j] = std::nan("");
    g_err = not_pd_text(inf, h_res[101]);
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
  (void)chol_plan(B, fill_pays(B));      // (uploads its tables on first use: here, not between the launches of a batch)
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
  potrf(bw.A.d(), bw.Linv.d(), inf, B, mat, mat, bw.diag.d(), true);
  trtri(bw.A.d(), bw.Linv.d(), bw.Tmp.d(), B, mat, mat, mat);
  solve_alpha(bw.Linv.d(), bw.w.d(), bw.alpha.d(), bw.part.d(), B, mat, vec, prt);
  // (the info word of slot b rides in res[b * 128 + 100]: one copy brings everything to the host)
  if (want_grad) {
    const int dcap = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
    const int ntiles = lauum(hs[0], bw.Linv.d(), bw.alpha.d(), bw.XsT.d(), nullptr, dcap, hdev, bw.gpart.d(), B, mat, vec,
                             xs, gps, bw.Tmp.d(), mat);
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
    if (inf_b != 0x7f7f7f7f || !pivots_resolved(hr[101], pivot_floor(bw.h_hyp[b]))) {
      mll[b] = std::nan("");
      if (gb)
        for (int j = 0; j <= d; ++j) gb[j] = std::nan("");
      g_err = not_pd_text(inf_b, hr[101]);
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

// ---- entry-point bodies of this unit (the extern "C" layer is gp_abi.hip) ----------------------------------------------
void bobe_gp::fill(double* p, int64_t n, double v) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, v);
}

void bobe_gp::set_data(const double* Xin, const double* ys, int64_t n) {
  use();
  sync();
  const int64_t np_new = round_up(n, TILE);
  N = n;
  if (np_new != Np) {
    Np = np_new;
    nb = (int)(Np / TILE);
    alloc_for_n();
  }
  X.ensure((size_t)(Np + TILE) * d * sizeof(double));     // (room for bobe_gp_append's rows up to the next block and one more)
  HIPCHK(hipMemcpyAsync(X.p, Xin, (size_t)N * d * sizeof(double),
                        is_device_ptr(Xin) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(y.p, 0, (size_t)Np * sizeof(double), stream));
  HIPCHK(hipMemcpyAsync(y.p, ys, (size_t)N * sizeof(double),
                        is_device_ptr(ys) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  sync();
  have_data = true;
  factored = false;
  forget_z();
}

int bobe_gp::factor_state() {
  use();
  factor_into(hyp, XsT.d(), A.d(), Linv.d(), w.d(), alpha.d());
  // the info word and the smallest pivot's root in one copy (k_mll_terms: res[100], res[101])
  hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)w.d(), (const double*)A.d(), Np, Np, res.d(),
                     (int64_t)0, (int64_t)0, (int64_t)0, (const int*)info.p);
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 102 * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync();
  int inf;
  std::memcpy(&inf, h_res + 100, sizeof(int));
  const double min_diag = h_res[101];
  factored = true;
  forget_z();
  not_pd = (inf != 0x7f7f7f7f) || !pivots_resolved(min_diag, pivot_floor(hyp));
  if (not_pd) {
    const double nan = std::nan("");
    fill(A.d(), Np * Np, nan);
    fill(Linv.d(), Np * Np, nan);
    fill(alpha.d(), Np, nan);
    LAUNCH_CHECK();
    sync();
    g_err = not_pd_text(inf, min_diag);
    refine_v = false;
    return BOBE_NOT_PD;
  }
  decide_refinement(min_diag);
  return BOBE_OK;
}

int bobe_gp::mll_batch(int64_t B, const double* ls, const double* kvar, double* mll, double* grad, int* status) {
  use();
  const Tuning& tu = tuning();
  int worst = BOBE_OK;
  if (B >= 2 && N >= tu.lockstep_min_n) {      // (kernel-class event timing works there too: one stream, no capture)
    // The evaluations advance in lock step through ONE batched launch sequence.  At every size (round 4, with the launches
    // of small matrices no longer as long as one workgroup's K loop): the 10-D Rosenbrock loop's fits 3.0 -> 2.2 s, the 2-D
    // examples 0.8-1.2 -> 0.4-0.8 s against one graph replay per evaluation on private streams (BOBE_LOCKSTEP_MIN_N = 1024,
    // the default until then); same bits either way.
    for (int64_t b0 = 0; b0 < B; b0 += BOBE_MAX_MLL_SLOTS) {
      const int nbat = (int)std::min<int64_t>(BOBE_MAX_MLL_SLOTS, B - b0);
      Hyper hs[BOBE_MAX_MLL_SLOTS];
      for (int i = 0; i < nbat; ++i) {
        hs[i] = hyp;
        for (int j = 0; j < d; ++j) hs[i].ls[j] = ls[(b0 + i) * d + j];
        hs[i].kvar = kvar[b0 + i];
      }
      mll_lockstep_enqueue(nbat, hs, grad != nullptr);
      const int st = mll_lockstep_collect(nbat, mll + b0, grad ? grad + b0 * (d + 1) : nullptr, status ? status + b0 : nullptr);
      if (st != BOBE_OK) worst = st;
    }
    return worst;
  }
  // (batches below BOBE_LOCKSTEP_MIN_N points, when that is raised: one evaluation slot - private stream, workspace, graph
  //  replay - per member, BOBE_MLL_SLOTS at a time; also what bobe_gp_mll_submit / _wait run on)
  const int width = std::max(1, std::min<int>(tu.mll_slots, BOBE_MAX_MLL_SLOTS));
  for (int64_t b0 = 0; b0 < B; b0 += width) {
    const int nbat = (int)std::min<int64_t>(width, B - b0);
    const auto t_start = std::chrono::steady_clock::now();
    if (nbat == 1) {   // a lone evaluation owns the whole chip on the handle's stream
      Hyper h = hyp;
      for (int j = 0; j < d; ++j) h.ls[j] = ls[b0 * d + j];
      h.kvar = kvar[b0];
      mll_enqueue(h, grad != nullptr);
    } else {
      ensure_slots(nbat);
      for (int i = 0; i < nbat; ++i)
        if (slots[i]->busy)
          throw Err(BOBE_ERR_STATE, "an evaluation submitted with bobe_gp_mll_submit is still in flight on a slot this batch needs");
      const std::vector<hipStream_t>& sts = slot_stream_set();
      // the batch streams start after everything already queued on the handle's stream (data uploads)
      HIPCHK(hipEventRecord(ev_batch, stream));
      for (int i = 0; i < nbat; ++i) {
        Hyper h = hyp;
        for (int j = 0; j < d; ++j) h.ls[j] = ls[(b0 + i) * d + j];
        h.kvar = kvar[b0 + i];
        Slot& sl = *slots[i];
        sl.stream = sts[i];
        HIPCHK(hipStreamWaitEvent(sl.stream, ev_batch, 0));
        swap_slot(sl);
        try {
          mll_enqueue(h, grad != nullptr);
        } catch (...) {
          swap_slot(sl);
          throw;
        }
        swap_slot(sl);
      }
    }
    const auto t_enq = std::chrono::steady_clock::now();
    for (int i = 0; i < nbat; ++i) {
      double* gi = grad ? grad + (b0 + i) * (d + 1) : nullptr;
      int st;
      if (nbat == 1) {
        st = mll_collect(mll + b0, gi);
      } else {
        Slot& sl = *slots[i];
        swap_slot(sl);
        try {
          st = mll_collect(mll + b0 + i, gi);
        } catch (...) {
          swap_slot(sl);
          throw;
        }
        swap_slot(sl);
      }
      if (status) status[b0 + i] = st;
      if (st != BOBE_OK) worst = st;
    }
    if (tu.trace) {
      const auto t_end = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[bobe] mll_batch B=%d: enqueue %.3f ms, total %.3f ms\n", nbat,
                   std::chrono::duration<double, std::milli>(t_enq - t_start).count(),
                   std::chrono::duration<double, std::milli>(t_end - t_start).count());
    }
  }
  return worst;
}

void bobe_gp::mll_submit(int slot, const double* ls, double kvar, int want_grad) {
  std::lock_guard<std::mutex> lock(submit_mutex);
  use();
  ensure_slots(slot + 1);
  const std::vector<hipStream_t>& sts = slot_stream_set();
  Hyper h = hyp;
  for (int j = 0; j < d; ++j) h.ls[j] = ls[j];
  h.kvar = kvar;
  Slot& sl = *slots[slot];
  if (sl.busy)      // its pinned inputs / workspace / results are still in use by the evaluation not yet collected
    throw Err(BOBE_ERR_STATE, "slot already has an evaluation in flight: call bobe_gp_mll_wait first");
  sl.stream = sts[slot];
  // ordered after whatever is queued on the handle's stream (data uploads); the event is private to the slot
  if (!sl.ev) HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(sl.ev, stream));
  HIPCHK(hipStreamWaitEvent(sl.stream, sl.ev, 0));
  swap_slot(sl);
  try {
    mll_enqueue(h, want_grad != 0);
  } catch (...) {
    swap_slot(sl);
    throw;
  }
  swap_slot(sl);
  sl.busy = true;
  sl.want_grad = want_grad != 0;
}

int bobe_gp::mll_wait(int slot, double* mll, double* grad) {
  Slot* slp = nullptr;
  {
    std::lock_guard<std::mutex> lock(submit_mutex);    // (the slot table grows under this mutex)
    if (slot < 0 || slot >= (int)slots.size() || !slots[slot]->busy)
      throw Err(BOBE_ERR_STATE, "no evaluation was submitted to this slot");
    slp = slots[slot];
  }
  HIPCHK(hipSetDevice(device));
  Slot& sl = *slp;
  struct Release {                                        // the slot is free again once its stream has drained
    Slot& s;
    ~Release() { s.busy = false; }
  } release{sl};
  return slot_collect(sl, mll, sl.want_grad ? grad : nullptr);
}

void bobe_gp::copy_out_matrix(const double* src, double* dst, int lower_only) {
  double* d_out = out_dev(dst, (size_t)N * N, kout);
  hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, stream, src, Np, d_out, N, N, N,
                     lower_only);
  LAUNCH_CHECK();
  out_finish(dst, (size_t)N * N, kout);
}

void bobe_gp::get_chol(double* L, double* alpha_out) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  use();
  if (L) copy_out_matrix(A.d(), L, not_pd ? 0 : 1);   // not PD: all-NaN, like jnp.linalg.cholesky
  if (alpha_out)
    HIPCHK(hipMemcpyAsync(alpha_out, alpha.p, (size_t)N * sizeof(double),
                          is_device_ptr(alpha_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
  sync();
}

void bobe_gp::set_chol(const double* L, const double* alpha_in) {
  if (!have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  use();
  const double* l_in = fetch(L, (size_t)N * N, kout);
  hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, stream, l_in, N,
                     A.d(), Np, Np);
  HIPCHK(hipMemsetAsync(alpha.p, 0, (size_t)Np * sizeof(double), stream));
  HIPCHK(hipMemcpyAsync(alpha.p, alpha_in, (size_t)N * sizeof(double),
                        is_device_ptr(alpha_in) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  scale(X.d(), N, Np, hyp, XsT.d(), Np);
  for (int k = 0; k < nb; ++k)                       // the 16 x 16 diagonal inverses the block inverse starts from
    hipLaunchKernelGGL(k_potf2<false>, dim3(1), dim3(256), POTF2_SMEM_BYTES, stream, A.d(), Np, Linv.d(), Np, k,
                       static_cast<int*>(info.p));
  LAUNCH_CHECK();
  trtri(A.d(), Linv.d(), Tmp.d());
  // the restored factor's smallest pivot decides how the products with Linv are formed, as after a factorisation
  const double min_diag = min_pivot_root();
  factored = true;
  forget_z();
  not_pd = false;
  decide_refinement(min_diag);
}

// the smallest L_jj of the factor in A (k_mll_terms without a right-hand side); synchronises
double bobe_gp::min_pivot_root() {
  hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)nullptr, (const double*)A.d(), Np, Np, res.d(),
                     (int64_t)0, (int64_t)0, (int64_t)0, (const int*)nullptr);
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 102 * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync();
  return h_res[101];
}

// ---- the reference's free functions on caller-supplied matrices (gp.py:170-197), on a handle that holds no training data
void bobe_gp::size_workspace(int64_t n) {
  if (have_data && n != N) throw Err(BOBE_ERR_STATE, "the handle holds training data of another size (use a data-less handle)");
  if (have_data) return;
  const int64_t np_new = round_up(n, TILE);
  N = n;
  if (np_new != Np) {
    Np = np_new;
    nb = (int)(Np / TILE);
    alloc_for_n();
  }
}

// gp_mll(k, train_y, num_points) (gp.py:170-178): Cholesky of the matrix handed in, alpha, the three MLL terms.  The
// matrix has no kernel variance to scale a rank test with: a pivot <= 0 (or NaN) fails, nothing else - LAPACK's rule.
int bobe_gp::mll_from_k(const double* K, int64_t n, const double* yv, double* mll) {
  use();
  sync();
  size_workspace(n);
  const double* k_in = fetch(K, (size_t)n * n, kout);
  hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, stream, k_in, n,
                     A2.d(), Np, Np, 1);
  HIPCHK(hipMemsetAsync(w.p, 0, (size_t)Np * sizeof(double), stream));            // (w: the padded right-hand side)
  HIPCHK(hipMemcpyAsync(w.p, yv, (size_t)n * sizeof(double),
                        is_device_ptr(yv) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(info.p, 0x7f, sizeof(int), stream));
  potrf(A2.d(), Linv2.d(), static_cast<int*>(info.p), 1, 0, 0, nullptr, true);
  trtri(A2.d(), Linv2.d(), Tmp.d());
  solve_alpha(Linv2.d(), w2.d(), alpha2.d(), part.d(), 1, 0, 0, 0, w.d(), 0);
  hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)w2.d(), (const double*)A2.d(), Np, Np, res.d(),
                     (int64_t)0, (int64_t)0, (int64_t)0, (const int*)info.p);
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 102 * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync();
  int inf;
  std::memcpy(&inf, h_res + 100, sizeof(int));
  if (inf != 0x7f7f7f7f || !(h_res[101] > 0.0)) {
    *mll = std::nan("");
    g_err = not_pd_text(inf, h_res[101]);
    return BOBE_NOT_PD;
  }
  *mll = -0.5 * h_res[0] - h_res[1] - 0.5 * (double)n * std::log(2.0 * M_PI);
  return BOBE_OK;
}

// fast_update_cholesky(L, k, k_self) (gp.py:181-197): v = L^-1 k and the new diagonal entry sqrt(k_self - v.v) (NaN when
// that is negative, like jnp.sqrt); the caller owns L and lays out the (n+1) x (n+1) factor itself.
void bobe_gp::chol_row_update(const double* L, int64_t n, const double* k, double k_self, double* v, double* diag_out) {
  use();
  sync();
  size_workspace(n);
  const double* l_in = fetch(L, (size_t)n * n, kout);
  hipLaunchKernelGGL(k_load_padded_lower, dim3((unsigned)((Np + 255) / 256), (unsigned)Np), dim3(256), 0, stream, l_in, n,
                     A2.d(), Np, Np, 0);
  HIPCHK(hipMemsetAsync(w.p, 0, (size_t)Np * sizeof(double), stream));
  HIPCHK(hipMemcpyAsync(w.p, k, (size_t)n * sizeof(double),
                        is_device_ptr(k) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  for (int kb = 0; kb < nb; ++kb)                      // the 16 x 16 diagonal inverses the block inverse starts from
    hipLaunchKernelGGL(k_potf2<false>, dim3(1), dim3(256), POTF2_SMEM_BYTES, stream, A2.d(), Np, Linv2.d(), Np, kb,
                       static_cast<int*>(info.p));
  LAUNCH_CHECK();
  trtri(A2.d(), Linv2.d(), Tmp.d());
  hipLaunchKernelGGL(k_gemv_lower, dim3((unsigned)(Np / 4), 1u), dim3(256), 0, stream, (const double*)Linv2.d(), Np, Np,
                     (const double*)w.d(), w2.d(), (int64_t)0, (int64_t)0, (int64_t)0);
  hipLaunchKernelGGL(k_mll_terms, dim3(1), dim3(256), 0, stream, (const double*)w2.d(), (const double*)A2.d(), Np, Np, res.d(),
                     (int64_t)0, (int64_t)0, (int64_t)0, (const int*)nullptr);
  LAUNCH_CHECK();
  HIPCHK(hipMemcpyAsync(h_res, res.p, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));      // [0] = v.v
  HIPCHK(hipMemcpyAsync(v, w2.p, (size_t)n * sizeof(double),
                        is_device_ptr(v) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
  sync();
  *diag_out = std::sqrt(k_self - h_res[0]);
}

void bobe_gp::kinv_debug(double* Kinv) {
  if (!factored) throw Err(BOBE_ERR_STATE, "call bobe_gp_factor first");
  use();
  lauum(hyp, Linv.d(), alpha.d(), XsT.d(), Tmp.d(), 32);
  // symmetrise on the host side of the copy
  std::vector<double> full((size_t)N * N);
  HIPCHK(hipMemcpy2DAsync(full.data(), (size_t)N * 8, Tmp.p, (size_t)Np * 8, (size_t)N * 8, (size_t)N, hipMemcpyDeviceToHost,
                          stream));
  sync();
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = i + 1; j < N; ++j) full[i * N + j] = full[j * N + i];
  if (is_device_ptr(Kinv)) HIPCHK(hipMemcpy(Kinv, full.data(), full.size() * 8, hipMemcpyHostToDevice));
  else std::memcpy(Kinv, full.data(), full.size() * 8);
}

// ---- factorisation timers (bench.py's Cholesky figures): the launch sequence of potrf() - the one bobe_gp_factor and the
// evaluations issue, its update fillers included - between two HIP events, after one untimed pass
namespace {
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  EventPair() {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
  }
  ~EventPair() {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  double ms() {
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    return t;
  }
};
}  // namespace

double bobe_gp::time_potrf(int reps) {
  if (!have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  use();
  EventPair ev;
  double total = 0.0;
  scale(X.d(), N, Np, hyp, XsT2.d(), Np);
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed)
    assemble_kxx(hyp, XsT2.d(), A2.d());
    HIPCHK(hipMemsetAsync(info.p, 0x7f, sizeof(int), stream));
    HIPCHK(hipEventRecord(ev.e0, stream));
    potrf(A2.d(), Linv2.d(), static_cast<int*>(info.p));
    HIPCHK(hipEventRecord(ev.e1, stream));
    const double t = ev.ms();
    if (r >= 0) total += t;
  }
  return total / reps;
}

// B factorisations in flight at once, one evaluation slot each (the state of the fit's concurrent restarts):
// device time from the first to the last factorisation kernel, averaged over reps, for all B together
double bobe_gp::time_potrf_batch(int B, int reps) {
  if (!have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  use();
  ensure_slots(B);
  const std::vector<hipStream_t>& sts = slot_stream_set();
  EventPair ev;
  std::vector<hipEvent_t> done(B);
  for (auto& e : done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  double total = 0.0;
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed)
    for (int i = 0; i < B; ++i) {      // K(X,X) of every slot, on the handle's stream
      Slot& sl = *slots[i];
      scale(X.d(), N, Np, hyp, sl.XsT2.d(), Np);
      assemble_kxx(hyp, sl.XsT2.d(), sl.A2.d());
      HIPCHK(hipMemsetAsync(sl.info.p, 0x7f, sizeof(int), stream));
    }
    HIPCHK(hipEventRecord(ev.e0, stream));
    for (int i = 0; i < B; ++i) {
      Slot& sl = *slots[i];
      sl.stream = sts[i];
      HIPCHK(hipStreamWaitEvent(sl.stream, ev.e0, 0));
      swap_slot(sl);
      try {
        potrf(A2.d(), Linv2.d(), static_cast<int*>(info.p));
      } catch (...) {
        swap_slot(sl);
        throw;
      }
      swap_slot(sl);
      HIPCHK(hipEventRecord(done[i], sl.stream));
      HIPCHK(hipStreamWaitEvent(stream, done[i], 0));
    }
    HIPCHK(hipEventRecord(ev.e1, stream));
    const double t = ev.ms();
    if (r >= 0) total += t;
  }
  for (auto& e : done) (void)hipEventDestroy(e);
  return total / reps;
}

// the same B factorisations advancing in lock step through one batched launch sequence (the fit's restarts from
// lockstep_min_n points up)
double bobe_gp::time_potrf_lockstep(int B, int reps) {
  if (!have_data) throw Err(BOBE_ERR_STATE, "call bobe_gp_set_data first");
  use();
  ensure_batch(B);
  const int64_t mat = Np * Np, xs = (int64_t)d * Np;
  for (int b = 0; b < B; ++b) bw.h_hyp[b] = hyp;
  HIPCHK(hipMemcpyAsync(bw.hyp.p, bw.h_hyp, (size_t)B * sizeof(Hyper), hipMemcpyHostToDevice, stream));
  const Hyper* hdev = static_cast<const Hyper*>(bw.hyp.p);
  EventPair ev;
  double total = 0.0;
  scale(X.d(), N, Np, hyp, bw.XsT.d(), Np, hdev, B, xs);
  for (int r = -1; r < reps; ++r) {                            // (pass -1 is untimed: first touch of the workspace, clocks)
    assemble_kxx(hyp, bw.XsT.d(), bw.A.d(), hdev, B, xs, mat);
    HIPCHK(hipMemsetAsync(bw.info.p, 0x7f, (size_t)B * sizeof(int), stream));
    HIPCHK(hipEventRecord(ev.e0, stream));
    potrf(bw.A.d(), bw.Linv.d(), static_cast<int*>(bw.info.p), B, mat, mat, bw.diag.d());
    HIPCHK(hipEventRecord(ev.e1, stream));
    const double t = ev.ms();
    if (r >= 0) total += t;
  }
  return total / reps;
}
