/* bobe_gp.h — C ABI of libbobe_gp.so, the MI355X (gfx950) GP-surrogate engine for BOBE.
 *
 * Drop-in boundary for the hot path of Ameek94/BOBE (BOBE/gp.py + the integrated-variance half of
 * BOBE/acquisition.py).  The reference has no FFI of its own (it is pure Python on jax); the boundary
 * it would bind is the method set of its `GP` class, so every entry point below names the reference
 * method(s) it replaces (file:line relative to the reference tree).  bobe_amd/gp.py is the ctypes
 * binding that presents those methods again; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - all arrays are C-contiguous fp64 (int64 for indices); matrices are row-major;
 *   - every data pointer may be HOST memory (NumPy) or DEVICE memory (hipMalloc / torch.cuda
 *     tensor); the library detects which and copies only when needed;
 *   - y / alpha / mean / var are in STANDARDISED units (the wrapper applies y_mean, y_std exactly
 *     where gp.py does: gp.py:456, 466, 576);
 *   - return value: 0 ok; > 0 numerical condition (BOBE_NOT_PD: outputs are NaN, like XLA's
 *     Cholesky, so np.isfinite filters such as optim.py:328,341 keep working); < 0 usage / HIP error,
 *     text in bobe_last_error().  Nothing throws or aborts across this boundary;
 *   - BOBE_NOT_PD is raised by a pivot <= 0 or NaN: what LAPACK's dpotrf, the reference's Cholesky (gp.py:175, 549), reports.
 *     A handle can be told to refuse a positive pivot below `ulp` machine epsilons of the kernel matrix's diagonal k(x,x) +
 *     noise as well (bobe_gp_set_pivot_floor_ulp; process default BOBE_PIVOT_FLOOR_ULP, else 0 = off): such a pivot is the
 *     rounding of its column's update, the log-determinant built on it is too small, and a fit is drawn to exactly those
 *     hyper-parameters (dpotrf passes or fails on the last bit there).  The BO driver (bobe_amd/bo.py) runs its surrogate
 *     with ulp = 64; at the reference's default noise of 1e-8 that bounds the usable kernel variance near 1e6;
 *   - a handle is not thread-safe; distinct handles are independent (own stream).
 *   - limits: d <= 32.
 */
#ifndef BOBE_GP_H
#define BOBE_GP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bobe_gp bobe_gp_t;

#define BOBE_OK 0
#define BOBE_NOT_PD 1
#define BOBE_ERR_ARG (-1)
#define BOBE_ERR_HIP (-2)
#define BOBE_ERR_STATE (-3)

#define BOBE_KERNEL_RBF 0    /* gp.py:124-154 */
#define BOBE_KERNEL_MATERN 1 /* gp.py:156-168 */

/* version / build info ("bobe_gp <ver> gfx950") */
const char* bobe_version(void);
/* thread-local text of the last error on this thread */
const char* bobe_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int bobe_device_count(void);

/* GP.__init__ (gp.py:201-281): handle for one GP on one GPU.  kernel = BOBE_KERNEL_*; d = ndim. */
int bobe_gp_create(bobe_gp_t** out, int device, int kernel, int d);
void bobe_gp_destroy(bobe_gp_t* gp);

/* The HIP stream all work of this handle is issued on (void* = hipStream_t).  set: adopt a caller
 * stream (e.g. torch.cuda.current_stream().cuda_stream) so caller-side events see the kernels. */
void* bobe_gp_get_stream(bobe_gp_t* gp);
int bobe_gp_set_stream(bobe_gp_t* gp, void* hip_stream);
/* block until everything issued on the handle's stream has finished */
int bobe_gp_sync(bobe_gp_t* gp);

/* GP._setup_training_data / GP.update data part (gp.py:283-307, 529-539): X is N x d, y is the
 * already-standardised target vector (N).  Invalidates the factorisation. */
int bobe_gp_set_data(bobe_gp_t* gp, const double* X, const double* y_standardised, int64_t N);

/* GP.lengthscales / kernel_variance / noise assignment (gp.py:253-255, 444-447).  Does NOT refactor. */
int bobe_gp_set_hyper(bobe_gp_t* gp, const double* lengthscales, double kernel_variance, double noise);

/* GP.recompute_cholesky (gp.py:544-550): K = k(X,X)+noise I, L = chol(K), alpha = K^-1 y, and the
 * inverse factor used by every later call.  BOBE_NOT_PD -> L, alpha are NaN. */
int bobe_gp_factor(bobe_gp_t* gp);

/* gp_mll (gp.py:170-178) + its gradient, i.e. the data term of GP.neg_mll (gp.py:385-398) and of the
 * jax.value_and_grad closure of optim.py:306-309, at the given hyper-parameters (noise from
 * set_hyper).  *mll = -1/2 y^T K^-1 y - sum log L_ii - N/2 log 2 pi.  grad (may be NULL) has d+1
 * entries: d mll / d log ls_j (j < d), d mll / d log kernel_variance.  Does not disturb the state
 * left by bobe_gp_factor.  BOBE_NOT_PD -> *mll and grad are NaN. */
int bobe_gp_mll(bobe_gp_t* gp, const double* lengthscales, double kernel_variance, double* mll, double* grad);

/* bobe_gp_mll for B hyper-parameter vectors at once: lengthscales is B x d, kernel_variance has B entries,
 * mll B, grad (may be NULL) B x (d+1), status (may be NULL) B per-vector codes (BOBE_OK / BOBE_NOT_PD).
 * The restarts of optimize_scipy (optim.py:335-354) are independent L-BFGS-B runs which the reference walks
 * one after the other; here up to BOBE_MAX_MLL_SLOTS of their evaluations advance together through ONE launch
 * sequence on the handle's stream (every kernel takes the batch member from a grid dimension), with exactly the
 * arithmetic of bobe_gp_mll per member (bit-identical results).  Returns BOBE_NOT_PD when any vector was not
 * positive definite (its outputs are NaN), < 0 on usage / HIP errors. */
#define BOBE_MAX_MLL_SLOTS 8
int bobe_gp_mll_batch(bobe_gp_t* gp, int64_t B, const double* lengthscales, const double* kernel_variance,
                      double* mll, double* grad, int* status);

/* The same evaluation, split into a non-blocking submit to one of BOBE_MAX_MLL_SLOTS slots and a blocking wait,
 * so that every restart of the fit can run in its own host thread and advance at its own pace (no round
 * barrier): the restarts end up in different phases of the pipeline, the single-workgroup factorisation steps of
 * one overlapping the matrix-core phases of the others.  Results are those of bobe_gp_mll, bit for bit.
 * Threading: submit calls are serialised inside the library; bobe_gp_mll_wait may run concurrently with submits
 * and waits on OTHER slots; a slot holds one evaluation at a time; no other entry point of the same handle may
 * run while evaluations are in flight.  wait returns BOBE_OK / BOBE_NOT_PD (NaN outputs) / < 0. */
int bobe_gp_mll_submit(bobe_gp_t* gp, int slot, const double* lengthscales, double kernel_variance, int want_grad);
int bobe_gp_mll_wait(bobe_gp_t* gp, int slot, double* mll, double* grad);

/* GP.predict_mean_batched / predict_var_batched / predict_batched (gp.py:450-493) for C query points
 * Xq (C x d).  mean[c] = k_c^T alpha; var[c] = kvar + noise - |L^-1 k_c|^2 with
 *   nan_policy 0: clip(var, 1e-12) keeps NaN (predict_var_single, gp.py:465)
 *   nan_policy 1: NaN and < 1e-12 -> 1e-12     (predict_single,     gp.py:487-488)
 * either output may be NULL. */
int bobe_gp_predict(bobe_gp_t* gp, const double* Xq, int64_t C, double* mean, double* var, int nan_policy);

/* WeightedIntegratedPosteriorBase.get_next_point sweep (acquisition.py:385-398) with WIPV.fun /
 * WIPStd.fun (acquisition.py:438-440, 463-465) over GP.fantasy_var (gp.py:552-576), evaluated for
 * EVERY candidate (C x d) against the integration points Z (M x d) in the algebraically identical
 * rank-1 form.  wipv[c] = mean_z var+(z|c) * y_std^2, wipstd[c] = mean_z sqrt(var+ * y_std^2).
 * argmin_* / min_* follow jnp.argmin (first occurrence).  mean / var: as bobe_gp_predict with
 * nan_policy 1.  Any output pointer may be NULL. */
int bobe_gp_wip_sweep(bobe_gp_t* gp, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                      double* wipv, double* wipstd, double* mean, double* var, int64_t* argmin_v, double* min_v,
                      int64_t* argmin_s, double* min_s);

/* GP.fantasy_var (gp.py:552-576) for C candidates at once: out is C x M, out[c][z] = var+(z|c)*y_std^2. */
int bobe_gp_fantasy_var(bobe_gp_t* gp, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                        double* out);

/* WIPV.fun / WIPStd.fun (acquisition.py:438-440, 463-465) of C candidates TOGETHER WITH their gradients with respect
 * to the candidate coordinates - what the reference obtains with jax.grad in the local refinement of
 * get_next_point (acquisition.py:403-412).  wipv, wipstd: C (may be NULL); dwipv, dwipstd: C x d (may be NULL).
 * Integration points whose fantasy variance sits at the 1e-12 floor contribute no gradient (gradient of where). */
int bobe_gp_wip_grad(bobe_gp_t* gp, const double* cand, int64_t C, const double* Z, int64_t M, double y_std,
                     double* wipv, double* wipstd, double* dwipv, double* dwipstd);

/* EI.fun / LogEI.fun (acquisition.py:226-253, 318-330) for C points: out[c] = +EI (mode 0) or
 * +log EI (mode 1); best_y, zeta in standardised units. */
int bobe_gp_acq_ei(bobe_gp_t* gp, const double* Xq, int64_t C, double best_y, double zeta, int mode, double* out);

/* Input gradients of GP.predict_single (gp.py:476-489) — what the reference obtains by differentiating through
 * the GP with JAX (EI restarts acquisition.py:246-253/281-290; NUTS on predict_mean_batched samplers.py:268-276).
 * mean, var (may be NULL): as bobe_gp_predict with nan_policy 1.  dmean, dvar: C x d, derivatives with respect
 * to the query coordinates, standardised units; dvar is 0 where var sits at its 1e-12 floor (gradient of where).
 * var == dvar == NULL selects the mean-only mode (one small kernel, no K(X,C)): what HMC on the surrogate calls. */
int bobe_gp_predict_grad(bobe_gp_t* gp, const double* Xq, int64_t C, double* mean, double* var, double* dmean,
                         double* dvar);

/* GP.kernel(xa, xb, ls, kvar, noise, include_noise) (gp.py:124-168; call site acquisition.py:388).
 * lengthscales == NULL uses the handle's current hyper-parameters (kernel_variance / noise arguments
 * are then ignored).  The factorised state is not touched.  out is nA x nB. */
int bobe_gp_kernel(bobe_gp_t* gp, const double* A, int64_t nA, const double* B, int64_t nB,
                   const double* lengthscales, double kernel_variance, double noise, int include_noise, double* out);

/* dist_sq(x, y) (gp.py:80-96): out[a][b] = |A_a - B_b|^2 of the rows as they are; nA x nB, d = the handle's. */
int bobe_gp_dist_sq(bobe_gp_t* gp, const double* A, int64_t nA, const double* B, int64_t nB, double* out);

/* gp_mll(k, train_y, num_points) (gp.py:170-178) on a kernel matrix the CALLER assembled: Cholesky, alpha, and
 *   mll = -0.5 y^T K^-1 y - sum_i log L_ii - 0.5 n log(2 pi).
 * K: n x n symmetric, row-major; y: n.  The handle must hold no training data of another size (a data-less handle is
 * sized on the fly and stays data-less).  BOBE_NOT_PD (mll = NaN) for a pivot <= 0 only: a bare matrix has no kernel
 * variance to scale the rank test with. */
int bobe_gp_mll_from_k(bobe_gp_t* gp, const double* K, int64_t n, const double* y, double* mll);

/* fast_update_cholesky(L, k, k_self) (gp.py:181-197): v = L^-1 k (n entries: the new factor's last row) and
 * diag = sqrt(k_self - v.v) (NaN when negative, as jnp.sqrt); L: n x n lower, row-major.  Same handle rule as above. */
int bobe_gp_chol_row_update(bobe_gp_t* gp, const double* L, int64_t n, const double* k, double k_self, double* v,
                            double* diag);

/* The rank test's factor (see "Conventions"): a positive pivot below `ulp` machine epsilons of k(x,x) + noise counts as
 * not positive definite.  0 (the default) = LAPACK's sign test alone (the reference's behaviour); applies to bobe_gp_factor, the
 * bobe_gp_mll family and bobe_gp_append of this handle; copied by bobe_gp_clone_state.  get returns -1 for NULL. */
int bobe_gp_set_pivot_floor_ulp(bobe_gp_t* gp, double ulp);
double bobe_gp_get_pivot_floor_ulp(bobe_gp_t* gp);

/* Where the installed factor is ill conditioned - (kernel_variance + noise) / smallest pivot above `kappa` - the products
 * with the explicit inverse factor (v = Linv k in bobe_gp_predict, _wip_sweep, _fantasy_var, _wip_grad, _predict_grad,
 * _append) are replaced by a SOLVE with the factor, as the reference does everywhere (solve_triangular, gp.py:462, 484, 571):
 * a blocked forward substitution V_t = inv(L_tt) (B_t - sum_{j<t} L_tj V_j) whose diagonal blocks of `rows` rows are the
 * diagonal blocks of the inverse factor (bobe_gp_wip_grad's path for <= 16 candidates: one step of iterative refinement in
 * vector form instead).  The plain product loses the fantasy variance at the reference's default noise of 1e-8 from kernel
 * variances of ~1e4 on; the substitution with rows = 128 is at or below the error of LAPACK's dtrsm on every rung of
 * profiles/r06_conditioning.txt, at 1.3 x the time of the plain product (N = 4096, 65 536 candidates).
 * kappa: default BOBE_REFINE_KAPPA, else 1e6; 0 = always, negative = never.  The decision is taken when a factor is
 * installed (bobe_gp_factor, _set_chol, _append, _clone_state) - set kappa before.  get: either output may be NULL; *active
 * = 1 while the current factor's products are solved for.
 * rows: a positive multiple of 128; default BOBE_SOLVE_BLOCK, else 128 (256 / 512 are ~10 % faster and lose the fantasy
 * variance two decades of kernel variance earlier); copied by bobe_gp_clone_state.  get returns -1 for NULL. */
int bobe_gp_set_refine_kappa(bobe_gp_t* gp, double kappa);
int bobe_gp_get_refine(bobe_gp_t* gp, double* kappa, int* active);
int bobe_gp_set_solve_block(bobe_gp_t* gp, int rows);
int bobe_gp_get_solve_block(bobe_gp_t* gp);

/* GP.cholesky / GP.alphas (gp.py:259-260; state_dict keys gp.py:626-627): L is N x N lower with
 * zeros above the diagonal, alpha has N entries.  Either may be NULL. */
int bobe_gp_get_chol(bobe_gp_t* gp, double* L, double* alpha);
/* GP.from_state_dict restore without refactorisation (gp.py:671-675). */
int bobe_gp_set_chol(bobe_gp_t* gp, const double* L, const double* alpha);

/* Hamiltonian Monte Carlo on the surrogate (the consumer behind sample_GP_NUTS, samplers.py:216-360, whose NUTS calls
 * the jitted predict_mean once per leapfrog step and chain, samplers.py:268-288): L leapfrog steps of P chains in ONE
 * launch.  Target on u = logit(x): logp = (mean(x) y_std + y_mean) / temp + sum_j [log x_j + log(1 - x_j)].
 * In/out: U (P x d positions), Pm (P x d momenta: in = p0 + eps/2 grad(U), out = final momenta); inv_mass (d).
 * Out: logp (P), grad (P x d, d logp / du at the end point), mean (P, physical units), X (P x d, the end points in
 * the unit cube).  Host or device pointers (all of one kind). */
int bobe_gp_hmc_leapfrog(bobe_gp_t* gp, int64_t P, double* U, double* Pm, const double* inv_mass, double eps, int L,
                         double y_std, double y_mean, double temp, double* logp, double* grad, double* mean, double* X);

/* Whole HMC chains on the device: `niter` trajectories (momentum draw, 4-12 leapfrog steps, Metropolis test and - with
 * do_adapt - the chain's own dual-averaging step-size update) of P chains in ONE launch.  Replaces the per-step JAX
 * calls of NumPyro's NUTS in sample_GP_NUTS (BOBE/samplers.py:216-360) for a plain GP.  All pointers are HOST memory.
 *   state [P][3d+2]  in/out: u = logit(x) (d), dlogp/du (d), x (d), logp, mean (physical units)
 *   adapt [P][5]     in/out: eps, mu, hbar, log_eps_bar, m
 *   hist  [niter - hist_from][P][d]  u after the iterations >= hist_from of this call (NULL: not recorded)
 *   keep  [niter / thin][P][d+1]     x and mean after every thin-th iteration of this call (NULL: not recorded)
 *   dbg   [P][d+3]   the last iteration's momentum draw, L, uniform and acceptance probability (NULL; tests replay it)
 * Random numbers are a counter hash of (seed, chain, it0 + iteration, index).  */
int bobe_gp_hmc_run(bobe_gp_t* g, int64_t P, double* state, double* adapt, const double* inv_mass, uint64_t seed,
                    int64_t it0, int niter, int do_adapt, double y_std, double y_mean, double temp, int hist_from,
                    double* hist, int thin, double* keep, double* dbg);

/* The replacement search of nested sampling on the surrogate (the consumer behind nested_sampling_Dy, samplers.py:55-194,
 * which hands gp.predict_mean_single to dynesty's 'rwalk' sampler one point per call, samplers.py:112-115, 152): P walkers,
 * each `walks` constrained Metropolis steps x' = x + step z, z ~ N(0, I), accepted when x' lies in the unit cube and
 * mean(x') * y_std + y_mean > lstar (and, with a classifier gate set, x' is feasible) - ALL steps of all walkers in ONE
 * launch.  step: d x d lower-triangular, row-major (scale x Cholesky factor of the live points' covariance).
 *   X [P][d]  in: start points (live points), out: end points     logl [P]  in / out: physical-unit mean at the point
 *   n_accepted [P], n_inside [P]: accepted steps / proposals inside the cube (= surrogate evaluations) per walker
 *   dbg [P][d]: every walker's LAST proposal (NULL: not recorded; tests replay it).  All pointers are HOST memory (a
 *   device pointer is refused with BOBE_ERR_ARG; the same holds for bobe_gp_hmc_run).
 * Random numbers: a counter hash of (seed, walker, step, index). */
int bobe_gp_rwalk(bobe_gp_t* gp, int64_t P, double* X, double* logl, const double* step, double lstar, int walks,
                  uint64_t seed, double y_std, double y_mean, int* n_accepted, int* n_inside, double* dbg);

/* GPwithClassifier's gate (clf_gp.py:173-205) with the SVM-RBF decision function of clf.py:188-213, evaluated on the
 * device by direct differences, as the reference computes it:
 *   decision(x) = sum_i dual_coef[i] exp(-gamma |support_vectors[i] - x|^2) + intercept      (svm_predict)
 *   proba(x)    = decision >= 0 ? 1 : 0                                                      (svm_predict_proba)
 *   feasible(x) = proba >= probability_threshold                                             (clf_gp.py:179)
 * support_vectors: n_sv x d (unit-cube coordinates), dual_coef: n_sv; scikit-learn's SVC supplies them (clf.py:36-69);
 * support_vectors == NULL or n_sv == 0 clears the gate.  While a gate is set, an infeasible query point comes back as
 *   bobe_gp_predict / bobe_gp_predict_grad   mean = -INFINITY (the mark for the wrapper, which returns minus_inf in the
 *                                            units of the method at hand: clf_gp.py:179 physical, :203 standardised),
 *                                            var = 1e-12 (clf_gp.py:189, 204), dmean = dvar = 0
 *   bobe_gp_acq_ei                           EI / LogEI of (mean = minus_inf, var = 1e-12), i.e. what EI.fun computes from
 *                                            the gated predict_single (acquisition.py:246, 323)
 *   bobe_gp_hmc_leapfrog / bobe_gp_hmc_run   mean = minus_inf (physical units), no mean gradient: never accepted
 *   bobe_gp_rwalk                            mean = minus_inf: never accepted
 * bobe_gp_wip_sweep, bobe_gp_fantasy_var and bobe_gp_wip_grad are NOT gated (fantasy_var is not, clf_gp.py:207-212).
 * One summation order serves every entry point (256 lane-strided partial sums, a fixed tree), so a point near the
 * boundary falls on the same side everywhere.  The gate is not part of the state bobe_gp_clone_state copies. */
int bobe_gp_set_gate(bobe_gp_t* gp, const double* support_vectors, int64_t n_sv, const double* dual_coef, double intercept,
                     double gamma, double probability_threshold, double minus_inf);
/* decision[c] (svm_predict) and feasible[c] (1.0 / 0.0) of C query points; either output may be NULL. */
int bobe_gp_gate_eval(bobe_gp_t* gp, const double* Xq, int64_t C, double* decision, double* feasible);

/* GP.copy (gp.py:740-750) without leaving the device: dst (created with the same kernel, d and device) receives
 * src's training data, hyper-parameters and factorised state by device-to-device copies - no host round trip of the
 * N x N factor and no refactorisation (the reference copies through state_dict / from_state_dict). */
int bobe_gp_clone_state(bobe_gp_t* dst, bobe_gp_t* src);

/* GP.update at unchanged hyper-parameters as a rank-b append, O(b N^2) instead of the O(N^3) recompute_cholesky of
 * gp.py:541-550 (the reference has the rank-1 form as fast_update_cholesky, gp.py:181-197, but uses it only inside
 * fantasy_var).  X_new: b x d new points (1 <= b <= 64); y_all: the N+b targets after re-standardisation
 * (gp.py:520-536 changes all of them; L does not depend on y, alpha is re-solved).  Requires a positive-definite
 * factorised state; if the appended matrix is not positive definite the call ends like bobe_gp_factor (NaN state,
 * BOBE_NOT_PD). */
int bobe_gp_append(bobe_gp_t* gp, const double* X_new, int64_t b, const double* y_all);

/* ---- multi-GPU exchange step (the reference's counterpart is the MPI pool's pickled send/recv,
 * BOBE/pool.py:298-326, and it has no candidate parallelism at all: acquisition.py:394 maps sequentially) ----------
 * One process per GPU.  The library owns a RCCL communicator (librccl.so is opened on first use):
 *   rank 0:      bobe_mgpu_unique_id(id)  -> ship the BOBE_MGPU_ID_BYTES bytes to the other ranks (any channel)
 *   every rank:  bobe_mgpu_init(id, world, rank, device)
 * bobe_mgpu_wip_sweep = bobe_gp_wip_sweep on this rank's contiguous shard [global_offset, global_offset + C) of the
 * candidates (C may be 0), then ONE ncclAllGather of (min wipv, index, min wipstd, index, status) - 40 bytes per rank -
 * and the merge every rank repeats: smallest score, ties to the lowest GLOBAL index (jnp.argmin's first occurrence,
 * acquisition.py:397), NaN counts as minimal.  argmin_* / min_* are the global results; the score vectors stay local.
 * A rank whose local sweep fails still joins the collective (status word): EVERY rank then returns that error code
 * instead of the healthy ones waiting in the all-gather for ever.  The handle must live on bobe_mgpu_init's device.
 * bobe_mgpu_best_fit = all-gather of (mll, theta) and max by mll (pool.py:322-326) for the restart-sharded fit. */
#define BOBE_MGPU_ID_BYTES 128
int bobe_mgpu_unique_id(char* id128);
int bobe_mgpu_init(const char* id128, int world, int rank, int device);
int bobe_mgpu_world(void);
int bobe_mgpu_rank(void);
void bobe_mgpu_finalize(void);
int bobe_mgpu_wip_sweep(bobe_gp_t* gp, const double* cand_shard, int64_t C, int64_t global_offset, const double* Z,
                        int64_t M, double y_std, double* wipv, double* wipstd, double* mean, double* var,
                        int64_t* argmin_v, double* min_v, int64_t* argmin_s, double* min_s);
int bobe_mgpu_best_fit(double mll, const double* theta, int n, double* best_mll, double* best_theta);

/* number of training points / padded leading dimension currently held */
int64_t bobe_gp_npoints(bobe_gp_t* gp);

/* test / bench hooks (not part of the reference surface) -------------------------------------- */
/* C[M x N] = sum_k A(m,k) B(n,k) through the fp64 MFMA tile core.  layout 0: element (r,k) at
 * p[r*ld+k]; layout 1: at p[k*ld+r].  M, N multiples of 128, K multiple of 16.  Device or host ptrs. */
int bobe_debug_gemm(int device, int layoutA, int layoutB, int64_t M, int64_t N, int64_t K, const double* A,
                    int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc);
/* K^-1 (N x N, full symmetric) from the current factorisation — used by the parity tests */
int bobe_debug_kinv(bobe_gp_t* gp, double* Kinv);
/* L^-1 (N x N lower) from the current factorisation */
int bobe_debug_linv(bobe_gp_t* gp, double* Linv);
/* run only the Cholesky factorisation of the current K `reps` times after ONE untimed pass (first touch of the
 * workspace, clocks) and return the mean device time per factorisation in milliseconds (HIP events on the handle's
 * stream).  The two batch forms below time the same way. */
int bobe_debug_time_potrf(bobe_gp_t* gp, int reps, double* ms);
/* B factorisations in flight at once (one evaluation slot each); *ms = device time for all B together */
int bobe_debug_time_potrf_batch(bobe_gp_t* gp, int B, int reps, double* ms);
/* the same B factorisations advancing in lock step through one batched launch sequence on the handle's stream */
int bobe_debug_time_potrf_lockstep(bobe_gp_t* gp, int B, int reps, double* ms);
/* Per-kernel-class device timing with HIP events recorded on the handle's stream around every launch
 * of the selected class (0 = off).  read() synchronises, returns the summed milliseconds and the
 * number of launches since the last select()/read(), and resets the counters. */
#define BOBE_PROF_POTF2 1  /* diagonal-block Cholesky + inverse */
#define BOBE_PROF_TRSM 2   /* panel solve (GEMM with the inverted diagonal block) */
#define BOBE_PROF_SYRK 3   /* trailing update of the blocked Cholesky */
#define BOBE_PROF_TRTRI 4  /* one level of the recursive triangular inverse (two launches) */
#define BOBE_PROF_LAUUM 5  /* K^-1 = Linv^T Linv fused with the gradient reduction */
#define BOBE_PROF_TRIMUL 6 /* sweep: V = Linv * K(X, C_chunk) with column sum of squares */
#define BOBE_PROF_CROSS 7  /* sweep: WIPV / WIPStd scoring */
#define BOBE_PROF_KXX 8    /* K(X,X) assembly */
#define BOBE_PROF_CROSSVV 9 /* sweep: cross-covariance GEMM V_Z^T V of the two solved factors */
#define BOBE_PROF_KXC 10   /* sweep: K(X, C_chunk) assembly */
int bobe_gp_profile_select(bobe_gp_t* gp, int kernel_class);
int bobe_gp_profile_read(bobe_gp_t* gp, double* total_ms, int64_t* launches);
/* back-to-back v_mfma_f64_16x16x4_f64 issue rate on all CUs (1 or 2 waves per SIMD), in TFLOP/s */
int bobe_debug_mfma_peak(int device, int waves_per_simd, double* tflops);
/* The samplers' cross-lane sums (v_permlane32/16_swap + DPP, kernels_common.hpp) on one wave: in [64][D] (lane, component),
 * D = 8, 16 or 32; out[j] = sum over the lanes of component j by the all-components butterfly, out[D + j] = the same by the
 * single-value sum. */
int bobe_debug_wave_sums(int device, int D, const double* in, double* out);
/* candidate chunk size of the sweep (multiple of 128); 0 returns to the defaults (8192; 32 768 on the substitution path of an
 * ill-conditioned factor) */
int bobe_gp_set_chunk(bobe_gp_t* gp, int64_t chunk);
/* launch shape of the blocked forward substitution (bobe_gp_set_solve_block), speed only: panel = rows per long update
 * launch (a multiple of 128), chunk = candidates per launch sequence (0: the sweep's) */
int bobe_debug_solve_opts(bobe_gp_t* gp, int panel, int64_t chunk);

#ifdef __cplusplus
}
#endif
#endif
