/*
 * ddmpc.h -- C ABI of the MI355X-native batched Data-Driven MPC QP engine.
 *
 * This is the drop-in boundary for the per-timestep QP path of
 * pavelacamposp/direct_data_driven_mpc.  The reference has no FFI of its own
 * (it is pure Python on CVXPY); each entry point below names the reference
 * interface it replaces (file:line relative to the reference repository root).
 * A Python `ctypes` shim (direct_data_driven_mpc_amd/_lib.py) binds exactly
 * these symbols and re-creates the `DirectDataDrivenMPCController` class
 * surface on top of them; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns 0 (DDMPC_OK) or a negative DDMPC_ERR_* code and
 *     never throws; `ddmpc_last_error()` returns a thread-local message;
 *   - plain pointers and sizes only; all arrays are C-order fp64 unless noted;
 *   - `mem` says where the caller's buffers live: DDMPC_MEM_HOST (the library
 *     stages them through its own device buffers and the call is synchronous)
 *     or DDMPC_MEM_DEVICE (HIP device pointers, the call is asynchronous on
 *     the handle's stream);
 *   - one handle = one batch of `batch` independent controller instances that
 *     share the controller parameters and differ in their data trajectories
 *     and past windows.  A bad instance never fails a call: it gets a
 *     per-instance status code.
 *   - there is NO CPU fallback: without a HIP device every call fails with
 *     DDMPC_ERR_NO_DEVICE.
 */
#ifndef DDMPC_H
#define DDMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDMPC_ABI_VERSION 2

/* return codes */
#define DDMPC_OK                 0
#define DDMPC_ERR_INVALID       -1  /* bad argument (message says which)            */
#define DDMPC_ERR_UNSUPPORTED   -2  /* valid request the HIP path does not cover     */
#define DDMPC_ERR_NO_DEVICE     -3  /* no usable HIP device                          */
#define DDMPC_ERR_HIP           -4  /* a HIP runtime call failed                     */
#define DDMPC_ERR_NOT_READY     -5  /* e.g. solve before set_data                    */

/* DataDrivenMPCType, direct_data_driven_mpc_controller.py:11-14 */
#define DDMPC_NOMINAL 0
#define DDMPC_ROBUST  1
/* SlackVarConstraintTypes, direct_data_driven_mpc_controller.py:16-20 */
#define DDMPC_SLACK_NON_CONVEX 0    /* rejected, controller.py:664-670 */
#define DDMPC_SLACK_CONVEX     1
#define DDMPC_SLACK_NONE       2

/* per-instance solve status; maps 1:1 onto the CVXPY strings the reference
 * tests for in get_optimal_control_input (controller.py:804-808) */
#define DDMPC_STATUS_OPTIMAL            0   /* "optimal"            */
#define DDMPC_STATUS_OPTIMAL_INACCURATE 1   /* "optimal_inaccurate" */
#define DDMPC_STATUS_INFEASIBLE         2   /* "infeasible"         */
#define DDMPC_STATUS_UNBOUNDED          3   /* "unbounded"          */
#define DDMPC_STATUS_SOLVER_ERROR       4   /* "solver_error"       */

#define DDMPC_WEIGHT_SCALAR 0       /* Q = q*I, R = r*I (what the reference loader builds,
                                       utilities/controller/controller_creation.py:125-127) */
#define DDMPC_WEIGHT_DIAG   1       /* Q = diag(q[0..p*L)), R = diag(r[0..m*L)); entries >= 0 (a zero weight leaves the
                                       component free) */
#define DDMPC_WEIGHT_DENSE  2       /* Q [p*L, p*L], R [m*L, m*L] row-major, symmetric; on the free prediction steps positive
                                       definite once all-zero rows / columns (components without a weight, as zeros on the
                                       diagonal of a DIAG matrix) are set aside (controller.py:121-124,708-710); every slack
                                       mode.  A null space that is not spanned by coordinate axes: DDMPC_ERR_UNSUPPORTED */

#define DDMPC_MEM_HOST   0
#define DDMPC_MEM_DEVICE 1

#define DDMPC_GRAM_AUTO       0     /* = STRUCTURED                                      */
#define DDMPC_GRAM_DENSE      1     /* H H' by fp64 MFMA over the implicit Hankel operand */
#define DDMPC_GRAM_STRUCTURED 2     /* Hankel sliding-window recurrence (hankel_matrix.py:5-53 is generic in the channel count
                                       and so is this: inside the cold-solve kernel for m + p == 2 or 4, by a launch ahead of
                                       it for any other count)                              */

/* ddmpc_get_solution selectors: the `.value` of the reference's cp.Variables
 * (controller.py:434-445) */
#define DDMPC_SOL_ALPHA 0           /* [batch, N-L-n+1]   */
#define DDMPC_SOL_UBAR  1           /* [batch, (L+n)*m]   */
#define DDMPC_SOL_YBAR  2           /* [batch, (L+n)*p]   */
#define DDMPC_SOL_SIGMA 3           /* [batch, (L+n)*p], robust only */

/* ddmpc_set_option */
#define DDMPC_OPT_CLOSED_LOOP_PATH 1
#define DDMPC_PATH_AUTO 0           /* warm: fused loop without inequality; per-step warm + cold re-solves with the slack box */
#define DDMPC_PATH_COLD 1           /* a full cold solve per control step                         */
#define DDMPC_PATH_WARM 2
#define DDMPC_OPT_CLOSED_LOOP_GRAPH 2 /* 1: record the per-step launches of ddmpc_closed_loop into a HIP graph and replay it
                                         (default 0: measured slower than plain asynchronous launches) */

#define DDMPC_OPT_REFINE 3            /* iterative refinement of the cold solve with exact Hankel products (residual
                                         t - (H(H'beta) + lam D beta) from the trajectory, correction through the factor at hand):
                                         0 off, 1 auto (default: every solve is checked with that residual, relative to |t|,
                                         and only the instances above the threshold are solved again with refinement),
                                         2 always.  Passes repeat until the correction is at rounding level or stops shrinking.
                                         ddmpc_prepare forms the affine law of ddmpc_step from refining solves as well
                                         (n(m+p)+1 launches, once per data set): with 2 for every instance, with 1 for the
                                         instances it flags; changing the mode invalidates the law. */
#define DDMPC_REFINE_OFF 0
#define DDMPC_REFINE_AUTO 1
#define DDMPC_REFINE_ALWAYS 2
#define DDMPC_OPT_REFINE_MAX 4        /* cap on refinement passes per factorisation (default 3) */
#define DDMPC_OPT_REFINE_RES_LOG10 5  /* auto mode: refine when |t - (H(H'beta) + lam D beta)|_inf / |t|_inf exceeds
                                         10^(-value/10), value in tenths of a decade below 1 (default DDMPC_REFINE_RES_DEFAULT;
                                         3000 refines everything, 0 only what comes out non-finite) */
#define DDMPC_OPT_LARGE_PIPELINE 6     /* NOMINAL controllers beyond the register-resident kernels ((m+p)(L+n) > 271): how the
                                         data-dependent half of a solve (Gram, rank-revealing Cholesky, C'WC and its factor;
                                         controller.py:506-538) is executed -- results agree to rounding */
#define DDMPC_PIPELINE_ONE_WORKGROUP 0 /* one workgroup per instance runs every phase (ddmpc_nominal_rr_kernel<1>) */
#define DDMPC_PIPELINE_PHASES 1        /* default: one kernel per phase over the whole batch in lock step, several workgroups
                                         per instance, Cholesky by 64-column panels (ddmpc_rr2.hpp) */
#define DDMPC_OPT_LARGE_AFFINE_LAW 7   /* NOMINAL controllers beyond the register-resident kernels, phase-kernel pipeline: 1 = ddmpc_prepare also
                                         forms the affine control law z(past) (solves at the zero window and the n(m+p) unit windows,
                                         once per data set) and ddmpc_step evaluates it in one HBM-bound launch; ddmpc_get_gain then
                                         returns it as [batch][n(m+p)+1][(m+p)(L+n)], z = [ubar; ybar] per component = gain[0] +
                                         gain[1:]' [u_past; y_past].  0 (default): ddmpc_step repeats the solve on the kept factors,
                                         bit-equal to ddmpc_solve (controller.py:389-407).  Scalar / diagonal weights, at most 1024
                                         rows (DDMPC_ERR_UNSUPPORTED otherwise) */
#define DDMPC_OPT_CONVEX_UPDATE 8      /* ROBUST controllers with the CONVEX slack box on the register-resident kernels
                                         (controller.py:631-677): 1 (default) = active-set iterations after the first keep the factor
                                         of the empty active set and treat the <= 4 switched slack components as a diagonal modification
                                         of rank k (Woodbury: one block forward substitution on the matrix pipe, a k x k system, one
                                         back substitution); more components, or anything unusual, falls back to 0 = the whole system is
                                         formed and factored again in every iteration (rounds 1-4).  Same active sets, iteration
                                         counts and -- to rounding -- solutions */
#define DDMPC_OPT_GRAM_LAUNCH 9        /* register-resident kernels, structured Gram of plants with other than two or four channels
                                         (hankel_matrix.py:5-53): which launch ahead of the solve forms the Gram tiles.  0 = the
                                         streaming matrix-pipe kernel (trajectory in chunks, several lags per tile for at most eight
                                         channels; the default from six channels on, and what trajectories beyond the LDS take), 1 = the
                                         launch of round 4 that stages the whole trajectory and walks on the vector units (the default
                                         up to five channels).  Same tiles to rounding; DDMPC_ERR_UNSUPPORTED when the requested launch
                                         cannot hold the shape */
#define DDMPC_REFINE_RES_DEFAULT 107  /* 2e-11: benchmark data stays below ~2e-12, the parity bars are missed from ~1.3e-10 on */

typedef struct ddmpc_handle ddmpc_handle;

/* Controller parameters = the constructor arguments of
 * DirectDataDrivenMPCController.__init__ (controller.py:95-116) minus the
 * per-instance data.  All pointers are HOST pointers, copied at create. */
typedef struct ddmpc_params {
  int32_t struct_size;              /* sizeof(ddmpc_params), ABI guard              */
  int32_t m, p, n, L, N;            /* inputs, outputs, est. order, horizon, data length */
  int32_t controller_type;          /* DDMPC_NOMINAL | DDMPC_ROBUST                  */
  int32_t slack_type;               /* DDMPC_SLACK_*                                 */
  int32_t use_terminal_constraint;  /* controller.py:229,489-492                     */
  int32_t weight_kind;              /* DDMPC_WEIGHT_*                                */
  const double* Q;                  /* 1 value (scalar), p*L values (diag) or (p*L)^2 (dense) */
  const double* R;                  /* 1 value (scalar), m*L values (diag) or (m*L)^2 (dense) */
  double eps_max, lamb_alpha, lamb_sigma, c;   /* robust parameters, controller.py:197-205 */
  const double* u_s;                /* [m]                                           */
  const double* y_s;                /* [p]                                           */
  int32_t max_iter;                 /* active-set iteration cap for slack CONVEX (0 -> 50) */
  int32_t gram_mode;                /* DDMPC_GRAM_*                                  */
} ddmpc_params;

int ddmpc_version(void);
const char* ddmpc_last_error(void);

/* Number of visible HIP devices (0 if none); never fails. */
int ddmpc_device_count(void);

/* Problem sizes: (m+p)(L+n) <= 271 rows run on the register-resident cold-solve kernels (all schemes, all weight
 * kinds).  Beyond that, controllers run on single-workgroup kernels that keep their
 * matrices in a global workspace (a Gram-route solve plus refinement with exact Hankel products; throughput: see
 * profiles/README.md): ROBUST ones on ddmpc_large_solve_kernel (same outputs, status, iterations, ddmpc_get_solution),
 * NOMINAL ones on the rank-revealing route -- phase kernels over the whole batch by default, the one-workgroup kernels on
 * request (DDMPC_OPT_LARGE_PIPELINE) -- with ddmpc_get_solution: ubar / ybar from its z, alpha = H'x from the vector it
 * exports.  By default no affine law is formed at that size (ddmpc_get_gain is DDMPC_ERR_UNSUPPORTED unless a NOMINAL
 * controller asked for it with DDMPC_OPT_LARGE_AFFINE_LAW); the warm path there is
 * factor reuse -- ddmpc_prepare forms what depends on the data and the weights alone (NOMINAL: Gram, its rank-revealing
 * factor, the reduced normal matrix and its factor; ROBUST: Gram + lam D, the factor of the columns outside the slack
 * box, the Schur complement of the boxed block), ddmpc_step and the per-step closed loop solve on what it kept, with
 * results bit-equal to ddmpc_solve's.  Both schemes run on phase kernels over the whole batch by default up to 1024 rows
 * (round 5: also ROBUST controllers, and dense weighting matrices of either scheme; NOMINAL + dense: phase kernels only);
 * ROBUST controllers of 1025 .. 2048 rows and NOMINAL ones of 1025 .. 1524 (diagonal weights) run on 1024-thread instances of
 * the one-workgroup kernels.  Anything larger is DDMPC_ERR_UNSUPPORTED (reported by ddmpc_create).
 *
 * Replaces DirectDataDrivenMPCController.__init__ parameter validation
 * (controller.py:165-168,211-222,298-343,664-670) for a batch of instances on
 * HIP device `device`.  Does not solve. */
int ddmpc_create(const ddmpc_params* params, int64_t batch, int device, ddmpc_handle** out);
int ddmpc_destroy(ddmpc_handle* h);

/* Use an existing hipStream_t (passed as void*) instead of the handle's own. */
int ddmpc_set_stream(ddmpc_handle* h, void* hip_stream);
int ddmpc_synchronize(ddmpc_handle* h);

/* Per-instance data trajectories u_d [batch,N,m], y_d [batch,N,p]
 * (controller.py:177-179).  HOST: copied.  DEVICE: borrowed -- the buffers must stay valid for as long as the handle
 * uses them.  Their CONTENTS may be rewritten in place between calls: ddmpc_solve / ddmpc_closed_loop with the cold
 * path read the trajectories anew in every call and keep nothing derived from them.  What IS derived from the data and
 * kept is the affine law of the warm path (ddmpc_prepare, ddmpc_step, ddmpc_get_gain, the warm closed loop): after
 * rewriting borrowed data call ddmpc_set_data again (it only re-registers the pointers and drops the law). */
int ddmpc_set_data(ddmpc_handle* h, const double* u_d, const double* y_d, int mem);

/* One control step for every instance = update_and_solve_data_driven_mpc +
 * get_optimal_control_input (+ get_optimal_cost_value / get_problem_solve_status),
 * controller.py:389-407,739-808.  "Cold" solve: implicit Hankel -> Gram ->
 * reduced KKT -> Cholesky -> solve (-> active-set iterations for slack CONVEX).
 *   u_past [batch, n*m], y_past [batch, n*p]   (controller.py:184-185,577-581)
 *   u_opt  [batch, L*m]  = ubar[n*m:]          (controller.py:799-805)
 *   cost   [batch]       = problem.value       (controller.py:778)
 *   status [batch] int32 DDMPC_STATUS_*        (controller.py:755)
 *   iters  [batch] int32 factorisations used (may be NULL)
 * NOMINAL controllers: an instance whose Gram matrix is singular (noise-free data) is re-solved in the same
 * call by a rank-revealing kernel; it comes back "optimal", or "infeasible" when the hard constraints cannot be
 * met by any trajectory in the range of the Hankel matrix. */
int ddmpc_solve(ddmpc_handle* h, const double* u_past, const double* y_past,
                double* u_opt, double* cost, int32_t* status, int32_t* iters, int mem);

/* ddmpc_set_data + ddmpc_solve for HOST buffers in one call, pipelined: the batch is cut into chunks of
 * instances, chunk k+1 is uploaded on a copy stream while chunk k is solved on the compute stream.  Same
 * outputs as ddmpc_solve; afterwards the handle holds the uploaded data (ddmpc_step / ddmpc_get_solution
 * work as after ddmpc_set_data + ddmpc_solve).  All pointers are host pointers. */
int ddmpc_solve_from_host(ddmpc_handle* h, const double* u_d, const double* y_d,
                          const double* u_past, const double* y_past,
                          double* u_opt, double* cost, int32_t* status, int32_t* iters);

/* Warm path = what the reference's per-step entry point could reuse but does not:
 * update_and_solve_data_driven_mpc (controller.py:389-407) rebuilds and re-solves the whole QP
 * although only u_past / y_past changed (:404-407, :577-581); the Hankel matrices, weights and
 * therefore the KKT matrix are step-invariant (:376-377).  For a nominal controller and for a
 * robust one with slack NONE the QP has no inequality, so the solution is an affine function of
 * [u_past; y_past].
 *
 * ddmpc_prepare: one cold factorisation per instance with the Cholesky factor exported, then
 *   nf = n*(m+p) triangular solves per instance give the affine law
 *   beta = gain[:,0] + gain[:,1:] [u_past; y_past].  Invalidated by ddmpc_set_data /
 *   ddmpc_set_setpoints.  With slack CONVEX the law is that of the EMPTY active set (no sigma at its
 *   bound), i.e. the first primal-dual active-set iterate.
 * ddmpc_step: same contract and outputs as ddmpc_solve; uses the affine law (preparing on first
 *   use).  With slack CONVEX an instance whose affine iterate keeps every boxed sigma inside
 *   |sigma| <= c*eps_max is optimal as it is (iters = 1); the others are re-solved by the cold kernel
 *   in the same call (full active-set iteration, iters >= 2), so the results equal ddmpc_solve's.
 *   The status of a warm step is the status of the factorisation it rests on.
 * ddmpc_get_gain: out [batch, nf+1, r] doubles, r = (m+p)(L+n) components in the internal
 *   time-major order rho = k*(m+p) + ch (ch < m: ubar, else ybar+sigma).
 * Beyond 271 rows: see the note on problem sizes at ddmpc_create (the data-dependent factors are kept, no law). */
int ddmpc_prepare(ddmpc_handle* h);
int ddmpc_step(ddmpc_handle* h, const double* u_past, const double* y_past,
               double* u_opt, double* cost, int32_t* status, int32_t* iters, int mem);
int ddmpc_get_gain(ddmpc_handle* h, double* out, int mem);

/* Engine options (DDMPC_OPT_*). */
int ddmpc_set_option(ddmpc_handle* h, int option, int value);

/* set_input_output_setpoints (controller.py:945-982); takes effect at the next solve. */
int ddmpc_set_setpoints(ddmpc_handle* h, const double* u_s, const double* y_s);

/* Values of the optimisation variables after the last ddmpc_solve
 * (controller.py:434-445 `.value`); `out` sized as listed at DDMPC_SOL_*.  Instances of a NOMINAL controller that
 * were solved by the rank-revealing rescue kernel (exact, rank-deficient data) report ubar / ybar from that kernel's
 * own solution and NaN for alpha (any alpha with H alpha = [ubar; ybar] is optimal there; none is formed). */
int ddmpc_get_solution(ddmpc_handle* h, int what, double* out, int mem);

/* hankel_matrix(X, L) for a batch (direct_data_driven_mpc/utilities/hankel_matrix.py:5-53):
 * X [batch,N,nch] -> H [batch, L*nch, N-L+1].  Stand-alone (no handle). */
int ddmpc_hankel(const double* X, int64_t batch, int32_t N, int32_t nch, int32_t L,
                 double* H, int mem, int device);

/* Batched persistent-excitation guard = the construction-time check of the reference
 * (controller.py:275-296 -> evaluate_persistent_excitation, hankel_matrix.py:55-87: rank of the
 * order-`order` Hankel matrix of u_d equals m*order, by SVD).  u_d [batch,N,m]; for every instance
 * ratio_lb[b] receives a rigorous lower bound of sigma_min/sigma_max of that Hankel matrix, from a
 * Cholesky factorisation of its Gram matrix (0 when the factorisation breaks down).  A bound well
 * above the SVD tolerance max(M,N)*eps certifies full rank on the device; the caller runs the exact
 * SVD test only on the instances it leaves undecided (see BatchedDDMPC.persistent_excitation_guard).
 * Stand-alone (no handle).  The packed Gram matrix lives in LDS when it fits (m*order <= ~190 rows), else in
 * a temporary global workspace. */
int ddmpc_pe_guard(const double* u_d, int64_t batch, int32_t N, int32_t m, int32_t order,
                   double* ratio_lb, int mem, int device);

/* LTI plant x+ = A x + B u, y = C x + D u + w (utilities/model_simulation.py:70-98); row-major HOST pointers. */
typedef struct ddmpc_plant {
  int32_t ns;                       /* state dimension                              */
  const double* A;                  /* [ns,ns]                                      */
  const double* B;                  /* [ns,m]                                       */
  const double* C;                  /* [p,ns]                                       */
  const double* D;                  /* [p,m]                                        */
} ddmpc_plant;

/* Batched closed loop, entirely on the device: the driver loop of
 * utilities/controller/controller_operation.py:259-305 (Algorithm 1, and the n-step
 * Algorithm 2 when n_mpc_step > 1) for every instance of the batch:
 *   for t = 0, n_mpc_step, 2 n_mpc_step, ... < n_steps:
 *     QP solve with the current past windows                   (controller.py:389-407)
 *       -- by the affine law of ddmpc_prepare when there is no inequality (the whole loop of an
 *          instance then runs inside one workgroup), else a cold solve per step; see
 *          DDMPC_OPT_CLOSED_LOOP_PATH
 *     for k = t .. min(t + n_mpc_step, n_steps) - 1:
 *       u[k] = optimal_u[(k-t) m : (k-t+1) m]                  (controller.py:839)
 *       y[k] = C x + D u[k] + w[k];  x <- A x + B u[k]         (model_simulation.py:93-98)
 *       FIFO push of (u[k], y[k]) into the past windows         (controller.py:893-895)
 * x [batch,ns], u_past [batch,n*m], y_past [batch,n*p] are updated in place;
 * w [batch,n_steps,p] is the measurement noise; u_sys [batch,n_steps,m], y_sys [batch,n_steps,p]
 * receive the closed-loop trajectories; status [batch] the worst solve status seen (an instance whose
 * solve is not optimal stops evolving -- the reference raises at that point, controller.py:808 --
 * and the rest of its trajectory is NaN).  All buffers follow `mem`. */
int ddmpc_closed_loop(ddmpc_handle* h, const ddmpc_plant* plant, int32_t n_steps, int32_t n_mpc_step,
                      double* x, double* u_past, double* y_past, const double* w,
                      double* u_sys, double* y_sys, int32_t* status, int mem);

/* Kernel facts for benchmarking: algorithmic flops and bytes of one cold solve
 * with this handle's configuration (see DESIGN.md "Roofline accounting"). */
int ddmpc_cost_model(ddmpc_handle* h, double* flops_per_solve, double* bytes_per_solve);

/* Name of the dominant kernel as it appears in rocprofv3 traces. */
const char* ddmpc_kernel_name(ddmpc_handle* h);

/* Diagnostics only: in-kernel phase stamps (shader-clock ticks) of the next solves.
 * `enable` != 0 turns stamping on (zeroing the buffer); `out` (host, [batch,16] uint64,
 * may be NULL) receives the stamps of the last solve: [0] kernel entry, [1..6] phase
 * ends (staging + tables, lag blocks, base tiles + diagonal walks, Cholesky, [5] = [4], back substitution);
 * [7..11] Cholesky sub-phase sums of wave 0, [14] exit,
 * [15]/[13] the 100 MHz real-time counter at entry/exit.  Off by default; a stamping
 * run must not be used for timing claims. */
int ddmpc_debug_stamps(ddmpc_handle* h, int enable, uint64_t* out);

/* Diagnostics only (problems beyond the register-resident kernels): copies instance `b`'s slice of the global workspace
 * (packed factor of the Gram matrix, then the packed factor of the reduced normal matrix; rows on 128-byte boundaries)
 * and its pivot record [skip (rv) | skipT (rv) | nlive | nRl] to host memory, at most `ws_count` doubles / `meta_count`
 * ints; either output may be NULL.  *ws_avail / *meta_avail (may be NULL) receive the sizes of the slices.
 * Round 5: ws_count < 0 (NOMINAL, phase kernels): the pivot candidates of G's factorisation in the order they were decided on
 * (16 ceil(r / 16) doubles: the pivot where a column was accepted, the residue where it was skipped; tools/pivot_gap_study.py).
 * ROBUST controllers beyond 271 rows on the phase kernels: meta_out receives the per-instance record of the last solve
 * [k, state, iterations, k, switched positions (64), active set (r rounded up to 2), start tick, ticks (100 MHz), -, -] and
 * *ws_avail = 0 (tools/rr3_schedule.py). */
int ddmpc_debug_workspace(ddmpc_handle* h, int64_t b, double* ws_out, int64_t ws_count, int32_t* meta_out, int64_t meta_count,
                          int64_t* ws_avail, int64_t* meta_avail);

/* Diagnostics only, process-wide: from now on every device buffer the library allocates is pre-filled with `byte` (0: off, the
 * default; 255 gives NaN bit patterns, 63 small finite doubles).  A result that changes with the fill has read memory no kernel
 * wrote -- fresh device memory is zero on this stack, which hides such reads in a standalone run.  Returns the previous setting.
 * (tests/test_gpu_round4.py) */
int ddmpc_debug_poison_allocations(int byte);

#ifdef __cplusplus
}
#endif
#endif /* DDMPC_H */
