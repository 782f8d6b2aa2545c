/* symode.h -- C ABI of libsymode_hip.so: the MI355X (gfx950) SINDy hot path.
 *
 * The reference (Rose-STL-Lab/symmetry-ode-discovery) has no FFI boundary: the hot path is
 * Python calling torch ops.  Each entry point below replaces the torch-op sequence named in
 * its comment (file:line of the reference) and is what a binding for that sequence would
 * call (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (fp32 unless stated), row-major,
 *     contiguous; nothing is allocated or freed inside the library;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously on it, no call synchronises;
 *   - return value: 0 = ok, < 0 = SYMODE_E_* argument error (nothing was launched),
 *     > 0 = a hipError_t reported by the launch;
 *   - `flags`: bit 0 = include_sine, bit 1 = include_exp (reference sindy.py:74-77);
 *   - library column order: [1 | x_i | x_i x_j (i<=j) | (x_i x_j) x_k (i<=j<=k) | ... |
 *     sin x_i | exp x_i]  (reference sindy.py:7-30, 68-77); orders 4-5 continue the nesting.
 *   - batched entry points take `n_problems` independent (trajectory, seed) problems laid
 *     out back to back: x[s] = x + s*n*d, xi[s] = xi + s*d*p, ...
 */
#ifndef SYMODE_H
#define SYMODE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYMODE_FLAG_SINE 1
#define SYMODE_FLAG_EXP 2

#define SYMODE_OK 0
#define SYMODE_E_UNSUPPORTED (-1) /* (d, order, flags) outside the compiled library set */
#define SYMODE_E_NULLPTR (-2)
#define SYMODE_E_BADSIZE (-3)
#define SYMODE_E_WORKSPACE (-4) /* workspace missing or too small */
#define SYMODE_E_ALIGN (-5)     /* pointer not 4-byte (fp32) / 8-byte (fp64) aligned */

/* ABI version of this header (bumped on any signature change). */
int symode_abi_version(void);

/* The optional SYMODE_* tuning / A-B variables are read once, when the library is first used; this
 * reads them again (tests and tuning tools that compare two settings inside one process).  No reference counterpart. */
void symode_reload_env(void);
const char* symode_error_string(int code);

/* p = number of library columns; SYMODE_E_UNSUPPORTED if (d, order, flags) is not compiled in.
 * replaces: SINDyRegression.get_term_num, sindy.py:179-189 */
int symode_lib_size(int d, int order, int flags);

/* Bytes of scratch the batched reductions need for (n_problems, n) at this library.
 * The same buffer serves symode_loss_grad / symode_aug_gram / the symreg entry points. */
size_t symode_workspace_bytes(int d, int order, int flags, long n_problems, long n);

/* One-time preparation of a freshly allocated workspace (enqueued on `stream`): writes the header the
 * one-launch reductions keep their "last workgroup done" tickets in.  The tickets reset themselves, so one
 * call per allocation is enough; a workspace that never saw this call makes the reductions return NaN
 * (never stale numbers).  A workspace must not be shared by calls that run concurrently on different streams.
 * replaces: nothing in the reference (its reductions are torch ops); it exists so that the closure body
 * train.py:663-664 + 689 is ONE kernel launch at the 50x2500x2 shape. */
int symode_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/* Theta(x) materialised: theta_out (n, p).
 * replaces: SINDyRegression.eval_Theta_at, sindy.py:201-203 (term functions sindy.py:7-30) */
int symode_theta(const float* x, long n, int d, int order, int flags, float* theta_out, void* stream);

/* out (n, d) = Theta(x) @ (xi * mask)^T ; mask may be NULL (all ones). xi, mask: (d, p).
 * replaces: SINDyRegression.forward, sindy.py:79-82 */
int symode_forward(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask,
                   float* out, void* stream);

/* Fused Theta-build + residual + MSE + gradient, never materialising Theta:
 *   r = Theta(x) (xi*mask)^T - dx ;  loss[s] = inv_count * sum r^2 ;
 *   grad[s] (d, p) = 2 * inv_count * (r^T Theta) * mask.
 * inv_count = 1/(n*d) gives torch's MSELoss mean; a rank holding a shard passes
 * 1/(n_global*d) and all-reduces (SUM) loss and grad.
 * replaces: closure body train.py:663-664 + backward train.py:689 (and 789-790, 802). */
int symode_loss_grad(const float* x, const float* dx, long n_problems, long n, int d, int order, int flags,
                     const float* xi, const float* mask, float inv_count, float* loss_out, float* grad_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* K explicit Euler steps x <- x + dt * Theta(x)(xi*mask)^T, out (n, d).
 * replaces: odeint(regressor, x, t, dt, 'euler'), model_utils.py:233-238 (n_steps = int(t/dt)
 * is computed by the caller). method: 0 = euler, 1 = rk4 (model_utils.py:241-247). */
int symode_odeint(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask,
                  int n_steps, float dt, int method, float* out, void* stream);

/* Same integration, every intermediate state kept: traj (n_steps, n, d), traj[s] = state after step s+1.
 * replaces: odeint(regressor, x0, t, dt, method, full_traj=True), model_utils.py:249-254 -- the long-term
 * prediction roll-out of evaluation/eval_ltp.py:31-37 (n_steps sequential Theta builds + GEMMs there). */
int symode_odeint_traj(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask,
                       int n_steps, float dt, int method, float* traj, void* stream);

/* Augmented Gram matrix in fp64 (MFMA f64): A = [Theta(x) | dx] (n, p+d),
 *   gram_out[s] (p+d, p+d) = A^T A   (row-major, both triangles filled).
 * Every quantity of the ridge-augmented least-squares solve (sindy.py:261-288) and of the
 * MSE closure is a function of this matrix; x is fixed for a whole run (train.py:626).
 * replaces: the normal-equation content of solve_SINDy_one_step, sindy.py:250-315. */
int symode_aug_gram(const float* x, const float* dx, long n_problems, long n, int d, int order, int flags,
                    double* gram_out, void* workspace, size_t workspace_bytes, void* stream);

/* Same, for n_problems index subsets of ONE shared data set: problem s uses the m points
 * x[idx[s*m + i]], i < m (int32 row indices < n_src; the caller guarantees the range).
 * replaces: the per-seed DataLoader subsample (main.py:36-38) of a seed sweep
 * (the run_scripts seed loops, `for i in {0..49}`) followed by sindy.py:261-288, for all seeds at once. */
int symode_aug_gram_gather(const float* x, const float* dx, long n_src, const int* idx, long n_problems, long m, int d,
                           int order, int flags, double* gram_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Weak SINDy: the contraction of the test functions with the library, fused with the library build (fp64 MFMA, Theta
 * never written).  x (n_t, d): ONE trajectory on a uniform time grid; V, V_drv (n_test, n_t) fp32 row-major: the test
 * functions and their derivatives times dt (sindy.py:346-347).  out: fp64 row-major (R, C) with R = 16*ceil(2 n_test/16),
 * C = 16*ceil((p+d)/16):
 *     out[k, j]            = sum_t V[k, t] Theta_j(x_t)          k < n_test, j < p        (= G, sindy.py:363)
 *     out[n_test + k, p+i] = -sum_t V_drv[k, t] x_i(t)           k < n_test, i < d        (= b, sindy.py:364)
 * (the other entries are by-products of the 16x16 tiles).  n_test <= 128.
 * replaces: self.V @ self.regressor.eval_Theta_at(x) and -self.V_drv @ x, sindy.py:362-364. */
int symode_weak_gram(const float* x, long n_t, int d, int order, int flags, const float* V, const float* V_drv, int n_test,
                     double* out, void* workspace, size_t workspace_bytes, void* stream);

/* S1, linear-latent symmetry regulariser (train.py:502-507 with the intended [1]):
 *   loss = sum_v sum_n || Xi_m J_Theta(z_n)(L_v z_n) - L_v Xi_m Theta(z_n) ||^2,
 *   grad (d, p) = dloss/dxi (masked).  L: (n_gen, d, d). */
int symode_symreg_linear(const float* z, long n, int d, int order, int flags, const float* xi, const float* mask,
                         const float* L, int n_gen, float* loss_out, float* grad_out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* S4, reversed symmetry regulariser with (g(x), J_g(x)) precomputed once
 * (model_utils.py:126-170; g and J_g do not depend on xi, model_utils.py:172-211):
 *   loss = sum_g mean_{n,j} ( J_g(x_n) h(x_n) - h(g(x_n)) )^2,  h = Theta(.)(xi*mask)^T,
 *   grad (d, p) = dloss/dxi (masked).  gx: (n_g, n, d), jgx: (n_g, n, d, d). */
int symode_symreg_reversed(const float* x, const float* gx, const float* jgx, int n_g, long n, int d, int order,
                           int flags, const float* xi, const float* mask, float* loss_out, float* grad_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The same regulariser for n_problems independent (trajectory, seed) problems in ONE launch:
 * x (S, n, d), gx (S, n_g, n, d), jgx (S, n_g, n, d, d), xi / mask (S, d, p); loss (S), grad (S, d, p).
 * inv_count as in symode_loss_grad (1/(n*d) for the plain mean; a rank holding a point shard passes 1/(n_global*d)
 * and all-reduces).  replaces: the per-seed processes of run_scripts/lv_noise99_eq_rreg.sh, each evaluating
 * model_utils.py:160-168 per closure. */
int symode_symreg_reversed_batched(const float* x, const float* gx, const float* jgx, int n_g, long n_problems, long n, int d,
                                   int order, int flags, const float* xi, const float* mask, float inv_count,
                                   float* loss_out, float* grad_out, void* workspace, size_t workspace_bytes, void* stream);

/* The whole closure of the reversed-regulariser fit in ONE pass over the points: the residual shares Theta(x) and
 * h(x) with the regulariser and x is read once (40 instead of 16 + 32 bytes per point at d = 2, n_g = 1):
 *     loss2_out[s] = inv_count * ( sum r^2 , sum_g sum u^2 ),   r = h(x) - dx,  u = J_g(x) h(x) - h(g(x)),
 *     grad_out[s]  = d( loss2[0] + w_sym * loss2[1] ) / dxi   (masked).
 * Layouts as in symode_symreg_reversed_batched; n_g >= 1.
 * replaces: train.py:663-664 + 675-679 + 689 with sym_reg_type 'r' (loss_sindy_x + w_sym_reg * symm_loss, backward). */
int symode_loss_grad_reversed(const float* x, const float* dx, const float* gx, const float* jgx, int n_g, long n_problems, long n,
                              int d, int order, int flags, const float* xi, const float* mask, float inv_count, float w_sym,
                              float* loss2_out, float* grad_out, void* workspace, size_t workspace_bytes, void* stream);

/* Reverse mode of symode_forward, given g = dL/d(out) (n, d):
 *   grad_x (n, d) = J_Theta(x)^T (xi*mask)^T g   (skipped when grad_x is NULL),
 *   grad_xi (d, p) = (g^T Theta(x)) * mask.
 * replaces: autograd's backward through sindy.py:79-82 (cat-of-products + matmul). */
int symode_vjp(const float* x, const float* g, long n, int d, int order, int flags, const float* xi, const float* mask,
               float* grad_x, float* grad_xi, void* workspace, size_t workspace_bytes, void* stream);

/* Forward mode: out (n, d) = Theta(x)(xi*mask)^T (skipped when out is NULL) and
 *   jv (n, d) = (J_Theta(x) v)(xi*mask)^T  for tangents v (n, d).
 * replaces: torch.autograd.functional.jvp(regressor, x, v)[1], the double-backward trick of
 * model_utils.py:56, 166 and train.py:505. */
int symode_forward_jvp(const float* x, const float* v, long n, int d, int order, int flags, const float* xi,
                       const float* mask, float* out, float* jv, void* stream);

/* Reverse mode of symode_forward_jvp, given g_out = dL/d(out) (may be NULL = 0) and g_jv = dL/d(jv):
 *   grad_x (n, d) (includes the second-order term through J_Theta(x) v), grad_v (n, d),
 *   grad_xi (d, p) = (g_out^T Theta + g_jv^T (J_Theta v)) * mask.
 * replaces: the backward of the create_graph=True jvp in model_utils.py:32, 56 (what makes the
 * reference need a twice-differentiable regressor). */
int symode_jvp_vjp(const float* x, const float* v, const float* g_out, const float* g_jv, long n, int d, int order,
                   int flags, const float* xi, const float* mask, float* grad_x, float* grad_v, float* grad_xi,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Offline data generation: n_traj RK4 orbits of dx/dt = Theta(x) xi^T (xi (d, p) fp64, unmasked) from x0
 * (n_traj, d) fp64, n_steps of size dt; every `subsample`-th state and its exact derivative are written as
 * fp32 x_out, dx_out (n_traj, ceil(n_steps/subsample), d).  One trajectory per thread, fp64 arithmetic.
 * replaces: solve_ode_batch + the subsample/transposes of gen_data, data_utils/ode.py:7-28, 45-48, for systems
 * the library can express (all four shipped ones: evaluation/eval_eq.py:88-105). */
int symode_rk4_traj(const double* x0, long n_traj, int d, int order, int flags, const double* xi, int n_steps, double dt,
                    int subsample, float* x_out, float* dx_out, void* stream);

/* Seed sweeps: idx_out (n_seeds, m) int32, rows ascending -- for every seed the m-subset of range(n) holding the m
 * smallest of n counter-based keys key(seed, row) (a 32-bit bijective mix of the row under two words derived from the
 * seed): a seed's subsample depends on that seed alone (not on the other seeds,
 * the world size or the device).  seeds: n_seeds int64 on the device.  One workgroup per seed (radix select + ordered
 * compaction).  The table is what symode_aug_gram_gather takes.
 * replaces: the first batch of DataLoader(train_dataset, batch_size=int(len * lbfgs_subsample), shuffle=True) of each
 * seed's process, main.py:36-38 (there: torch's generator, one process per seed). */
int symode_seeded_subsamples(long n, long m, const long long* seeds, int n_seeds, int* idx_out, void* stream);

/* Fused K-step Euler flow f and its tangent map: x_out = f(x), t_out = J_f(x) v  for
 * f = n_steps explicit Euler steps of dx/dt = Theta(x)(xi*mask)^T (all steps in registers).
 * replaces: forward_step = odeint(regressor, ., int_t, int_dt) and jvp(forward_step, x, v_x)[1] of the
 * infinitesimal symmetry regulariser, train.py:669-673 + model_utils.py:56. */
int symode_euler_jvp(const float* x, const float* v, long n, int d, int order, int flags, const float* xi,
                     const float* mask, int n_steps, float dt, float* x_out, float* t_out, void* stream);

/* Reverse mode of symode_euler_jvp, given g_x = dL/d(x_out), g_t = dL/d(t_out):
 * grad_x = dL/dx, grad_v = dL/dv (n, d), grad_xi (d, p) masked.  Step states are recomputed from (x, v).
 * replaces: autograd's double-backward through n_steps regressor calls (model_utils.py:32, 56). */
int symode_euler_jvp_vjp(const float* x, const float* v, const float* g_x, const float* g_t, long n, int d, int order,
                         int flags, const float* xi, const float* mask, int n_steps, float dt, float* grad_x,
                         float* grad_v, float* grad_xi, void* workspace, size_t workspace_bytes, void* stream);

/* L-BFGS search direction d = -H g (two-loop recursion) for n_problems independent problems of n <= 256
 * parameters: curvature pairs in ring buffers old_dirs / old_stps (S, history, n), ro (S, history), with
 * per-problem `head` (oldest slot) and `count` (pairs stored), int64; h_diag (S) scales the initial Hessian.
 * replaces: the history loops of torch.optim.LBFGS.step that train.py:630-695 runs per seed and per
 * inner iteration (new: the reference sweeps seeds as separate processes). */
int symode_lbfgs_direction(const float* g, const float* old_dirs, const float* old_stps, const float* ro,
                           const long* head, const long* count, const float* h_diag, long n_problems, int n,
                           int history, float* d_out, void* stream);

/* Self-test hook: every lane's wave-wide sum of in (n_waves * 64 floats) by the shuffle butterfly and by the
 * permlane-swap / DPP form the L-BFGS kernels use; the two must agree bit for bit (tests/test_gpu_kernels.py). */
int symode_selftest_wave_sum(const float* in, float* butterfly_out, float* dpp_out, long n_waves, void* stream);

/* One inner iteration of torch.optim.LBFGS.step (no line search) for n_problems problems, up to and including the move
 * x += t d: curvature-pair update of the ring buffers, two-loop recursion, step length (first iteration:
 * min(1, 1/|g|_1) lr), directional-derivative test.  One wavefront per problem, one launch for everything
 * torch/optim/lbfgs.py does between two closure evaluations.  State arrays as in symode_lbfgs_direction plus
 * n_iter (S) int64, d / prev_g (S, n), t / prev_loss (S); act (S) bytes: in = problem is active, out = problem moved
 * (its closure must be re-evaluated).  Inactive problems are left untouched.
 * replaces: the body of torch.optim.LBFGS.step that train.py:630-695 runs per seed and per inner iteration. */
int symode_lbfgs_update(float* params, const float* g, const float* loss, unsigned char* act, long* n_iter, float* d,
                        float* t, float* old_dirs, float* old_stps, float* ro, long* head, long* count, float* h_diag,
                        float* prev_g, float* prev_loss, long n_problems, int n, int history, float lr,
                        float tol_change, void* stream);

/* symode_lbfgs_accept (below) followed by symode_lbfgs_update as ONE launch: problems that moved take new_loss / new_g
 * (l1 != 0: as the bare data term, objective w_x * data + w_reg * |params|_1), run the stopping tests, and those still
 * active go straight into the next iteration's update.  Between two closure evaluations the optimiser is one kernel. */
int symode_lbfgs_accept_update(const float* new_loss, const float* new_g, float tol_grad, int l1, float w_x, float w_reg,
                               float* params, float* g, float* loss, unsigned char* act, long* n_iter, float* d, float* t,
                               float* old_dirs, float* old_stps, float* ro, long* head, long* count, float* h_diag,
                               float* prev_g, float* prev_loss, long n_problems, int n, int history, float lr,
                               float tol_change, void* stream);

/* Second half of that iteration, after the closure was re-evaluated at the moved parameters: problems with act != 0
 * take new_loss / new_g into loss / g and run the three stopping tests of torch/optim/lbfgs.py (max|g| <= tol_grad,
 * max|t d| <= tol_change, |loss - prev_loss| < tol_change); act: in = moved, out = still active.
 * params != NULL: new_loss / new_g are the bare data term and the objective is w_x * loss + w_reg * |params|_1 (the L1
 * term over the raw parameters, train.py:680-688), gradient w_x * g + w_reg * sign(params), formed here. */
int symode_lbfgs_accept(const float* new_loss, const float* new_g, float* loss, float* g, unsigned char* act, const float* d,
                        const float* t, const float* prev_loss, long n_problems, int n, float tol_grad, float tol_change,
                        const float* params, float w_x, float w_reg, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Device-resident L-BFGS trainer: train_SIGED_lbfgs (non-latent branch) for n_problems independent problems with NOTHING
 * on the host between two epochs.  An epoch is
 *     closure, BEGIN-update, (closure, ACCEPT-update) x (max_iter - 1), epoch-end
 * -- 2 max_iter + 1 launches: the closure is symode_loss_grad (or symode_loss_grad_reversed when gx / jgx are given), the
 * update launches are the kernel of symode_lbfgs_accept_update extended by the coefficient map of the equivariance
 * constraint (Xi = reshape(Q beta) + const and its chain rule, sindy.py:169-176, formed inside the launch), and the
 * epoch-end launch is the per-epoch logic of train.py:697-725: NaN guard, the two update norms, convergence- / period-
 * triggered thresholding (strict >, monotone, sindy.py:192-195) with optimiser reset, twice-converged stop.  Every epoch
 * leaves one record per problem in `log` (and, with log_detail, the coefficients and mask after the epoch's events), which
 * may be pinned host memory: the host reads it after synchronising on the epoch and produces the reference's prints,
 * wandb record and checkpoints from it.
 * replaces: train.py:630-725 (torch.optim.LBFGS.step + the epoch logic), per seed. */
typedef struct symode_trainer {
    /* closure data, read only: S = n_problems problems back to back */
    const float* x;            /* (S, n_points, d) */
    const float* dx;           /* (S, n_points, d) */
    const float* gx;           /* (S, n_g, n_points, d) reversed-regulariser operands, or NULL (plain MSE closure) */
    const float* jgx;          /* (S, n_g, n_points, d, d) */
    int n_g;
    float w_sym;               /* regulariser weight relative to the MSE: data term = mse + w_sym * regulariser */
    long n_problems, n_points;
    int d, order, flags;
    float inv_count;           /* 1 / (points over all ranks * d) */
    void* workspace;           /* reduction scratch of the closure kernel (symode_workspace_bytes / _init) */
    size_t workspace_bytes;
    /* parametrisation */
    const float* q_eff;        /* (d p, r) row-major: Q with rows in Xi's (d, p) order; NULL = unconstrained (params are Xi) */
    int r;                     /* columns of Q */
    int allow_constant;        /* const is added to column 0 of Xi (sindy.py:173-175) */
    int n_params;              /* d p, or r + d under the constraint */
    /* objective  w_x * data + w_reg * |params|_1  (l1 != 0) and torch.optim.LBFGS's settings */
    float w_x, w_reg;
    int l1;
    float lr, tol_grad, tol_change;
    int max_iter, history;
    /* epoch logic (train.py:697-725) */
    float threshold, tol_update, near_band;
    int st_freq;
    /* state: ONE device allocation of symode_trainer_layout(...) bytes, prepared by symode_trainer_init */
    void* state;
    size_t state_bytes;
    /* per-epoch records, ring of log_epochs epochs (device-visible memory; pinned host memory allowed):
     *   log        (log_epochs, S, 8): [event code, mse, regulariser, |params|_1 of the last closure, update norm,
     *              update norm vs the last convergence, near-threshold coefficients at this event, epoch]
     *              event code: 0 none, 1 threshold on convergence, 2 threshold on st_freq, 3 final convergence, 4 NaN,
     *              -1 problem already finished
     *   log_test   (log_epochs, S, 2): closure value at the epoch's final coefficients and mask (train.py:739-751), written
     *              by symode_trainer_run(..., test_eval != 0)
     *   log_xi, log_mask (log_epochs, S, d p) and log_params (log_epochs, S, n_params), or all NULL: coefficients, mask and
     *              parameters after the epoch's events */
    float* log;
    float* log_test;
    float* log_xi;
    float* log_mask;
    float* log_params;
    int log_epochs;
} symode_trainer;

#define SYMODE_TRAINER_FIELDS 29
/* Byte offsets of the state block's arrays (offsets_out[SYMODE_TRAINER_FIELDS], may be NULL) and its total size.  Order:
 * params, xi, mask, cl_loss (S, 2), cl_grad (S, d p), g, loss, act (u8), n_iter (i64), d, t, old_dirs, old_stps, ro, head
 * (i64), count (i64), h_diag, prev_g, prev_loss, prev, pprev, n_iters (i32), done (u8), nan (u8), finished (u8), epochs
 * (i32), near (i32), l1_last, test_grad (S, d p); fp32 unless stated.  Unconstrained problems keep ONE array for params and xi.  cl_loss and
 * cl_grad are adjacent: a sharded run sums [cl_loss | cl_grad] over the ranks between closure and update. */
size_t symode_trainer_layout(long n_problems, int n_params, int dp, int history, int constrained, size_t* offsets_out);

/* Prepare the state block: zero it, h_diag = 1, params = prev = pprev = params0 (S, n_params), mask = mask0 (S, d p; NULL =
 * ones), xi from params.  params0 / mask0 may be host or device pointers (copied with hipMemcpyAsync). */
int symode_trainer_init(const symode_trainer* T, const float* params0, const float* mask0, void* stream);

/* The three launches of an epoch on their own (a sharded run all-reduces [cl_loss | cl_grad] between closure and update).
 * symode_trainer_closure: loss_out / grad_out NULL = the state's cl_loss / cl_grad.  symode_trainer_update: mode 2 = BEGIN
 * (first iteration of an optimiser step), 1 = ACCEPT (all later ones). */
int symode_trainer_closure(const symode_trainer* T, float* loss_out, float* grad_out, void* stream);
int symode_trainer_update(const symode_trainer* T, int mode, void* stream);
int symode_trainer_epoch_end(const symode_trainer* T, int epoch, void* stream);

/* n_epochs whole epochs starting at epoch0, enqueued back to back; test_eval != 0 adds one closure launch per epoch into
 * log_test.  Problems that finished earlier cost empty launches only. */
int symode_trainer_run(const symode_trainer* T, int epoch0, int n_epochs, int test_eval, void* stream);

/* HOST function (no GPU work): least squares on the normal equations G = A^T A (n, n), C = A^T b (n, k), fp64
 * row-major host arrays; A had m_rows rows.  driver 0 = LAPACK gelsy semantics (pivoted QR rank rule with
 * rcond < 0 -> torch's default eps_fp32 * max(m_rows, n), minimum-norm solution), driver 1 = gels (full rank).
 * W (n, k) receives the solution, *rank_out the numerical rank.
 * replaces: torch.linalg.lstsq(A, B) on the ridge-augmented / block-diagonal system, sindy.py:288. */
int symode_host_lstsq_normal(const double* G, const double* C, int n, int k, long m_rows, int driver, double rcond,
                             double* W, int* rank_out);

/* HOST function (no GPU work): sequential-threshold least squares to convergence for S problems from their augmented Gram
 * matrices G (S, p+d, p+d) fp64 (symode_aug_gram / _gather output copied to the host); n_points rows each, ridge weight
 * gamma, strict > threshold on fp32-rounded coefficients, at most max_iter passes, driver as symode_host_lstsq_normal.
 * xi_out (S, d, p) fp32, mask_out (S, d, p) bytes, passes_out (S); near_out (S, may be NULL): coefficients met within
 * near_band of the threshold; fallback_out (S, may be NULL): singular systems handed to the
 * rank-revealing solve.
 * replaces: the per-seed loop of train.py:872-887 over solve_SINDy_one_step (sindy.py:250-315), unconstrained case. */
int symode_host_stlsq_sweep(const double* G, int S, int d, int p, long n_points, double gamma, double threshold, int max_iter,
                            int driver, double near_band, float* xi_out, unsigned char* mask_out, int* passes_out,
                            int* near_out, int* fallback_out);

#ifdef __cplusplus
}
#endif
#endif /* SYMODE_H */
