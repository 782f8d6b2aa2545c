// C ABI of libsymode_hip.so (declarations and per-entry reference citations: include/symode.h).
// Argument validation happens here, on the host, before anything is launched: a call that
// returns a negative code has touched no device memory.
#include <cstdint>

#include "../../include/symode.h"
#include "ops_table.hpp"

using namespace symode;

namespace {

const LibOps* find_ops(int d, int order, int flags) {
    if (order < 1 || order > MAX_ORDER || flags < 0 || flags > 3) return nullptr;
#define SYMODE_CASE(D, O) case D * 10 + O: return ops_d##D##_o##O(flags);
    switch (d * 10 + order) {
        SYMODE_CASE(1, 1) SYMODE_CASE(1, 2) SYMODE_CASE(1, 3) SYMODE_CASE(1, 4) SYMODE_CASE(1, 5)
        SYMODE_CASE(2, 1) SYMODE_CASE(2, 2) SYMODE_CASE(2, 3) SYMODE_CASE(2, 4) SYMODE_CASE(2, 5)
        SYMODE_CASE(3, 1) SYMODE_CASE(3, 2) SYMODE_CASE(3, 3) SYMODE_CASE(3, 4)
#ifdef SYMODE_WITH_D4                                 // `make ALL=1`: no task of the reference has four state variables
        SYMODE_CASE(4, 1) SYMODE_CASE(4, 2) SYMODE_CASE(4, 3)
#endif
        default: return nullptr;                      // (d, order) outside the compiled set
    }
#undef SYMODE_CASE
}

inline bool misaligned(const void* p, size_t a) { return ((uintptr_t)p % a) != 0; }

// Gram passes are MFMA-bound, not latency-bound: a bounded number of workgroups per launch (default 1024 = 4 per CU:
// one wave per SIMD cannot cover the LDS round trip between MFMAs; SYMODE_GRAM_GRID for tuning runs) keeps the
// number of partials the single finalize block has to add small.
inline int gram_grid(long n, long S) {
    const long total = knobs().gram_grid < 0 ? 1024 : (knobs().gram_grid < 2 ? 2 : knobs().gram_grid);
    const long g = grid_x_by_points(n, 1);
    long cap = S >= total / 2 ? 2 : total / S;
    // one problem: the single finalize block adds every partial (6 KB each at two tiles): 125 000 points cost 84 us on
    // 1024 workgroups, 34 us on 128; 1 M points 178 vs 83 us on 256; 16 M 596 vs 564 us on 512 (r02_single_grid.txt)
    if (S == 1 && knobs().gram_grid < 0) cap = n <= 300000 ? 128 : n <= 2000000 ? 256 : n <= 32000000 ? 512 : total;
    return (int)(g > cap ? cap : g);
}

// Workgroups of ONE closure problem (the last of them adds the partial rows alone; all of them stream one moving window
// of memory): measured best counts (profiles/r02_single_grid.txt, us per launch at cap 128 / 256 / 512 / round-2 default):
//   Theta + residual + grad, 16 B/point:   1 M points 10.5 / 11.4 / 15.7 / 10.3    4 M  19.0 / 16.5 / 20.5 / 30.0
//                                          16 M       58.8 / 44.9 / 48.3 / 51.6    64 M  211 / 169.5 / 175 / 195.5
//   fused with the regulariser, 40 B/point: 1 M       24.5 / 17.1 / 17.8 / 24.7    4 M  80.7 / 47.0 / 36.9 / 43.5
//                                          16 M        308 /  170 /  113 /  118    64 M 1195 /  645 /  432 /  447
// SYMODE_SMALL_GRID fixes the cap for tuning runs (0 = none).
inline int small_grid_cap(long n, bool with_regulariser) {
    if (knobs().small_grid >= 0) return (int)knobs().small_grid;
    if (with_regulariser) return n <= 300000 ? 128 : (n <= 1500000 ? 256 : 512);
    return n <= 1500000 ? 128 : 256;
}

// The vector-pipe Gram (gram_valu.hpp) holds two workgroups per CU (register-bound): one balanced round of them for a
// single large problem, the streaming grid for batches; its partial rows are 78 doubles, so many workgroups are cheap.
inline int gram_valu_grid(long n, long S, int d) {
    const long total = knobs().gram_valu_grid < 0 ? 1024 : (knobs().gram_valu_grid < 2 ? 2 : knobs().gram_valu_grid);
    // at least 32 points per thread: the epilogue transposes 78 fp64 sums per thread through LDS (ten rounds), which a
    // thread must amortise over its own 78-fma-per-point work; beyond that, enough workgroups to fill the chip
    long g = (n + 256L * 32 - 1) / (256L * 32);
    long want = (total + S - 1) / S;                       // ~`total` workgroups in all
    if (S == 1 && want > 512 && knobs().gram_valu_grid < 0) want = 512;   // one problem: 16 M points 84 vs 97 us, 64 M equal
    if (g > want) g = want;
    if (g < 1) g = 1;
    return (int)g;
}

// The 4x4-tile matrix-core Gram (gram_m4.hpp): a wave takes 64 points per pass and wants >= 8 passes to pay for its
// epilogue (21 tiles across lanes and waves); ~1024 workgroups in all (3-4 resident per CU), its partials are 336 doubles.
inline int gram_m4_grid(long n, long S) {
    const long total = knobs().gram_m4_grid < 0 ? 1024 : (knobs().gram_m4_grid < 1 ? 1 : knobs().gram_m4_grid);
    long g = (n + 256L * 8 - 1) / (256L * 8);
    // one problem: exactly one resident round (3 workgroups per CU at 134 registers): 16 M points 286 us on 1024, where the
    // last 256 workgroups run on a third of the chip
    const long cap = S >= total / 2 ? 2 : (S == 1 && knobs().gram_m4_grid < 0 ? 768 : total / S);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// One problem through a reduction kernel: the workgroups loop over the data with a grid-wide stride, so all of them read
// one moving window of memory -- fewer, longer-lived workgroups keep that window (and the DRAM pages under it) tighter,
// and the last workgroup has fewer partial rows to add.  The best count grows with the problem
// (profiles/r02_single_grid.txt, us per launch at 64 / 128 / 256 / 512 / 1024 workgroups, d = 2, order 3):
//   125 000 points  vjp 7.8 / 8.2 / 13.3 / 13.2 / 13.2      jvp_vjp 8.5 / 8.4 / 13.8 / 13.3 / 13.3
//   1 M             vjp 18.2 / 13.8 / 13.2 / 16.8 / 16.7    jvp_vjp 31.5 / 21.0 / 17.2 / 21.3 / 21.5
//   8 M             vjp 181 / 103 / 61 / 45.8 / 48.8        jvp_vjp 205 / 112 / 70.2 / 79.3 / 118
//   64 M            vjp 1461 / 780 / 436 / 310 / 289        jvp_vjp 1605 / 841 / 495 / 572 / 561
// (round 3: with the register-ring pipelines the optimum moved to ONE workgroup per CU from 4 M points up --
//  profiles/r03_stream_ab.txt, 64 .. 1024 workgroups at 125 000 .. 64 M points)
// hence a cap per size class: <= 300 K points, <= 2 M, <= 16 M, beyond.  SYMODE_REDUCE_GRID overrides it for tuning runs.
inline int single_problem_grid(int gx, long n, int c_small, int c_mid, int c_large, int c_huge) {
    const int env = (int)knobs().reduce_grid;
    const int c = env > 0 ? env : (n <= 300000 ? c_small : n <= 2000000 ? c_mid : n <= 16000000 ? c_large : c_huge);
    return gx > c ? c : gx;
}

// points per 16-byte chunk step (points.hpp, Chunk<D>::PPT)
inline int ppt_for(int d) { return d == 2 ? 2 : d == 4 ? 1 : 4; }

// scratch (in doubles) of the widest reduction for this library at (S, n)
size_t workspace_doubles(const LibOps* ops, long S, long n) {
    const int F = ops->p + ops->d;
    const int T = (F + 15) / 16;
    const size_t gram_partial = (size_t)(T * (T + 1) / 2) * 256;
    const size_t nacc = 2 + (size_t)ops->d * ops->p;          // widest row: the fused closure keeps two scalar sums
    size_t g_red = (size_t)grid_x_for(n, S, 1, 512);    // widest grid any reduction uses
    if ((size_t)gram_grid(n, S) > g_red) g_red = (size_t)gram_grid(n, S);
    if ((size_t)gram_valu_grid(n, S, ops->d) + 8 > g_red) g_red = (size_t)gram_valu_grid(n, S, ops->d) + 8;   // (+8: the split form rounds up)
    const size_t a = (size_t)S * g_red * nacc;
    size_t b = (size_t)S * g_red * gram_partial;
    const size_t T4 = (size_t)(F + 3) / 4, m4 = (size_t)S * (size_t)gram_m4_grid(n, S) * (T4 * (T4 + 1) / 2 * 16);
    if (m4 > b) b = m4;
    return (size_t)WS_HEADER_DOUBLES + (a > b ? a : b);      // [magic + tickets | partial rows]
}

}  // namespace

extern "C" {

int symode_abi_version(void) { return 5; }

void symode_reload_env(void) { knobs() = read_knobs(); }

const char* symode_error_string(int code) {
    switch (code) {
        case SYMODE_OK: return "ok";
        case SYMODE_E_UNSUPPORTED: return "library (d, order, flags) not compiled into libsymode_hip";
        case SYMODE_E_NULLPTR: return "null pointer argument";
        case SYMODE_E_BADSIZE: return "bad size argument";
        case SYMODE_E_WORKSPACE: return "workspace missing or too small";
        case SYMODE_E_ALIGN: return "misaligned pointer";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown symode error";
    }
}

int symode_lib_size(int d, int order, int flags) {
    const LibOps* ops = find_ops(d, order, flags);
    return ops ? ops->p : SYMODE_E_UNSUPPORTED;
}

size_t symode_workspace_bytes(int d, int order, int flags, long n_problems, long n) {
    const LibOps* ops = find_ops(d, order, flags);
    if (!ops || n_problems < 1 || n < 0) return 0;
    return workspace_doubles(ops, n_problems, n) * sizeof(double);
}

int symode_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return SYMODE_E_NULLPTR;
    if (misaligned(workspace, 8)) return SYMODE_E_ALIGN;
    if (workspace_bytes < (size_t)WS_HEADER_DOUBLES * sizeof(double)) return SYMODE_E_WORKSPACE;
    const long words = WS_HEADER_DOUBLES;
    workspace_init_kernel<0><<<dim3((unsigned)((words + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream>>>(
        (unsigned long long*)workspace, words);
    return (int)hipGetLastError();
}

#define SYMODE_GET_OPS()                              \
    const LibOps* ops = find_ops(d, order, flags);    \
    if (!ops) return SYMODE_E_UNSUPPORTED;

#define SYMODE_CHECK_WS(S_, n_)                                                              \
    if (!workspace || misaligned(workspace, 8)) return SYMODE_E_WORKSPACE;                   \
    if (workspace_bytes < workspace_doubles(ops, (S_), (n_)) * sizeof(double)) return SYMODE_E_WORKSPACE;

int symode_theta(const float* x, long n, int d, int order, int flags, float* theta_out, void* stream) {
    SYMODE_GET_OPS();
    if (n < 0) return SYMODE_E_BADSIZE;
    if (n == 0) return SYMODE_OK;
    if (!x || !theta_out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(theta_out, 4)) return SYMODE_E_ALIGN;
    return (int)ops->theta(x, n, theta_out, (hipStream_t)stream);
}

int symode_forward(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask, float* out,
                   void* stream) {
    SYMODE_GET_OPS();
    if (n < 0) return SYMODE_E_BADSIZE;
    if (n == 0) return SYMODE_OK;
    if (!x || !xi || !out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(out, 4) || misaligned(xi, 4) || misaligned(mask, 4)) return SYMODE_E_ALIGN;
    return (int)ops->forward(x, n, xi, mask, out, (hipStream_t)stream);
}

int symode_odeint(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask, int n_steps,
                  float dt, int method, float* out, void* stream) {
    SYMODE_GET_OPS();
    if (n < 0 || n_steps < 0 || (method != 0 && method != 1)) return SYMODE_E_BADSIZE;
    if (n == 0) return SYMODE_OK;
    if (!x || !xi || !out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(out, 4) || misaligned(xi, 4) || misaligned(mask, 4)) return SYMODE_E_ALIGN;
    return (int)ops->odeint(x, n, xi, mask, n_steps, dt, method, out, (hipStream_t)stream);
}

int symode_odeint_traj(const float* x, long n, int d, int order, int flags, const float* xi, const float* mask, int n_steps,
                       float dt, int method, float* traj, void* stream) {
    SYMODE_GET_OPS();
    if (n < 0 || n_steps < 0 || (method != 0 && method != 1)) return SYMODE_E_BADSIZE;
    if (n == 0 || n_steps == 0) return SYMODE_OK;
    if (!x || !xi || !traj) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(traj, 4) || misaligned(xi, 4) || misaligned(mask, 4)) return SYMODE_E_ALIGN;
    return (int)ops->odeint_traj(x, n, xi, mask, n_steps, dt, method, traj, (hipStream_t)stream);
}

int symode_loss_grad(const float* x, const float* dx, long n_problems, long n, int d, int order, int flags,
                     const float* xi, const float* mask, float inv_count, float* loss_out, float* grad_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n_problems < 1 || n_problems > 65535 || n < 1) return SYMODE_E_BADSIZE;
    if (!x || !dx || !xi || !loss_out || !grad_out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(dx, 4) || misaligned(xi, 4) || misaligned(mask, 4) || misaligned(loss_out, 4) ||
        misaligned(grad_out, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(n_problems, n);
    int gx = grid_x_for(n, n_problems, ppt_for(d));
    // a single latency-bound problem: the last workgroup adds gx partial rows alone, so fewer, longer workgroups win
    // (SYMODE_SMALL_GRID overrides the cap for tuning runs; 0 = no cap)
    if (n_problems == 1) {
        // (order 4-5 libraries are as arithmetic-heavy as the regulariser closures: 64 M points at order 5, 240 us at 256
        //  workgroups, 184 at 512 -- profiles/r03_closure_ab.txt; order 4 (d p = 30): 191 against 176 us; order 3 (20) is
        //  best at 256: 161 against 169)
        const int cap = small_grid_cap(n, ops->d * ops->p > 24);
        if (cap > 0 && gx > cap) gx = cap;
    }
    return (int)ops->loss_grad(x, dx, n_problems, n, xi, mask, inv_count, loss_out, grad_out, (double*)workspace, gx,
                               (hipStream_t)stream);
}

int symode_aug_gram(const float* x, const float* dx, long n_problems, long n, int d, int order, int flags,
                    double* gram_out, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n_problems < 1 || n_problems > 65535 || n < 1) return SYMODE_E_BADSIZE;
    if (!x || !dx || !gram_out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(dx, 4) || misaligned(gram_out, 8)) return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(n_problems, n);
    return (int)ops->aug_gram(x, dx, n_problems, n, nullptr, gram_out, (double*)workspace, gram_grid(n, n_problems),
                              gram_valu_grid(n, n_problems, d), gram_m4_grid(n, n_problems), (hipStream_t)stream);
}

int symode_aug_gram_gather(const float* x, const float* dx, long n_src, const int* idx, long n_problems, long m, int d,
                           int order, int flags, double* gram_out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    SYMODE_GET_OPS();
    if (n_problems < 1 || n_problems > 65535 || m < 1 || n_src < 1 || n_src > 2147483647L) return SYMODE_E_BADSIZE;
    if (!x || !dx || !idx || !gram_out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(dx, 4) || misaligned(idx, 4) || misaligned(gram_out, 8)) return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(n_problems, m);
    return (int)ops->aug_gram(x, dx, n_problems, m, idx, gram_out, (double*)workspace, gram_grid(m, n_problems),
                              gram_valu_grid(m, n_problems, d), gram_m4_grid(m, n_problems), (hipStream_t)stream);
}

int symode_symreg_linear(const float* z, long n, int d, int order, int flags, const float* xi, const float* mask,
                         const float* L, int n_gen, float* loss_out, float* grad_out, void* workspace,
                         size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1 || n_gen < 0) return SYMODE_E_BADSIZE;
    if (!z || !xi || !loss_out || !grad_out || (n_gen > 0 && !L)) return SYMODE_E_NULLPTR;
    if (misaligned(z, 4) || misaligned(xi, 4) || misaligned(mask, 4) || misaligned(L, 4)) return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(1, n);
    const int gx = single_problem_grid(grid_x_for(n, 1, ppt_for(d)), n, 128, 256, 512, 512);      // r03_stream_ab.txt
    return (int)ops->symreg_linear(z, n, xi, mask, L, n_gen, loss_out, grad_out, (double*)workspace, gx,
                                   (hipStream_t)stream);
}

int symode_symreg_reversed_batched(const float* x, const float* gx_, const float* jgx, int n_g, long n_problems, long n, int d,
                                   int order, int flags, const float* xi, const float* mask, float inv_count,
                                   float* loss_out, float* grad_out, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1 || n_g < 0 || n_problems < 1 || n_problems > 65535) return SYMODE_E_BADSIZE;
    if (!x || !xi || !loss_out || !grad_out || (n_g > 0 && (!gx_ || !jgx))) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(gx_, 4) || misaligned(jgx, 4) || misaligned(xi, 4) || misaligned(mask, 4) ||
        misaligned(loss_out, 4) || misaligned(grad_out, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(n_problems, n);
    int gx = grid_x_for(n, n_problems, ppt_for(d), 512);
    if (n_problems == 1) {
        // (32 B/point: the small libraries stream best from ONE workgroup per CU like the other reductions -- order 3 at 2^26
        //  points 314 us = 0.855 of HBM on 256 workgroups against 338 on 512; order 4-5 want the second one: 340 against 374 us)
        int cap = small_grid_cap(n, true);
        if (cap > 256 && ops->d * ops->p <= 24 && knobs().small_grid < 0) cap = 256;
        if (cap > 0 && gx > cap) gx = cap;
    }
    return (int)ops->symreg_reversed(x, nullptr, gx_, jgx, n_g, n_problems, n, xi, mask, inv_count, 1.0f, loss_out, grad_out,
                                     (double*)workspace, gx, (hipStream_t)stream);
}

int symode_loss_grad_reversed(const float* x, const float* dx, const float* gx_, const float* jgx, int n_g, long n_problems, long n,
                              int d, int order, int flags, const float* xi, const float* mask, float inv_count, float w_sym,
                              float* loss2_out, float* grad_out, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1 || n_g < 1 || n_problems < 1 || n_problems > 65535) return SYMODE_E_BADSIZE;
    if (!x || !dx || !xi || !loss2_out || !grad_out || !gx_ || !jgx) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(dx, 4) || misaligned(gx_, 4) || misaligned(jgx, 4) || misaligned(xi, 4) || misaligned(mask, 4) ||
        misaligned(loss2_out, 4) || misaligned(grad_out, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(n_problems, n);
    int gx = grid_x_for(n, n_problems, ppt_for(d), 512);
    if (n_problems == 1) {
        const int cap = small_grid_cap(n, true);
        if (cap > 0 && gx > cap) gx = cap;
    }
    return (int)ops->symreg_reversed(x, dx, gx_, jgx, n_g, n_problems, n, xi, mask, inv_count, w_sym, loss2_out, grad_out,
                                     (double*)workspace, gx, (hipStream_t)stream);
}

int symode_symreg_reversed(const float* x, const float* gx_, const float* jgx, int n_g, long n, int d, int order,
                           int flags, const float* xi, const float* mask, float* loss_out, float* grad_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || d < 1) return SYMODE_E_BADSIZE;
    return symode_symreg_reversed_batched(x, gx_, jgx, n_g, 1, n, d, order, flags, xi, mask, 1.0f / ((float)n * (float)d),
                                          loss_out, grad_out, workspace, workspace_bytes, stream);
}

int symode_weak_gram(const float* x, long n_t, int d, int order, int flags, const float* V, const float* V_drv, int n_test,
                     double* out, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n_t < 1 || n_test < 1 || n_test > 128) return SYMODE_E_BADSIZE;
    if (!x || !V || !V_drv || !out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(V, 4) || misaligned(V_drv, 4) || misaligned(out, 8)) return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(1, n_t);
    const int RT = (2 * n_test + 15) / 16, CT = (ops->p + ops->d + 15) / 16;
    long gx = (n_t + 255) / 256;                                   // a workgroup = 4 waves x 64 time points per step
    if (gx > 64) gx = 64;
    const size_t avail = workspace_bytes / sizeof(double) - (size_t)WS_HEADER_DOUBLES;
    while (gx > 1 && (size_t)gx * RT * CT * 256 > avail) gx /= 2;
    if ((size_t)gx * RT * CT * 256 > avail) return SYMODE_E_WORKSPACE;
    return (int)ops->weak_gram(x, n_t, V, V_drv, n_test, out, (double*)workspace, (int)gx, (hipStream_t)stream);
}

int symode_vjp(const float* x, const float* g, long n, int d, int order, int flags, const float* xi, const float* mask,
               float* grad_x, float* grad_xi, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1) return SYMODE_E_BADSIZE;
    if (!x || !g || !xi || !grad_xi) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(g, 4) || misaligned(xi, 4) || misaligned(mask, 4) || misaligned(grad_x, 4) ||
        misaligned(grad_xi, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(1, n);
    // register ring of 3 chunks per lane: one workgroup per CU streams best from 4 M points up (round 3,
    // profiles/r03_stream_ab.txt: 64 M points 262 us at 256 workgroups against 321 at 1024; without grad_x 152 against 188)
    // (order 5, d p = 42: the second workgroup per CU pays from 8 M points -- 2^26 points 316 -> 295 us; orders 3-4 lose 8-11 % on it)
    const bool heavy = ops->d * ops->p > 32 && grad_x != nullptr;        // (without grad_x one per CU stays best: 163 against 175 us)
    const int gx = single_problem_grid(grid_x_for(n, 1, ppt_for(d)), n, 64, 128, heavy ? 512 : 256, heavy ? 512 : 256);
    return (int)ops->vjp(x, g, n, xi, mask, grad_x, grad_xi, (double*)workspace, gx, (hipStream_t)stream);
}

int symode_forward_jvp(const float* x, const float* v, long n, int d, int order, int flags, const float* xi,
                       const float* mask, float* out, float* jv, void* stream) {
    SYMODE_GET_OPS();
    if (n < 0) return SYMODE_E_BADSIZE;
    if (n == 0) return SYMODE_OK;
    if (!x || !v || !xi || !jv) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(v, 4) || misaligned(xi, 4) || misaligned(mask, 4) || misaligned(out, 4) ||
        misaligned(jv, 4))
        return SYMODE_E_ALIGN;
    return (int)ops->forward_jvp(x, v, n, xi, mask, out, jv, (hipStream_t)stream);
}

int symode_jvp_vjp(const float* x, const float* v, const float* g_out, const float* g_jv, long n, int d, int order,
                   int flags, const float* xi, const float* mask, float* grad_x, float* grad_v, float* grad_xi,
                   void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1) return SYMODE_E_BADSIZE;
    if (!x || !v || !g_jv || !xi || !grad_x || !grad_v || !grad_xi) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(v, 4) || misaligned(g_out, 4) || misaligned(g_jv, 4) || misaligned(xi, 4) ||
        misaligned(mask, 4) || misaligned(grad_x, 4) || misaligned(grad_v, 4) || misaligned(grad_xi, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(1, n);
    // ring of 3: r03_stream_ab.txt.  Order 5 (d p = 42) wants more than one workgroup per CU: 2^26 points, us at 256 / 512 /
    // 768 workgroups 693 / 652 / 646 on one box, 727 / - / 672-694 on another; 8 M points 94 -> 82 us at 512.  Orders 3-4
    // stay at one per CU (768 won on one box, 531 against 546 us, and lost on the next, 586-602 against 545-567)
    const bool heavy = ops->d * ops->p > 32;
    const int gx = single_problem_grid(grid_x_for(n, 1, ppt_for(d)), n, 128, 256, heavy ? 512 : 256, heavy ? 768 : 256);
    return (int)ops->jvp_vjp(x, v, g_out, g_jv, n, xi, mask, grad_x, grad_v, grad_xi, (double*)workspace, gx,
                             (hipStream_t)stream);
}

int symode_rk4_traj(const double* x0, long n_traj, int d, int order, int flags, const double* xi, int n_steps, double dt,
                    int subsample, float* x_out, float* dx_out, void* stream) {
    SYMODE_GET_OPS();
    if (n_traj < 0 || n_steps < 1 || subsample < 1) return SYMODE_E_BADSIZE;
    if (n_traj == 0) return SYMODE_OK;
    if (!x0 || !xi || !x_out || !dx_out) return SYMODE_E_NULLPTR;
    if (misaligned(x0, 8) || misaligned(xi, 8) || misaligned(x_out, 4) || misaligned(dx_out, 4)) return SYMODE_E_ALIGN;
    return (int)ops->rk4_traj(x0, n_traj, xi, n_steps, dt, subsample, x_out, dx_out, (hipStream_t)stream);
}

int symode_euler_jvp(const float* x, const float* v, long n, int d, int order, int flags, const float* xi,
                     const float* mask, int n_steps, float dt, float* x_out, float* t_out, void* stream) {
    SYMODE_GET_OPS();
    if (n < 0 || n_steps < 0) return SYMODE_E_BADSIZE;
    if (n == 0) return SYMODE_OK;
    if (!x || !v || !xi || !x_out || !t_out) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(v, 4) || misaligned(xi, 4) || misaligned(mask, 4) || misaligned(x_out, 4) ||
        misaligned(t_out, 4))
        return SYMODE_E_ALIGN;
    return (int)ops->euler_jvp(x, v, n, xi, mask, n_steps, dt, x_out, t_out, (hipStream_t)stream);
}

int symode_euler_jvp_vjp(const float* x, const float* v, const float* g_x, const float* g_t, long n, int d, int order,
                         int flags, const float* xi, const float* mask, int n_steps, float dt, float* grad_x,
                         float* grad_v, float* grad_xi, void* workspace, size_t workspace_bytes, void* stream) {
    SYMODE_GET_OPS();
    if (n < 1 || n_steps < 0) return SYMODE_E_BADSIZE;
    if (!x || !v || !g_x || !g_t || !xi || !grad_x || !grad_v || !grad_xi) return SYMODE_E_NULLPTR;
    if (misaligned(x, 4) || misaligned(v, 4) || misaligned(g_x, 4) || misaligned(g_t, 4) || misaligned(xi, 4) ||
        misaligned(mask, 4) || misaligned(grad_x, 4) || misaligned(grad_v, 4) || misaligned(grad_xi, 4))
        return SYMODE_E_ALIGN;
    SYMODE_CHECK_WS(1, n);
    const int gx = grid_x_for(n, 1, ppt_for(d));
    return (int)ops->euler_jvp_vjp(x, v, g_x, g_t, n, xi, mask, n_steps, dt, grad_x, grad_v, grad_xi,
                                   (double*)workspace, gx, (hipStream_t)stream);
}

}  // extern "C"
