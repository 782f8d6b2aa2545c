// HIP kernels of the SINDy hot path, templated on the compile-time library description.
// gfx950 only: 64-wide wavefronts, 256-thread workgroups (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "library.hpp"
#include "points.hpp"
#include "reduce.hpp"

namespace symode {

constexpr int BLOCK = 256;
#ifndef SYMODE_MAP_CHUNKS
#define SYMODE_MAP_CHUNKS 1
#endif
constexpr int MAP_CHUNKS_PER_THREAD = SYMODE_MAP_CHUNKS;   // chunks a thread of a map kernel visits (grid = index space / this)

// Every SYMODE_* tuning / A-B variable the library knows, read ONCE when the library is first used.
// A value of -1 means "not set: the library's own rule".  symode_reload_env() (C ABI) reads the environment again --
// for the tests and tuning tools that compare two settings inside one process; no launch path calls getenv.
struct Knobs {
    long max_grid, min_grid_x, map_grid, gram_grid, gram_valu_grid, small_grid, reduce_grid;
    int fused_finalize, euler_stack, gram_valu, gram_split, gram_valu_gather, segmented, row_split, gram_m4;
    long gram_m4_grid;
};

inline Knobs read_knobs() {
    auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
    auto on = [](const char* name) { const char* e = getenv(name); return (e && e[0] == '0') ? 0 : 1; };
    Knobs k;
    k.max_grid = num("SYMODE_MAX_GRID", -1);
    k.min_grid_x = num("SYMODE_MIN_GRID_X", 2);
    k.map_grid = num("SYMODE_MAP_GRID", 1L << 20);
    k.gram_grid = num("SYMODE_GRAM_GRID", -1);
    k.gram_valu_grid = num("SYMODE_GRAM_VALU_GRID", -1);
    k.small_grid = num("SYMODE_SMALL_GRID", -1);
    k.reduce_grid = num("SYMODE_REDUCE_GRID", -1);
    k.fused_finalize = (int)num("SYMODE_FUSED_FINALIZE", 1);
    k.euler_stack = on("SYMODE_EULER_STACK");
    k.gram_valu = on("SYMODE_GRAM_VALU");
    k.gram_split = on("SYMODE_GRAM_SPLIT");
    k.gram_valu_gather = on("SYMODE_GRAM_VALU_GATHER");
    k.segmented = (int)num("SYMODE_SEGMENTED", 1);
    k.row_split = (int)num("SYMODE_ROW_SPLIT", 1);
    k.gram_m4 = on("SYMODE_GRAM_M4");
    k.gram_m4_grid = num("SYMODE_GRAM_M4_GRID", -1);
    return k;
}

inline Knobs& knobs() {
    static Knobs k = read_knobs();
    return k;
}
// Workgroup budget of a BATCHED reduction launch (S > 1 problems on grid.y).  Every workgroup pays a fixed epilogue (LDS
// transpose of its d*p + 1 sums, a partial row, a ticket) and the last one of each problem adds that problem's rows
// alone, so what a launch wants is few, long-lived workgroups: ~16 K points each, between one per CU and four per CU in
// all.  Measured on loss_grad, order 3 (us at a budget of 256 / 1024 / 8192 workgroups; round 2 shipped 8192):
//   16 x 125 000: 11.9 / 18.1 / 55.3     64 x 50 000: 11.4 / 13.4 / 64.9     256 x 50 000: 32.4 / 35.0 / 70.1
//   512 x 125 000 and up: the floor of two workgroups per problem decides (162-178 us either way).
// SYMODE_MAX_GRID fixes the budget for tuning runs; SYMODE_MIN_GRID_X the floor per problem.
// The fused closure (40 B/point, 3-4 waves per SIMD) wants twice the floor: 16 x 125 000: 26.6 us at 256, 22.4 at 512, 37.0
// at 2048; 64 x 50 000: 38.0 / 27.3 / 36.5 (profiles/r02_batched_grid.txt).
inline long batch_grid_budget(long total_points, long floor_) {
    const long fixed = knobs().max_grid;
    if (fixed > 0) return fixed < 2 ? 2 : fixed;
    long b = total_points / 16384;
    if (b < floor_) b = floor_;
    if (b > 1024) b = 1024;
    return b;
}

inline long min_grid_x() {
    const long v = knobs().min_grid_x;
    return v < 1 ? 1 : v;
}

// One row of the dispatch table: everything the C ABI needs for one (D, ORDER, FLAGS).
struct LibOps {
    int d, order, flags, p;
    hipError_t (*theta)(const float* x, long n, float* out, hipStream_t st);
    hipError_t (*forward)(const float* x, long n, const float* xi, const float* mask, float* out, hipStream_t st);
    hipError_t (*odeint)(const float* x, long n, const float* xi, const float* mask, int n_steps, float dt, int method,
                         float* out, hipStream_t st);
    hipError_t (*odeint_traj)(const float* x, long n, const float* xi, const float* mask, int n_steps, float dt, int method,
                              float* traj, hipStream_t st);
    hipError_t (*loss_grad)(const float* x, const float* dx, long S, long n, const float* xi, const float* mask,
                            float inv_count, float* loss, float* grad, double* ws, int gx, hipStream_t st);
    hipError_t (*symreg_linear)(const float* z, long n, const float* xi, const float* mask, const float* L, int n_gen,
                                float* loss, float* grad, double* ws, int gx, hipStream_t st);
    // dx == nullptr: the regulariser alone (loss (S)); else the fused closure MSE + w_sym * regulariser (loss (S, 2))
    hipError_t (*symreg_reversed)(const float* x, const float* dx, const float* gx_, const float* jgx, int n_g, long S, long n,
                                  const float* xi, const float* mask, float inv_count, float w_sym, float* loss, float* grad,
                                  double* ws, int gx, hipStream_t st);
    hipError_t (*aug_gram)(const float* x, const float* dx, long S, long n, const int* idx, double* gram, double* ws,
                           int gx_mfma, int gx_valu, int gx_m4, hipStream_t st);
    hipError_t (*vjp)(const float* x, const float* g, long n, const float* xi, const float* mask, float* grad_x,
                      float* grad_xi, double* ws, int gx, hipStream_t st);
    hipError_t (*forward_jvp)(const float* x, const float* v, long n, const float* xi, const float* mask, float* out,
                              float* jv, hipStream_t st);
    hipError_t (*jvp_vjp)(const float* x, const float* v, const float* g_out, const float* g_jv, long n, const float* xi,
                          const float* mask, float* grad_x, float* grad_v, float* grad_xi, double* ws, int gx,
                          hipStream_t st);
    hipError_t (*rk4_traj)(const double* x0, long n_traj, const double* xi, int n_steps, double dt, int subsample,
                           float* x_out, float* dx_out, hipStream_t st);
    hipError_t (*euler_jvp)(const float* x, const float* v, long n, const float* xi, const float* mask, int n_steps,
                            float dt, float* x_out, float* t_out, hipStream_t st);
    hipError_t (*euler_jvp_vjp)(const float* x, const float* v, const float* g_x, const float* g_t, long n, const float* xi,
                                const float* mask, int n_steps, float dt, float* grad_x, float* grad_v, float* grad_xi,
                                double* ws, int gx, hipStream_t st);
    hipError_t (*weak_gram)(const float* x, long T, const float* V, const float* Vd, int K, double* out, double* ws, int gx,
                            hipStream_t st);
};

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------

// Grid width for a streaming pass over n points (per problem), S problems on grid.y.
// Small problems favour latency: one step per thread until every CU has two blocks; beyond that,
// at least 4 steps per thread so the reduction epilogue amortises.
inline long grid_x_by_points(long n, int pts_per_thread_iter) {
    const long per_block = (long)BLOCK * pts_per_thread_iter;
    long g = (n + per_block - 1) / per_block;
    if (g > 512) {
        g = (n + 4 * per_block - 1) / (4 * per_block);
        if (g < 512) g = 512;
    }
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

inline int grid_x_for(long n, long S, int pts_per_thread_iter, long batch_floor = 256) {
    long g = grid_x_by_points(n, pts_per_thread_iter);
    long cap = 2048;                   // one problem: the last workgroup adds the partial rows alone, keep them few
    if (S > 1) {
        cap = batch_grid_budget(n * S, batch_floor) / S;
        if (cap < min_grid_x()) cap = min_grid_x();
    }
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of a pure map (no reduction epilogue, so no partial rows to keep few): by default ONE chunk pair per thread --
// the whole index space as workgroups, which the dispatcher streams through the chip in address order (measured on the
// forward map at 2^26 points: 0.77 of the HBM roof, against 0.58-0.71 for 1024-8192 looping workgroups).
// SYMODE_MAP_GRID caps it for tuning runs.
inline int map_grid_for(long n, int pts_per_thread_iter) {
    const long cap = knobs().map_grid;
    const long per_block = (long)BLOCK * pts_per_thread_iter * MAP_CHUNKS_PER_THREAD;
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Rounds (of BLOCK chunks) per workgroup of the forward map: a workgroup starts with D*P coefficient (and mask) loads and
// products, which at order 4-5 cost as much issue time as two points of work.  2^26 points, d = 2, us at 1 / 4 rounds, same
// box: order 4 208 / 172, order 5 209-216 / 169 (0.63 -> 0.79 of 8 TB/s); order 3 and the d = 3 libraries (coalesced-tile
// loads) are best at 1.  forward_jvp (two more streams) moved by +-5 % either way over 1 / 2 / 4 rounds and coefficients
// in VGPRs / SGPRs -- left at 1 (profiles/r03_map_rounds.txt).
template <class Lib>
constexpr int map_rounds = (Lib::D == 2 && Lib::D * Lib::P > 24) ? 4 : 1;

inline bool vec_ok(const void* p, long n, int d, long S) {
    return ((uintptr_t)p % 16 == 0) && (S == 1 || (n * d) % 4 == 0);
}

#ifndef SYMODE_ROWS_XI_SGPR
#define SYMODE_ROWS_XI_SGPR 1       // loss_grad_rows_kernel: its row of Xi in SGPRs (1) or VGPRs (0); measured 555 vs 587 us at p = 35
#endif
// D = 3 libraries with sine / exp columns stream point by point, not in 16-byte chunks: a chunk is four points, and four
// interleaved copies of a per-point body with inlined sinf / expf want more than the register file (odeint, sine + exp:
// 248 VGPRs and 480 bytes of scratch per lane).  Those kernels are bound by the transcendentals, not by how the 12-byte
// points arrive; the per-point path is the one every kernel already has for unaligned bases.
template <class Lib>
constexpr bool chunked_stream = !((Lib::D == 3) && (Lib::SINE || Lib::EXP));

constexpr int VGPR_XI_MAX = 48;     // up to here the masked coefficients simply stay in VGPRs (see load_xi)
constexpr int SGPR_XI_MAX = 64;     // VGPR_XI_MAX < D*P <= this: Xi in SGPRs (the wave has ~100 of them: d = 3 order 3, d = 4 order 2)

// Masked coefficients of problem s into registers (uniform across the block -> scalar loads).
template <class Lib, int SGPR_FROM = VGPR_XI_MAX + 1, int SGPR_TO = SGPR_XI_MAX>
__device__ __forceinline__ void load_xi(const float* __restrict__ xi, const float* __restrict__ mask, long s,
                                        float (&w)[Lib::D * Lib::P]) {
    constexpr int DP = Lib::D * Lib::P;
    const float* a = xi + s * DP;
#pragma unroll
    for (int i = 0; i < DP; ++i) w[i] = a[i];
    if (mask != nullptr) {
        const float* m = mask + s * DP;
#pragma unroll
        for (int i = 0; i < DP; ++i) w[i] *= m[i];
    }
    // The masked product is a VALU result and stays in VGPRs for small libraries: an SGPR operand costs issue time
    // (constant-bus read; measured with the register-ring kernel: Xi in VGPRs +1.5 % at d = 2 order 5, +4.5 % at order 3).
    // Mid-size libraries (48 < D*P <= 64: d = 3 order 3, d = 4 order 2) are short of registers instead, so their
    // coefficients -- wave-uniform values -- go back to the scalar file and the D*P VGPRs become occupancy (6.4 -> 6.8 TB/s).
    // (SGPR_FROM: kernels with more per-point state than K1 move the coefficients out of the vector file earlier.)
    if constexpr (DP >= SGPR_FROM && DP <= SGPR_TO) {
#pragma unroll
        for (int i = 0; i < DP; ++i)
            w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[i])));
    }
}

// h = Theta . Xi_m^T for one point (fp32 fma chain over the library columns).
template <class Lib>
__device__ __forceinline__ void apply_xi(const float (&w)[Lib::D * Lib::P], const float (&th)[Lib::P],
                                         float (&h)[Lib::D]) {
#pragma unroll
    for (int j = 0; j < Lib::D; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < Lib::P; ++k) s = fmaf(w[j * Lib::P + k], th[k], s);
        h[j] = s;
    }
}

template <class Lib>
__device__ __forceinline__ void rhs(const float (&w)[Lib::D * Lib::P], const float (&x)[Lib::D], float (&h)[Lib::D]) {
    float th[Lib::P];
    Lib::eval(x, th);
    apply_xi<Lib>(w, th, h);
}

// ---------------------------------------------------------------------------------------
// Theta(x) materialised                                        (compat: eval_Theta_at)
// ---------------------------------------------------------------------------------------
// A workgroup turns 256 consecutive points into a (256, P) tile: thread-per-point rows go to LDS
// (row stride odd -> conflict-free), then the tile -- contiguous in global memory -- leaves as
// coalesced 16-byte stores.  Direct per-thread row stores reach 2.7 TB/s, this form is store-bound.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void theta_kernel(const float* __restrict__ x, long N, bool vec,
                                                      float* __restrict__ out) {
    constexpr int D = Lib::D, P = Lib::P, PS = (P % 2 == 0) ? P + 1 : P;
    __shared__ float tile[BLOCK * PS];
    const int tid = threadIdx.x;
    for (long base = (long)blockIdx.x * BLOCK; base < N; base += (long)gridDim.x * BLOCK) {
        const long n = base + tid;
        if (n < N) {
            float xp[D], th[P];
            load_point<D>(x, n, xp);
            Lib::eval(xp, th);
#pragma unroll
            for (int k = 0; k < P; ++k) tile[tid * PS + k] = th[k];
        }
        __syncthreads();
        const long rows = (N - base < BLOCK) ? (N - base) : BLOCK;
        const long total = rows * P;
        float* dst = out + base * P;
        const long nvec = vec ? total / 4 : 0;                  // vec: `out` is 16-byte aligned (base*P*4 always is)
        for (long v = tid; v < nvec; v += BLOCK) {
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = (int)(4 * v + i);
                e[i] = tile[(f / P) * PS + (f % P)];
            }
            typedef float f4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f4v{e[0], e[1], e[2], e[3]}, reinterpret_cast<f4v*>(dst) + v);     // written once, never read here
        }
        for (long f = 4 * nvec + tid; f < total; f += BLOCK) dst[f] = tile[(int)(f / P) * PS + (int)(f % P)];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// forward: out = Theta(x) Xi_m^T                               (regressor.forward)
// ---------------------------------------------------------------------------------------
template <class Lib>
__global__ __launch_bounds__(BLOCK) void forward_kernel(const float* __restrict__ x, long N, bool vec,
                                                        const float* __restrict__ xi, const float* __restrict__ mask,
                                                        float* __restrict__ out) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, PPT = Chunk<D>::PPT;
    float w[D * Lib::P];
    load_xi<Lib>(xi, mask, 0, w);
    struct Ops {
        float x[PPT][D];
    };
    for_each_chunk2<D, BLOCK, Ops, map_rounds<Lib>>(
        N, vec, [&](long c, Ops& o) { load_chunk<D>(x, c, o.x); },
        [&](long c, Ops& o) {
            float h[PPT][D];
            each_point<PPT>([&](auto i) { rhs<Lib>(w, o.x[i], h[i]); });
            store_chunk_nt<D>(out, c, h);
        },
        [&](long n) {
            float xp[D], h[D];
            load_point<D>(x, n, xp);
            rhs<Lib>(w, xp, h);
            store_point<D>(out, n, h);
        });
}

// ---------------------------------------------------------------------------------------
// fixed-step integrator: K Euler / RK4 steps of dx/dt = Theta(x) Xi_m^T   (odeint)
// ---------------------------------------------------------------------------------------
template <class Lib, class OnStep>
__device__ __forceinline__ void integrate_steps(const float (&w)[Lib::D * Lib::P], float (&x)[Lib::D], int n_steps, float dt,
                                                int method, OnStep on_step) {
    constexpr int D = Lib::D;
    if (method == 0) {
        for (int s = 0; s < n_steps; ++s) {
            float h[D];
            rhs<Lib>(w, x, h);
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] = x[j] + dt * h[j];               // model_utils.py:238
            on_step(s, x);
        }
    } else {
        for (int s = 0; s < n_steps; ++s) {                                    // model_utils.py:242-247
            float k1[D], k2[D], k3[D], k4[D], y[D];
            rhs<Lib>(w, x, k1);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = x[j] + dt / 2 * k1[j];
            rhs<Lib>(w, y, k2);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = x[j] + dt / 2 * k2[j];
            rhs<Lib>(w, y, k3);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = x[j] + dt * k3[j];
            rhs<Lib>(w, y, k4);
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] = x[j] + dt / 6 * (k1[j] + 2 * k2[j] + 2 * k3[j] + k4[j]);
            on_step(s, x);
        }
    }
}

template <class Lib>
__device__ __forceinline__ void integrate(const float (&w)[Lib::D * Lib::P], float (&x)[Lib::D], int n_steps, float dt,
                                          int method) {
    integrate_steps<Lib>(w, x, n_steps, dt, method, [](int, const float (&)[Lib::D]) {});
}

// Full trajectory (odeint(..., full_traj=True), model_utils.py:249-254): a thread per initial state, the state after
// every step written to traj[step][point][:] -- consecutive lanes write consecutive points (coalesced).
template <class Lib>
__global__ __launch_bounds__(BLOCK) void odeint_traj_kernel(const float* __restrict__ x, long N, const float* __restrict__ xi,
                                                            const float* __restrict__ mask, int n_steps, float dt, int method,
                                                            float* __restrict__ traj) {
    constexpr int D = Lib::D;
    float w[D * Lib::P];
    load_xi<Lib>(xi, mask, 0, w);
    for (long n = (long)blockIdx.x * BLOCK + threadIdx.x; n < N; n += (long)gridDim.x * BLOCK) {
        float xp[D];
        load_point<D>(x, n, xp);
        integrate_steps<Lib>(w, xp, n_steps, dt, method, [&](int s, const float (&cur)[D]) {
            store_point<D>(traj + (long)s * N * D, n, cur);
        });
    }
}

template <class Lib>
__global__ __launch_bounds__(BLOCK) void odeint_kernel(const float* __restrict__ x, long N, bool vec,
                                                       const float* __restrict__ xi, const float* __restrict__ mask,
                                                       int n_steps, float dt, int method, float* __restrict__ out) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, PPT = Chunk<D>::PPT;
    float w[D * Lib::P];
    load_xi<Lib>(xi, mask, 0, w);
    struct Ops {
        float x[PPT][D];
    };
    for_each_chunk2<D, BLOCK, Ops>(
        N, vec, [&](long c, Ops& o) { load_chunk<D>(x, c, o.x); },
        [&](long c, Ops& o) {
            each_point<PPT>([&](auto i) { integrate<Lib>(w, o.x[i], n_steps, dt, method); });
            store_chunk_nt<D>(out, c, o.x);
        },
        [&](long n) {
            float xp[D];
            load_point<D>(x, n, xp);
            integrate<Lib>(w, xp, n_steps, dt, method);
            store_point<D>(out, n, xp);
        });
}

// ---------------------------------------------------------------------------------------
// Reduction epilogue shared by every "scalar loss + (d,p) gradient" kernel.
//   each block leaves NACC fp64 partial sums in part[(s*G + b)*NACC + k];
//   ONE launch (default): the block whose ticket comes last adds the G rows in fixed order, scales, masks,
//       rounds to fp32 and resets the ticket -- "last block done", no second kernel;
//   two launches (SYMODE_FUSED_FINALIZE=0, and the row-per-wave kernel): finalize_kernel does the same sums.
// Both orders of addition are identical, so the two paths are bit-identical (tests/test_gpu_kernels.py).
//
// Cross-workgroup hand-off without fences (MI355X_MICROARCH.md, "Valid forms", table row 1): the partial row is
// written with agent-scope (sc1, write-through) stores, every storing wave drains them (s_waitcnt vmcnt(0)), a
// workgroup barrier, then ONE lane adds to the problem's ticket; the block that reads G-1 back loads the rows with
// agent-scope (sc1, L1-bypassing) loads after a barrier its adding wave joins.  The ticket lives in the header of the
// caller's workspace (zeroed once by symode_workspace_init, self-resetting afterwards); a workspace that was never
// initialised is recognised by its magic word and answered with NaN outputs instead of stale ones.
// ---------------------------------------------------------------------------------------
constexpr unsigned long long WS_MAGIC = 0x53594d4f44453032ull;      // "SYMODE02"
constexpr long WS_MAX_PROBLEMS = 65536;                              // tickets in the header (grid.y limit is 65535)
constexpr long WS_HEADER_DOUBLES = 8 + WS_MAX_PROBLEMS / 2;          // [magic, 7 reserved | uint32 tickets] in front of the partials

struct Finish {
    unsigned long long* header;      // workspace base: header[0] = magic, tickets behind it
    const float* mask;               // (S, NACC-1) or null
    float loss_scale, grad_scale;
    float* loss;                     // (S) or null
    float* grad;                     // (S, NACC - n_loss)
    int fused;                       // 1: last-block finalisation inside this launch
    int n_loss;                      // leading scalar sums per problem (1; 2 for the fused MSE + regulariser closure)
};

__device__ __forceinline__ unsigned* ws_tickets(unsigned long long* header) {
    return reinterpret_cast<unsigned*>(header + 8);
}

// Sum the G partial rows of problem s in fixed order and write the outputs: thread (part, lane) adds rows
// part, part + 4, ... of value k = k0 + lane; the 4 parts are combined in fixed order through LDS.
// SC1: rows come from other workgroups of the SAME launch (agent-scope loads), else plain loads.
template <bool SC1>
__device__ __forceinline__ void combine_rows(const double* __restrict__ src, int G, int nacc, int n_loss, long s,
                                             const float* __restrict__ mask, float loss_scale, float grad_scale,
                                             float* __restrict__ loss, float* __restrict__ grad, double* comb) {
    auto ld = [&](long i) -> double {
        if constexpr (SC1)
            return __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            return src[i];
    };
    const int lane = threadIdx.x & (WAVE - 1), part = threadIdx.x / WAVE;
    for (int k0 = 0; k0 < nacc; k0 += WAVE) {
        const int k = k0 + lane;
        double v = 0.0;
        if (k < nacc) {
            constexpr int NP_ = BLOCK / WAVE, U = 16;
            int g = part;
            for (; g + (U - 1) * NP_ < G; g += U * NP_) {      // 16 independent loads in flight, added in order
                double t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = ld((long)(g + u * NP_) * nacc + k);
#pragma unroll
                for (int u = 0; u < U; ++u) v += t[u];
            }
            for (; g < G; g += NP_) v += ld((long)g * nacc + k);
        }
        if (k0 > 0) __syncthreads();
        comb[threadIdx.x] = v;
        __syncthreads();
        if (part == 0 && k < nacc) {
            v = comb[lane] + comb[WAVE + lane] + comb[2 * WAVE + lane] + comb[3 * WAVE + lane];
            if (k < n_loss) {
                if (loss != nullptr) loss[s * n_loss + k] = (float)(v * (double)loss_scale);
            } else {
                const long i = s * (nacc - n_loss) + (k - n_loss);
                const float m = mask ? mask[i] : 1.0f;
                grad[i] = (float)(v * (double)grad_scale) * m;
            }
        }
    }
}

// `lds`: the block reduction's staging area, reduce_lds_floats(BLOCK) floats.  The second form lets a kernel that holds
// a large LDS region of its own during the point loop (the Euler reverse sweep's state column) hand that region over
// instead of adding 17 KB per workgroup to it; the caller's threads are past their last access to it (barrier inside).
template <int NACC>
__device__ __forceinline__ void emit_partials_in(float (&acc)[NACC], double* __restrict__ part, const Finish& fin, float* lds);

template <int NACC>
__device__ __forceinline__ void emit_partials(float (&acc)[NACC], double* __restrict__ part, const Finish& fin) {
    __shared__ float lds[reduce_lds_floats(BLOCK)];
    emit_partials_in<NACC>(acc, part, fin, lds);
}

template <int NACC>
__device__ __forceinline__ void emit_partials_in(float (&acc)[NACC], double* __restrict__ part, const Finish& fin, float* lds) {
    __shared__ unsigned last_flag;
    const long s = blockIdx.y;
    const int G = gridDim.x;
    double* dst = part + (s * G + blockIdx.x) * NACC;
    if (!fin.fused) {
        block_reduce_emit_lds<NACC, BLOCK>(acc, lds, [&](int k, double v) { dst[k] = v; });
        return;
    }
    if (fin.header[0] != WS_MAGIC) {                     // workspace never initialised: fail loudly, touch no ticket
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (fin.loss != nullptr) fin.loss[s * fin.n_loss] = __builtin_nanf("");
            fin.grad[s * (NACC - fin.n_loss)] = __builtin_nanf("");
        }
        return;
    }
    unsigned* ticket = ws_tickets(fin.header) + s;
    double* comb = reinterpret_cast<double*>(lds);       // BLOCK doubles fit in the staging area
    static_assert(sizeof(float) * reduce_lds_floats(BLOCK) >= sizeof(double) * BLOCK, "LDS too small for the combine");
    if (fin.fused == 2) {
        // SYMODE_FUSED_FINALIZE=2: the hand-off in the memory model's own terms (MI355X_MICROARCH.md, Valid forms, first
        // bullet) -- plain stores, every wave drained, barrier, ONE agent-scope release by the signalling lane before its
        // ticket add; the last block makes ONE agent-scope acquire before anybody reads the rows with plain loads.
        // ~3 us per launch dearer than the sc1 form below (a write-back and an invalidate on the critical path of the last
        // block), which is why it is not the default; same sums in the same order, so the same bits (tested).
        block_reduce_emit_lds<NACC, BLOCK>(acc, lds, [&](int k, double v) { dst[k] = v; });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (t == (unsigned)(G - 1)) ? 1u : 0u;
            if (last_flag != 0u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (last_flag == 0u) return;
        combine_rows<false>(part + s * (long)G * NACC, G, NACC, fin.n_loss, s, fin.mask, fin.loss_scale, fin.grad_scale, fin.loss,
                            fin.grad, comb);
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    block_reduce_emit_lds<NACC, BLOCK>(acc, lds, [&](int k, double v) {
        __hip_atomic_store(dst + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave has its row out before the ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (t == (unsigned)(G - 1)) ? 1u : 0u;
    }
    __syncthreads();
    if (last_flag == 0u) return;
    combine_rows<true>(part + s * (long)G * NACC, G, NACC, fin.n_loss, s, fin.mask, fin.loss_scale, fin.grad_scale, fin.loss,
                       fin.grad, comb);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

// out[0] = loss_scale * sum_0 ; grad[k-1] = grad_scale * sum_k * mask[k-1]
// (a template only so that the header-defined kernel has vague linkage across the per-D TUs)
template <int TAG = 0>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const double* __restrict__ ws, int G, int nacc, int n_loss,
                                                         const float* __restrict__ mask, float loss_scale,
                                                         float grad_scale, float* __restrict__ loss,
                                                         float* __restrict__ grad) {
    __shared__ double comb[BLOCK];
    const long s = blockIdx.x;
    combine_rows<false>(ws + s * (long)G * nacc, G, nacc, n_loss, s, mask, loss_scale, grad_scale, loss, grad, comb);
}

template <int TAG = 0>
__global__ __launch_bounds__(BLOCK) void workspace_init_kernel(unsigned long long* header, long n_words) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_words) header[i] = (i == 0) ? WS_MAGIC : 0ull;
}

inline int fused_finalize_mode() { return (int)knobs().fused_finalize; }      // 0 two launches, 1 sc1 hand-off (default), 2 fenced hand-off

// ---------------------------------------------------------------------------------------
// K1: fused Theta + residual + MSE + gradient                  (closure body + backward)
// ---------------------------------------------------------------------------------------
// Streaming schedule: a register ring of chunk slots, each refilled right after use (points.hpp, chunk_ring).
//   RING4 = true   the D = 2 libraries up to 64 coefficients -- every task the reference ships: four chunks of x and dx in
//                  flight per lane (6-8 KB per wave); one big problem gives every workgroup its own contiguous slab;
//   RING4 = false  everything else: two chunks in flight (D = 3: coalesced tile loads through the wave's LDS slab).
// Round 1 measured the alternatives on MI355X (profiles/r01_ab_variants.txt; S = 2048 x 125 000 points, d = 2, algorithmic
// bytes per launch): plain grid-stride loop 4.8 / 5.6 TB/s at order 5 / 3, two chunks per step with non-temporal loads
// 5.4 / 6.35, a register double buffer 5.55 / 6.45, this ring 5.65 / 6.45; a packed-fp32 form (v_pk_fma_f32 over the two
// equation rows: same roundings, half the instructions) and an LDS-DMA ring measured within 1 % of it -- gfx950 retires a
// v_pk_fma_f32 in the time of two v_fma_f32 (profiles/r03_issue_probe.txt) -- and were removed in round 3.
// With the masked Xi in VGPRs the order-5 kernel needs 166 VGPRs (3 waves/SIMD); handing Xi to SGPRs (126 VGPRs,
// 4 waves) measured 1.5 % slower -- an SGPR operand costs issue time -- and grid widths 4096-16384 measured the same.
// Why order 5 stops there: 108 VALU ops per point = 422 K wave-instructions per SIMD per launch, and a SIMD with 3
// resident waves retires one every 1.23 ns (tools/micro/valu_rate.hip), i.e. 0.52 ms of VALU beside 0.52-0.64 ms of
// HBM stream in a 0.73 ms launch: both pipes are > 70 % busy.  Forms with sched_barrier between points were slower or
// equal; forcing 5-6 waves/SIMD (amdgpu_waves_per_eu) spills at order 5 (8-11x slower) and 4 waves is within 1.3 % of
// the free allocation; -fno-slp-vectorize (Makefile) is worth 6 % at order 5: SLP-packed FMAs force the uniform
// coefficients out of SGPRs into VGPR pairs.
template <class Lib, bool RING4>
__device__ __forceinline__ void loss_grad_body(const float* __restrict__ x, const float* __restrict__ dx, long N, bool vec,
                                               const float* __restrict__ xi, const float* __restrict__ mask,
                                               double* __restrict__ ws, const Finish& fin, const bool SEGMENTED) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, PPT = Chunk<D>::PPT, NV = Chunk<D>::NV, NACC = 1 + D * P;
    const long s = blockIdx.y;
    const float* xs = x + s * N * D;
    const float* ys = dx + s * N * D;
    float w[D * P];
    load_xi<Lib>(xi, mask, s, w);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;

    auto one = [&](const float (&xp)[D], const float (&yp)[D]) {
        float th[P], r[D];
        Lib::eval(xp, th);
        apply_xi<Lib>(w, th, r);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            r[j] -= yp[j];
            acc[0] = fmaf(r[j], r[j], acc[0]);
        }
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < P; ++k) acc[1 + j * P + k] = fmaf(r[j], th[k], acc[1 + j * P + k]);
    };
    auto point = [&](long n) {
        float xp[D], yp[D];
        load_point<D>(xs, n, xp);
        load_point<D>(ys, n, yp);
        one(xp, yp);
    };
    if constexpr (RING4) {
        const long tid = (long)blockIdx.x * BLOCK + threadIdx.x;
        if (vec) {
            const long nchunks_all = N / PPT;
            long nchunks = nchunks_all, c0 = tid, stride = (long)gridDim.x * BLOCK;
            if (SEGMENTED) {
                // every workgroup streams its own contiguous slab of the problem (as the batched launches do)
                const long per = (nchunks_all + gridDim.x - 1) / gridDim.x;
                const long lo = (long)blockIdx.x * per;
                nchunks = lo + per < nchunks_all ? lo + per : nchunks_all;
                c0 = lo + threadIdx.x;
                stride = BLOCK;
            }
            chunk_ring<4, 2 * NV>(
                nchunks, c0, stride,
                [&](long q, float4 (&slot)[2 * NV]) {
                    float4 vx[NV], vy[NV];
                    load_chunk_raw<D, true>(xs, q, vx);
                    load_chunk_raw<D, true>(ys, q, vy);
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        slot[i] = vx[i];
                        slot[NV + i] = vy[i];
                    }
                },
                [&](long, const float4 (&slot)[2 * NV]) {
                    float4 vx[NV], vy[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        vx[i] = slot[i];
                        vy[i] = slot[NV + i];
                    }
                    float xp[PPT][D], yp[PPT][D];
                    unpack_chunk<D>(vx, xp);
                    unpack_chunk<D>(vy, yp);
                    each_point<PPT>([&](auto i) { one(xp[i], yp[i]); });
                });
            const long n = nchunks_all * PPT + tid;
            if (n < N) point(n);
        } else {
            for (long n = tid; n < N; n += (long)gridDim.x * BLOCK) point(n);
        }
    } else {
        const float* const arrs[2] = {xs, ys};
        for_each_chunk_ring<D, BLOCK, 2, 2>(
            N, vec, arrs, [&](long, float (&o)[2][PPT][D]) { each_point<PPT>([&](auto i) { one(o[0][i], o[1][i]); }); }, point);
    }
    emit_partials<NACC>(acc, ws, fin);
}

template <class Lib, bool RING4>
__global__ __launch_bounds__(BLOCK) void loss_grad_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                          long N, bool vec, const float* __restrict__ xi,
                                                          const float* __restrict__ mask, double* __restrict__ ws,
                                                          Finish fin, bool segmented) {
    loss_grad_body<Lib, RING4>(x, dx, N, vec, xi, mask, ws, fin, segmented);
}

// ---------------------------------------------------------------------------------------
// S1: linear-latent symmetry regulariser
//   u = Xi_m (J_Theta(z) L z) - L (Xi_m Theta(z));  loss = sum |u|^2 over points and generators;
//   dloss/dXi[j,k] = 2 sum ( u_j dth_k - (L^T u)_j th_k ).
// ---------------------------------------------------------------------------------------
template <class Lib, int R = 4, int XI_SGPR_FROM = VGPR_XI_MAX + 1, int MINW = 1>
__global__ __launch_bounds__(BLOCK, MINW) void symreg_linear_kernel(const float* __restrict__ z, long N, bool vec,
                                                              const float* __restrict__ xi,
                                                              const float* __restrict__ mask,
                                                              const float* __restrict__ Lg, int n_gen,
                                                              double* __restrict__ ws, Finish fin) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, PPT = Chunk<D>::PPT, NACC = 1 + D * P;
    float w[D * P];
    load_xi<Lib, XI_SGPR_FROM>(xi, mask, 0, w);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;

    // the first generator is read ONCE (uniform -> scalar registers): with one generator -- every fixed group of the
    // reference, a single learned channel -- no scalar load is left inside the point loop and the loop over the further
    // generators is one skipped uniform branch
    float L0[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) L0[a][b] = n_gen > 0 ? Lg[a * D + b] : 0.0f;
    auto one_gen = [&](const float (&zp)[D], const float (&L)[D][D]) {
        float v[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float t = 0.0f;
#pragma unroll
            for (int b = 0; b < D; ++b) t = fmaf(L[a][b], zp[b], t);
            v[a] = t;
        }
        float th[P], dth[P], h[D], jv[D], u[D], ltu[D];
        Lib::eval_jvp(zp, v, th, dth);
        apply_xi<Lib>(w, th, h);
        apply_xi<Lib>(w, dth, jv);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float t = jv[a];
#pragma unroll
            for (int b = 0; b < D; ++b) t = fmaf(-L[a][b], h[b], t);
            u[a] = t;
            acc[0] = fmaf(t, t, acc[0]);
        }
#pragma unroll
        for (int b = 0; b < D; ++b) {
            float t = 0.0f;
#pragma unroll
            for (int a = 0; a < D; ++a) t = fmaf(L[a][b], u[a], t);
            ltu[b] = t;
        }
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < P; ++k)
                acc[1 + j * P + k] = fmaf(u[j], dth[k], fmaf(-ltu[j], th[k], acc[1 + j * P + k]));
    };
    auto one = [&](const float (&zp)[D]) {
        if (n_gen > 0) one_gen(zp, L0);
        for (int g = 1; g < n_gen; ++g) {
            float L[D][D];
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) L[a][b] = Lg[(g * D + a) * D + b];
            one_gen(zp, L);
        }
    };
    // 8 bytes per point against ~115 vector instructions: the stream is thin, but a wave that waits out every miss
    // idles its SIMD -- four chunks per lane in flight (points.hpp, for_each_chunk_ring)
    const float* const arrs[1] = {z};
    for_each_chunk_ring<D, BLOCK, R, 1>(
        N, vec, arrs,
        [&](long, float (&zp)[1][PPT][D]) {
            each_point<PPT>([&](auto i) { one(zp[0][i]); });
        },
        [&](long n) {
            float zp[D];
            load_point<D>(z, n, zp);
            one(zp);
        });
    emit_partials<NACC>(acc, ws, fin);
}

// ---------------------------------------------------------------------------------------
// S4: reversed symmetry regulariser with precomputed (g(x), J_g(x))
//   u = J_g(x) h(x) - h(g(x));  loss = sum_g mean(u^2);
//   dloss/dXi[j,k] = (2/(N D)) sum ( (J_g^T u)_j th_k(x) - u_j th_k(g x) ).
// Batched like K1: problem s = blockIdx.y owns x[s] (N, D), gx[s] (n_g, N, D), jgx[s] (n_g, N, D, D), xi[s], mask[s].
// A pure stream (8 + n_g * 24 bytes per point at D = 2, each read once): every operand arrives as non-temporal
// 16-byte vectors -- a chunk of PPT points is one vector of x, one of g(x) and PPT*D*D/4 consecutive vectors of J_g --
// and a step issues the loads of two chunks before the arithmetic of either.
// ---------------------------------------------------------------------------------------
template <int D>
struct JChunk {
    static constexpr int NV = Chunk<D>::PPT * D * D / 4;              // dwordx4 per chunk of Jacobians
    static_assert(Chunk<D>::PPT * D * D == NV * 4, "Jacobian chunk must be whole 16-byte vectors");
};

// MSE = true: the whole closure of the reversed-regulariser runs in ONE pass -- the residual r = h(x) - dx shares
// Theta(x) and h(x) with the regulariser, x is read once (40 instead of 16 + 32 bytes per point at D = 2, n_g = 1):
//   sums[0] = sum r^2, sums[1] = sum_g sum u^2,  grad = d( sums[0] + w_sym sums[1] ) / dXi  (both under the same 1/(N D)).
template <class Lib, bool MSE, int RING = 2, int XI_SGPR_FROM = 32>
__global__ __launch_bounds__(BLOCK) void symreg_reversed_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                                const float* __restrict__ gx,
                                                                const float* __restrict__ jgx, int n_g, long N, bool vec,
                                                                const float* __restrict__ xi,
                                                                const float* __restrict__ mask, float w_sym,
                                                                double* __restrict__ ws, Finish fin) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, NL = MSE ? 2 : 1, NACC = NL + D * P, PPT = Chunk<D>::PPT, NV = Chunk<D>::NV,
                  NVJ = JChunk<D>::NV, SYM0 = NL - 1;
    const long s = blockIdx.y;
    const float* xs = x + s * N * D;
    const float* ys = MSE ? dx + s * N * D : nullptr;
    const float* gs = gx + s * (long)n_g * N * D;
    const float* js = jgx + s * (long)n_g * N * D * D;
    float w[D * P];
    // two libraries per point live here: Xi in SGPRs from d*p = 32 (3 waves/SIMD at order 5) -- and up to 80, the d = 3
    // order-3 libraries with sine / exp columns (69-78 coefficients would otherwise sit in VGPRs beside 70-79 sums)
    load_xi<Lib, XI_SGPR_FROM, 80>(xi, mask, s, w);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
    const float ws_ = MSE ? w_sym : 1.0f;

    // one point against one group element: th, h belong to x (shared by all group elements of the point);
    // `extra` (MSE form, first group element only) is the residual r, whose gradient rides on the same Theta(x) products
    auto one = [&](const float (&th)[P], const float (&h)[D], const float (&gp)[D], const float (&J)[D * D], const float (&extra)[D]) {
        float thg[P], hg[D], u[D], jtu[D];
        Lib::eval(gp, thg);
        apply_xi<Lib>(w, thg, hg);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float t = -hg[a];
#pragma unroll
            for (int b = 0; b < D; ++b) t = fmaf(J[a * D + b], h[b], t);
            u[a] = t;
            acc[SYM0] = fmaf(t, t, acc[SYM0]);
        }
#pragma unroll
        for (int b = 0; b < D; ++b) {
            float t = 0.0f;
#pragma unroll
            for (int a = 0; a < D; ++a) t = fmaf(J[a * D + b], u[a], t);
            jtu[b] = MSE ? fmaf(ws_, t, extra[b]) : t;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float uj = MSE ? ws_ * u[j] : u[j];
#pragma unroll
            for (int k = 0; k < P; ++k) acc[NL + j * P + k] = fmaf(jtu[j], th[k], fmaf(-uj, thg[k], acc[NL + j * P + k]));
        }
    };
    // residual of one point (MSE form): r = h - dx, sums[0] += r^2; returned for the first group element's `extra`
    auto resid = [&](const float (&h)[D], const float (&yp)[D], float (&r)[D]) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            r[j] = h[j] - yp[j];
            acc[0] = fmaf(r[j], r[j], acc[0]);
        }
    };
    auto load_j = [&](const float* base, long c, float4 (&v)[NVJ]) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v* q = reinterpret_cast<const f4v*>(base) + c * NVJ;
#pragma unroll
        for (int i = 0; i < NVJ; ++i) {
            const f4v t = __builtin_nontemporal_load(q + i);
            v[i] = make_float4(t.x, t.y, t.z, t.w);
        }
    };
    // the chunk's points against group element g, operands already in registers
    auto chunk_g = [&](float (&th)[PPT][P], float (&h)[PPT][D], const float4 (&vg)[NV], const float4 (&vj)[NVJ],
                       float (&extra)[PPT][D]) {
        float gp[PPT][D], jf[NVJ * 4], J[PPT][D * D];
        unpack_chunk<D>(vg, gp);
#pragma unroll
        for (int i = 0; i < NVJ; ++i) {
            jf[4 * i + 0] = vj[i].x;
            jf[4 * i + 1] = vj[i].y;
            jf[4 * i + 2] = vj[i].z;
            jf[4 * i + 3] = vj[i].w;
        }
#pragma unroll
        for (int e = 0; e < PPT * D * D; ++e) J[e / (D * D)][e % (D * D)] = jf[e];
        each_point<PPT>([&](auto i) { one(th[i], h[i], gp[i], J[i], extra[i]); });
    };
    auto point = [&](long n) {
        float xp[D], th[P], h[D], r[D], zero[D];
        load_point<D>(xs, n, xp);
        Lib::eval(xp, th);
        apply_xi<Lib>(w, th, h);
#pragma unroll
        for (int j = 0; j < D; ++j) r[j] = zero[j] = 0.0f;
        if constexpr (MSE) {
            float yp[D];
            load_point<D>(ys, n, yp);
            resid(h, yp, r);
        }
        for (int g = 0; g < n_g; ++g) {
            float gp[D], J[D * D];
            load_point<D>(gs + (long)g * N * D, n, gp);
            const float* Jp = js + ((long)g * N + n) * D * D;
#pragma unroll
            for (int e = 0; e < D * D; ++e) J[e] = Jp[e];
            if (g == 0)
                one(th, h, gp, J, r);
            else
                one(th, h, gp, J, zero);
        }
    };
    auto eval_x = [&](const float4 (&vx)[NV], float (&th)[PPT][P], float (&h)[PPT][D]) {
        float xp[PPT][D];
        unpack_chunk<D>(vx, xp);
        each_point<PPT>([&](auto i) {
            Lib::eval(xp[i], th[i]);
            apply_xi<Lib>(w, th[i], h[i]);
        });
    };

    const long tid = (long)blockIdx.x * BLOCK + threadIdx.x, nthreads = (long)gridDim.x * BLOCK;
    // one chunk: its points' Theta(x), h(x) are formed once and live only while this chunk's group elements are visited
    auto chunk_all = [&](long c, const float4 (&vx)[NV], const float4 (&vy)[NV], const float4 (&vg)[NV], const float4 (&vj)[NVJ]) {
        if constexpr (D == 3) {
            // four points per chunk: Theta(x), h(x) of ALL of them (4 P + 12 registers) do not fit beside the sums -- point by
            // point instead, Theta(x) re-evaluated for every further group element (same values, same order of every sum)
            float xp[PPT][D], yp[PPT][D];
            unpack_chunk<D>(vx, xp);
            if constexpr (MSE) unpack_chunk<D>(vy, yp);
            auto group = [&](const float4 (&ug)[NV], const float4 (&uj)[NVJ], bool first) {
                float gp[PPT][D], jf[NVJ * 4], J[PPT][D * D];
                unpack_chunk<D>(ug, gp);
#pragma unroll
                for (int i = 0; i < NVJ; ++i) {
                    jf[4 * i + 0] = uj[i].x;
                    jf[4 * i + 1] = uj[i].y;
                    jf[4 * i + 2] = uj[i].z;
                    jf[4 * i + 3] = uj[i].w;
                }
#pragma unroll
                for (int e = 0; e < PPT * D * D; ++e) J[e / (D * D)][e % (D * D)] = jf[e];
                auto body = [&](auto i) {
                    float th[P], h[D], r[D];
                    Lib::eval(xp[i], th);
                    apply_xi<Lib>(w, th, h);
#pragma unroll
                    for (int j = 0; j < D; ++j) r[j] = 0.0f;
                    if constexpr (MSE) {
                        if (first) resid(h, yp[i], r);
                    }
                    one(th, h, gp[i], J[i], r);
                };
                each_point<PPT>(body);
            };
            group(vg, vj, true);
            for (int g = 1; g < n_g; ++g) {
                float4 ng[NV], nj[NVJ];
                load_chunk_raw<D, true>(gs + (long)g * N * D, c, ng);
                load_j(js + (long)g * N * D * D, c, nj);
                group(ng, nj, false);
            }
            return;
        }
        float th[PPT][P], h[PPT][D], r[PPT][D], zero[PPT][D];
        eval_x(vx, th, h);
#pragma unroll
        for (int i = 0; i < PPT; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) r[i][j] = zero[i][j] = 0.0f;
        if constexpr (MSE) {
            float yp[PPT][D];
            unpack_chunk<D>(vy, yp);
            each_point<PPT>([&](auto i) { resid(h[i], yp[i], r[i]); });
        }
        chunk_g(th, h, vg, vj, r);
        for (int g = 1; g < n_g; ++g) {
            float4 ng[NV], nj[NVJ];
            load_chunk_raw<D, true>(gs + (long)g * N * D, c, ng);
            load_j(js + (long)g * N * D * D, c, nj);
            chunk_g(th, h, ng, nj, zero);
        }
    };
    if constexpr (D == 3) {
        // 12-byte points and 36-byte Jacobians: whole waves fetch their tiles coalesced and redistribute through a
        // wave-private LDS slab (points.hpp, exchange_tile); ragged waves and the tail keep the per-lane loads
        if (vec && n_g > 0) {
            __shared__ float4 slab3[BLOCK / WAVE][NVJ * WAVE];
            const int lane = threadIdx.x & (WAVE - 1);
            float4* slab = slab3[threadIdx.x / WAVE];
            const long nchunks = N / PPT;
            for (long c = tid;; c += nthreads) {
                const bool in = c < nchunks;
                const unsigned long long live = __builtin_amdgcn_ballot_w64(in);
                if (live == 0ull) break;
                float4 ax[NV], ay[NV], ag[NV], aj[NVJ];
                if (live == ~0ull) {
                    const long c0 = c - lane;
                    float4 tx[NV], ty[NV], tg[NV], tj[NVJ];
                    load_tile_raw<NV, true>(xs, c0, lane, tx);
                    if constexpr (MSE) load_tile_raw<NV, true>(ys, c0, lane, ty);
                    load_tile_raw<NV, true>(gs, c0, lane, tg);
                    load_tile_raw<NVJ, true>(js, c0, lane, tj);
                    exchange_tile<NV>(tx, ax, slab, lane);
                    if constexpr (MSE) exchange_tile<NV>(ty, ay, slab, lane);
                    exchange_tile<NV>(tg, ag, slab, lane);
                    exchange_tile<NVJ>(tj, aj, slab, lane);
                } else if (in) {
                    load_chunk_raw<D, true>(xs, c, ax);
                    if constexpr (MSE) load_chunk_raw<D, true>(ys, c, ay);
                    load_chunk_raw<D, true>(gs, c, ag);
                    load_j(js, c, aj);
                }
                if (in) chunk_all(c, ax, ay, ag, aj);
            }
            const long n = nchunks * PPT + tid;
            if (n < N) point(n);
        } else {
            for (long n = tid; n < N; n += nthreads) point(n);
        }
    } else if (vec && n_g > 0) {
        // register ring: RING chunks of x, (dx,) g(x), J_g of the first group element in flight per lane, a slot refilled
        // as soon as its chunk has been consumed (closure_ab: one 2^26-point problem 334 -> 325 us at order 3, the fused
        // closure at order 5 470 -> 442 us; the batched bench shape is VALU / HBM co-limited either way)
        constexpr int NVT = (MSE ? 3 : 2) * NV + NVJ, OY = NV, OG = (MSE ? 2 : 1) * NV, OJ = OG + NV;
        const long nchunks = N / PPT;
        chunk_ring<RING, NVT>(
            nchunks, tid, nthreads,
            [&](long q, float4 (&slot)[NVT]) {
                float4 tx[NV], tg[NV], tj[NVJ];
                load_chunk_raw<D, true>(xs, q, tx);
                load_chunk_raw<D, true>(gs, q, tg);
                load_j(js, q, tj);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    slot[i] = tx[i];
                    slot[OG + i] = tg[i];
                }
#pragma unroll
                for (int i = 0; i < NVJ; ++i) slot[OJ + i] = tj[i];
                if constexpr (MSE) {
                    float4 ty[NV];
                    load_chunk_raw<D, true>(ys, q, ty);
#pragma unroll
                    for (int i = 0; i < NV; ++i) slot[OY + i] = ty[i];
                }
            },
            [&](long c, const float4 (&slot)[NVT]) {
                float4 ax[NV], ay[NV], ag[NV], aj[NVJ];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    ax[i] = slot[i];
                    ay[i] = MSE ? slot[OY + i] : slot[i];
                    ag[i] = slot[OG + i];
                }
#pragma unroll
                for (int i = 0; i < NVJ; ++i) aj[i] = slot[OJ + i];
                chunk_all(c, ax, ay, ag, aj);
            });
        const long n = nchunks * PPT + tid;
        if (n < N) point(n);
    } else {
        for (long n = tid; n < N; n += nthreads) point(n);
    }
    emit_partials<NACC>(acc, ws, fin);
}

// ---------------------------------------------------------------------------------------
// vjp of forward: given g = dL/d(out) (N, D)
//   grad_x[n] = J_Theta(x_n)^T (Xi_m^T g_n)           (optional)
//   grad_xi   = (sum_n g_n Theta(x_n)^T) * mask        (through the partial-sum epilogue)
// ---------------------------------------------------------------------------------------
template <class Lib, bool GX, int R = 3>
__global__ __launch_bounds__(BLOCK) void vjp_kernel(const float* __restrict__ x, const float* __restrict__ g, long N, bool vec,
                                                    const float* __restrict__ xi, const float* __restrict__ mask,
                                                    float* __restrict__ grad_x, double* __restrict__ ws, Finish fin) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, NACC = 1 + D * P, PPT = Chunk<D>::PPT;
    float w[D * P];
    load_xi<Lib>(xi, mask, 0, w);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
    auto one = [&](const float (&xp)[D], const float (&gp)[D], float (&bx)[D]) {
        float th[P];
        Lib::eval(xp, th);
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < P; ++k) acc[1 + j * P + k] = fmaf(gp[j], th[k], acc[1 + j * P + k]);
        if constexpr (GX) {
            float bar[P];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < D; ++j) t = fmaf(gp[j], w[j * P + k], t);
                bar[k] = t;
            }
            Lib::vjp(xp, th, bar, bx);
        }
    };
    // a reduction launch: few long-lived workgroups, so the wave's own run-ahead hides the latency (three chunks of x
    // and g in flight per lane; the grad_x stores of a chunk sit in the same in-order queue as the loads behind them)
    const float* const arrs[2] = {x, g};
    for_each_chunk_ring<D, BLOCK, R, 2>(
        N, vec, arrs,
        [&](long c, float (&o)[2][PPT][D]) {
            float bx[PPT][D];
            each_point<PPT>([&](auto i) { one(o[0][i], o[1][i], bx[i]); });
            if constexpr (GX) store_chunk_nt<D>(grad_x, c, bx);
        },
        [&](long n) {
            float xp[D], gp[D], bx[D];
            load_point<D>(x, n, xp);
            load_point<D>(g, n, gp);
            one(xp, gp, bx);
            if constexpr (GX) store_point<D>(grad_x, n, bx);
        });
    emit_partials<NACC>(acc, ws, fin);
}

// out = Theta(x) Xi_m^T and jv = (J_Theta(x) v) Xi_m^T in one pass (forward-mode tangent).
template <class Lib>
__global__ __launch_bounds__(BLOCK) void forward_jvp_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                            long N, bool vec, const float* __restrict__ xi,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            float* __restrict__ jv) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, PPT = Chunk<D>::PPT;
    float w[D * P];
    load_xi<Lib>(xi, mask, 0, w);
    auto one = [&](const float (&xp)[D], const float (&vp)[D], float (&h)[D], float (&t)[D]) {
        float th[P], dth[P];
        Lib::eval_jvp(xp, vp, th, dth);
        apply_xi<Lib>(w, th, h);
        apply_xi<Lib>(w, dth, t);
    };
    struct Ops {
        float x[PPT][D], v[PPT][D];
    };
    for_each_chunk2<D, BLOCK, Ops>(
        N, vec,
        [&](long c, Ops& o) {
            load_chunk<D>(x, c, o.x);
            load_chunk<D>(v, c, o.v);
        },
        [&](long c, Ops& o) {
            float h[PPT][D], t[PPT][D];
            each_point<PPT>([&](auto i) { one(o.x[i], o.v[i], h[i], t[i]); });
            if (out != nullptr) store_chunk_nt<D>(out, c, h);
            store_chunk_nt<D>(jv, c, t);
        },
        [&](long n) {
            float xp[D], vp[D], h[D], t[D];
            load_point<D>(x, n, xp);
            load_point<D>(v, n, vp);
            one(xp, vp, h, t);
            if (out != nullptr) store_point<D>(out, n, h);
            store_point<D>(jv, n, t);
        });
}

// Reverse mode of forward_jvp_kernel: upstream g_out on out (may be null) and g_jv on jv.
template <class Lib, bool GO = true, int R = 3>
__global__ __launch_bounds__(BLOCK) void jvp_vjp_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                        const float* __restrict__ g_out,
                                                        const float* __restrict__ g_jv, long N, bool vec,
                                                        const float* __restrict__ xi, const float* __restrict__ mask,
                                                        float* __restrict__ grad_x, float* __restrict__ grad_v,
                                                        double* __restrict__ ws, Finish fin) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, NACC = 1 + D * P, PPT = Chunk<D>::PPT;
    float w[D * P];
    load_xi<Lib, VGPR_XI_MAX + 1, 80>(xi, mask, 0, w);   // (th, dth, bar, dbar live per point: Xi leaves the vector file up to 80 coefficients)
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
    auto one = [&](const float (&xp)[D], const float (&vp)[D], const float (&go)[D], const float (&gt)[D], float (&bx)[D],
                   float (&bv)[D]) {
        float th[P], dth[P], bar[P], dbar[P];
        Lib::eval_jvp(xp, vp, th, dth);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            float a = 0.0f, c = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                a = fmaf(go[j], w[j * P + k], a);
                c = fmaf(gt[j], w[j * P + k], c);
                acc[1 + j * P + k] = fmaf(go[j], th[k], fmaf(gt[j], dth[k], acc[1 + j * P + k]));
            }
            bar[k] = a;
            dbar[k] = c;
        }
        Lib::vjp_of_jvp(xp, vp, th, dth, bar, dbar, bx, bv);
    };
    // GO: an upstream gradient on `out` exists (g_out != nullptr); else its chunk is not loaded at all
    constexpr int NA = GO ? 4 : 3;
    auto chunk = [&](long c, float (&o)[NA][PPT][D]) {
        float bx[PPT][D], bv[PPT][D], zero[D];
#pragma unroll
        for (int j = 0; j < D; ++j) zero[j] = 0.0f;
        auto body = [&](auto i) {
            if constexpr (GO)
                one(o[0][i], o[1][i], o[NA - 1][i], o[2][i], bx[i], bv[i]);
            else
                one(o[0][i], o[1][i], zero, o[2][i], bx[i], bv[i]);
        };
        each_point<PPT>(body);
        store_chunk_nt<D>(grad_x, c, bx);
        store_chunk_nt<D>(grad_v, c, bv);
    };
    auto point = [&](long n) {
        float xp[D], vp[D], go[D], gt[D], bx[D], bv[D];
        load_point<D>(x, n, xp);
        load_point<D>(v, n, vp);
        load_point<D>(g_jv, n, gt);
        if constexpr (GO) {
            load_point<D>(g_out, n, go);
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) go[j] = 0.0f;
        }
        one(xp, vp, go, gt, bx, bv);
        store_point<D>(grad_x, n, bx);
        store_point<D>(grad_v, n, bv);
    };
    if constexpr (GO) {
        const float* const arrs[4] = {x, v, g_jv, g_out};
        for_each_chunk_ring<D, BLOCK, R, 4>(N, vec, arrs, chunk, point);
    } else {
        const float* const arrs[3] = {x, v, g_jv};
        for_each_chunk_ring<D, BLOCK, R, 3>(N, vec, arrs, chunk, point);
    }
    emit_partials<NACC>(acc, ws, fin);
}

// ---------------------------------------------------------------------------------------
// Offline data generation (reference data_utils/ode.py:7-28, 45-48): fixed-step RK4 orbits of
// dx/dt = Theta(x) Xi^T in fp64, one trajectory per thread, every `subsample`-th state and its exact
// derivative written as fp32 (n_traj, n_out, D).  Not part of the timed path.
// ---------------------------------------------------------------------------------------
template <class Lib>
__global__ __launch_bounds__(BLOCK) void rk4_traj_kernel(const double* __restrict__ x0, long n_traj,
                                                         const double* __restrict__ xi, int n_steps, double dt,
                                                         int subsample, float* __restrict__ x_out,
                                                         float* __restrict__ dx_out) {
    constexpr int D = Lib::D, P = Lib::P;
    const long tr = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (tr >= n_traj) return;
    double w[D * P];
#pragma unroll
    for (int i = 0; i < D * P; ++i) w[i] = xi[i];
    auto f = [&](const double (&y)[D], double (&h)[D]) {
        double th[P];
        Lib::eval_f64(y, th);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k) s = fma(w[j * P + k], th[k], s);
            h[j] = s;
        }
    };
    double x[D];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = x0[tr * D + j];
    const long n_out = (n_steps + subsample - 1) / subsample;
    for (int i = 0; i < n_steps; ++i) {
        double d1[D], d2[D], d3[D], d4[D], y[D];
        f(x, d1);
        if (i % subsample == 0) {
            const long o = (tr * n_out + i / subsample) * D;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                x_out[o + j] = (float)x[j];
                dx_out[o + j] = (float)d1[j];
            }
        }
        if (i == n_steps - 1) break;
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = x[j] + 0.5 * (dt * d1[j]);
        f(y, d2);
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = x[j] + 0.5 * (dt * d2[j]);
        f(y, d3);
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = x[j] + dt * d3[j];
        f(y, d4);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = x[j] + (dt * d1[j] + 2 * (dt * d2[j]) + 2 * (dt * d3[j]) + dt * d4[j]) / 6;
    }
}

// ---------------------------------------------------------------------------------------
// Fused K-step Euler flow with its tangent map (the `f` of the infinitesimal regulariser S2):
//   x_{k+1} = x_k + dt h(x_k),   t_{k+1} = t_k + dt J_h(x_k) t_k,   h = Theta(.) Xi_m^T
// forward: (x_K, t_K) = (f(x_0), J_f(x_0) t_0) with all K steps in registers;
// reverse: adjoints (a_x, a_t) walk the steps backwards; step k's state is recomputed from (x_0, t_0)
//          (K(K-1)/2 cheap library evaluations instead of K state vectors per thread).
// ---------------------------------------------------------------------------------------
template <class Lib>
__device__ __forceinline__ void euler_tangent_steps(const float (&w)[Lib::D * Lib::P], float (&x)[Lib::D],
                                                    float (&t)[Lib::D], int n_steps, float dt) {
    constexpr int D = Lib::D, P = Lib::P;
    for (int s = 0; s < n_steps; ++s) {
        float th[P], dth[P], h[D], jt[D];
        Lib::eval_jvp(x, t, th, dth);
        apply_xi<Lib>(w, th, h);
        apply_xi<Lib>(w, dth, jt);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            x[j] = x[j] + dt * h[j];
            t[j] = t[j] + dt * jt[j];
        }
    }
}

template <class Lib>
__global__ __launch_bounds__(BLOCK) void euler_jvp_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                          long N, bool vec, const float* __restrict__ xi,
                                                          const float* __restrict__ mask, int n_steps, float dt,
                                                          float* __restrict__ x_out, float* __restrict__ t_out) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, PPT = Chunk<D>::PPT;
    float w[D * Lib::P];
    load_xi<Lib>(xi, mask, 0, w);
    struct Ops {
        float x[PPT][D], t[PPT][D];
    };
    for_each_chunk2<D, BLOCK, Ops>(
        N, vec,
        [&](long c, Ops& o) {
            load_chunk<D>(x, c, o.x);
            load_chunk<D>(v, c, o.t);
        },
        [&](long c, Ops& o) {
            each_point<PPT>([&](auto i) { euler_tangent_steps<Lib>(w, o.x[i], o.t[i], n_steps, dt); });
            store_chunk_nt<D>(x_out, c, o.x);
            store_chunk_nt<D>(t_out, c, o.t);
        },
        [&](long n) {
            float xp[D], tp[D];
            load_point<D>(x, n, xp);
            load_point<D>(v, n, tp);
            euler_tangent_steps<Lib>(w, xp, tp, n_steps, dt);
            store_point<D>(x_out, n, xp);
            store_point<D>(t_out, n, tp);
        });
}

// Reverse sweep.  STACK = true: the forward pass parks the state (x_k, t_k) entering every step in a per-thread LDS
// column (stack[(k * 2D + i) * BLOCK + tid]: consecutive lanes on consecutive banks), so the reverse pass is K library
// evaluations; the launcher picks it while K * 2D * BLOCK floats fit beside the reduction's staging area (K <= 16 at
// D = 2).  STACK = false: state k is recomputed from (x_0, t_0) -- K(K-1)/2 extra evaluations, no storage, any K.
template <class Lib, bool STACK>
__global__ __launch_bounds__(BLOCK) void euler_jvp_vjp_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                              const float* __restrict__ g_x,
                                                              const float* __restrict__ g_t, long N, bool vec,
                                                              const float* __restrict__ xi,
                                                              const float* __restrict__ mask, int n_steps, float dt,
                                                              float* __restrict__ grad_x, float* __restrict__ grad_v,
                                                              double* __restrict__ ws, Finish fin) {
    vec = vec && chunked_stream<Lib>;            // (D = 3 sine / exp libraries: point by point, see chunked_stream)
    constexpr int D = Lib::D, P = Lib::P, NACC = 1 + D * P, PPT = Chunk<D>::PPT;
    extern __shared__ float stack[];
    float w[D * P];
    load_xi<Lib, VGPR_XI_MAX + 1, 80>(xi, mask, 0, w);   // (th, dth, bar, dbar live per point: Xi leaves the vector file up to 80 coefficients)
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
    const int tid = threadIdx.x;

    // adjoints (ax, at) of (x_K, t_K) in, of (x_0, t_0) out
    auto one = [&](const float (&x0)[D], const float (&t0)[D], float (&ax)[D], float (&at)[D]) {
        if constexpr (STACK) {
            float xk[D], tk[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                xk[j] = x0[j];
                tk[j] = t0[j];
            }
            for (int k = 0; k < n_steps; ++k) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    stack[(k * 2 * D + j) * BLOCK + tid] = xk[j];
                    stack[(k * 2 * D + D + j) * BLOCK + tid] = tk[j];
                }
                euler_tangent_steps<Lib>(w, xk, tk, 1, dt);
            }
        }
        for (int k = n_steps - 1; k >= 0; --k) {
            float xk[D], tk[D];
            if constexpr (STACK) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    xk[j] = stack[(k * 2 * D + j) * BLOCK + tid];
                    tk[j] = stack[(k * 2 * D + D + j) * BLOCK + tid];
                }
            } else {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    xk[j] = x0[j];
                    tk[j] = t0[j];
                }
                euler_tangent_steps<Lib>(w, xk, tk, k, dt);               // state entering step k
            }
            float th[P], dth[P], bar[P], dbar[P], bx[D], bv[D];
            Lib::eval_jvp(xk, tk, th, dth);
#pragma unroll
            for (int q = 0; q < P; ++q) {
                float a = 0.0f, c = 0.0f;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float go = dt * ax[j], gt = dt * at[j];   // adjoints on h(x_k) and on J_h(x_k) t_k
                    a = fmaf(go, w[j * P + q], a);
                    c = fmaf(gt, w[j * P + q], c);
                    acc[1 + j * P + q] = fmaf(go, th[q], fmaf(gt, dth[q], acc[1 + j * P + q]));
                }
                bar[q] = a;
                dbar[q] = c;
            }
            Lib::vjp_of_jvp(xk, tk, th, dth, bar, dbar, bx, bv);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                ax[j] += bx[j];
                at[j] += bv[j];
            }
        }
    };
    struct Ops {
        float x[PPT][D], t[PPT][D], ax[PPT][D], at[PPT][D];
    };
    for_each_chunk2<D, BLOCK, Ops>(
        N, vec,
        [&](long c, Ops& o) {
            load_chunk<D>(x, c, o.x);
            load_chunk<D>(v, c, o.t);
            load_chunk<D>(g_x, c, o.ax);
            load_chunk<D>(g_t, c, o.at);
        },
        [&](long c, Ops& o) {
            each_point<PPT>([&](auto i) { one(o.x[i], o.t[i], o.ax[i], o.at[i]); });
            store_chunk_nt<D>(grad_x, c, o.ax);
            store_chunk_nt<D>(grad_v, c, o.at);
        },
        [&](long n) {
            float x0[D], t0[D], ax[D], at[D];
            load_point<D>(x, n, x0);
            load_point<D>(v, n, t0);
            load_point<D>(g_x, n, ax);
            load_point<D>(g_t, n, at);
            one(x0, t0, ax, at);
            store_point<D>(grad_x, n, ax);
            store_point<D>(grad_v, n, at);
        });
    if constexpr (STACK) {
        // the state column is dead now: the reduction stages in it (the launcher sizes the region for both), which is
        // what lets a K = 10 workgroup fit four to a CU (40 KB instead of 40 + 17)
        __syncthreads();
        emit_partials_in<NACC>(acc, ws, fin, stack);
    } else {
        emit_partials<NACC>(acc, ws, fin);
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
#define SYMODE_LAUNCH_CHECK() \
    do {                      \
        hipError_t e_ = hipGetLastError(); \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

template <class Lib>
hipError_t launch_theta(const float* x, long n, float* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int g = map_grid_for(n, 1);
    theta_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x, n, ((uintptr_t)out % 16) == 0, out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

template <class Lib>
hipError_t launch_forward(const float* x, long n, const float* xi, const float* mask, float* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int g = map_grid_for(n, Chunk<Lib::D>::PPT * map_rounds<Lib>);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(out, n, Lib::D, 1);
    forward_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x, n, vec, xi, mask, out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

template <class Lib>
hipError_t launch_odeint_traj(const float* x, long n, const float* xi, const float* mask, int n_steps, float dt, int method,
                              float* traj, hipStream_t st) {
    long g = (n + BLOCK - 1) / BLOCK;
    if (g > 4096) g = 4096;
    odeint_traj_kernel<Lib><<<dim3((unsigned)g), dim3(BLOCK), 0, st>>>(x, n, xi, mask, n_steps, dt, method, traj);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

template <class Lib>
hipError_t launch_odeint(const float* x, long n, const float* xi, const float* mask, int n_steps, float dt, int method,
                         float* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int g = map_grid_for(n, Chunk<Lib::D>::PPT);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(out, n, Lib::D, 1);
    odeint_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x, n, vec, xi, mask, n_steps, dt, method, out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}


// ---------------------------------------------------------------------------------------
// K1 for large libraries (D * P > SGPR_XI_MAX: d = 3 order 4, d = 4 order 3, sine/exp variants): one ROW per wave.
// The thread-per-point form needs 1 + D*P accumulators and D*P coefficients in VGPRs per lane there (390-430
// registers: one wave per SIMD, 1.5-1.8 TB/s).  Here a workgroup is D waves over the same 64 points per step; wave j
// owns row j of Xi -- P coefficients in SGPRs, P + 1 accumulators -- evaluates Theta itself (34 redundant multiplies
// per wave at p = 35 against 70 fused multiply-adds of its own) and reads only component j of dx.  ~100 VGPRs, 4-5
// waves per SIMD; the D waves hit the same lines, so HBM sees every point once.
// Partials land in the same ws[(s*G + b)*NACC + k] layout as the thread-per-point form, finalize_kernel is shared.
// ---------------------------------------------------------------------------------------
template <class Lib>
__global__ __launch_bounds__(Lib::D* WAVE) void loss_grad_rows_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                                      long N, const float* __restrict__ xi,
                                                                      const float* __restrict__ mask, double* __restrict__ ws) {
    constexpr int D = Lib::D, P = Lib::P, NACC = 1 + D * P;
    __shared__ double loss_part[D];
    const long s = blockIdx.y;
    const int row = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const float* xs = x + s * N * D;
    const float* ys = dx + s * N * D;
    float w[P];                                                     // row `row` of Xi * mask, wave-uniform -> SGPRs
    {
        const float* a = xi + (s * D + row) * P;
#pragma unroll
        for (int k = 0; k < P; ++k) w[k] = a[k];
        if (mask != nullptr) {
            const float* m = mask + (s * D + row) * P;
#pragma unroll
            for (int k = 0; k < P; ++k) w[k] *= m[k];
        }
#if SYMODE_ROWS_XI_SGPR
#pragma unroll
        for (int k = 0; k < P; ++k) w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[k])));
#endif
    }
    float acc[P], sq = 0.0f;
#pragma unroll
    for (int k = 0; k < P; ++k) acc[k] = 0.0f;
    // one point per lane per step, the next step's operands requested before this step's ~3p VALU ops (clamped index:
    // the last prefetch re-reads an in-bounds point)
    const long stride = (long)gridDim.x * WAVE;
    long n = (long)blockIdx.x * WAVE + lane;
    float xn[D], yn = 0.0f;
    {
        const long q = n < N ? n : N - 1;
        load_point<D>(xs, q, xn);
        yn = ys[q * D + row];
    }
    for (; n < N; n += stride) {
        float xp[D], th[P];
#pragma unroll
        for (int i = 0; i < D; ++i) xp[i] = xn[i];
        const float y = yn;
        {
            const long q = n + stride < N ? n + stride : N - 1;
            load_point<D>(xs, q, xn);
            yn = ys[q * D + row];
        }
        Lib::eval(xp, th);
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < P; ++k) r = fmaf(w[k], th[k], r);
        r -= y;
        sq = fmaf(r, r, sq);
#pragma unroll
        for (int k = 0; k < P; ++k) acc[k] = fmaf(r, th[k], acc[k]);
    }
    // wave-level butterfly (deterministic), lane 0 of every wave leaves its row's fp64 partials
    double* dst = ws + ((long)blockIdx.y * gridDim.x + blockIdx.x) * NACC;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float t = wave_sum(acc[k]);
        if (lane == 0) dst[1 + row * P + k] = (double)t;
    }
    const float tsq = wave_sum(sq);
    if (lane == 0) loss_part[row] = (double)tsq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) t += loss_part[j];
        dst[0] = t;
    }
}

// `ws` of every launcher below is the caller's workspace base: [header | partials].
inline Finish make_finish(double* ws, const float* mask, float loss_scale, float grad_scale, float* loss, float* grad) {
    return Finish{reinterpret_cast<unsigned long long*>(ws), mask, loss_scale, grad_scale, loss, grad,
                  fused_finalize_mode(), 1};
}

// second launch of the two-launch path
inline hipError_t launch_finalize(const Finish& fin, double* part, long S, int gx, int nacc, hipStream_t st) {
    if (fin.fused) return hipSuccess;
    finalize_kernel<0><<<dim3((unsigned)S), dim3(BLOCK), 0, st>>>(part, gx, nacc, fin.n_loss, fin.mask, fin.loss_scale,
                                                               fin.grad_scale, fin.loss, fin.grad);
    return hipGetLastError();
}

template <class Lib>
hipError_t launch_loss_grad(const float* x, const float* dx, long S, long n, const float* xi, const float* mask,
                            float inv_count, float* loss, float* grad, double* ws, int gx, hipStream_t st) {
    constexpr int NACC = 1 + Lib::D * Lib::P;
    const bool vec = vec_ok(x, n, Lib::D, S) && vec_ok(dx, n, Lib::D, S);
    // A single large problem runs as ONE balanced round of resident workgroups (grid-stride inside):
    // 2048 blocks over 768 resident slots (order 5) left a 2/3-empty last round (-20 % at N = 2^27).
    static const int resident = [] {
        int nb = 0, cu = 256, dev = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, loss_grad_kernel<Lib, (Lib::D == 2) && (Lib::D * Lib::P <= SGPR_XI_MAX)>, BLOCK, 0) != hipSuccess || nb < 1) nb = 2;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 256;
        return nb * cu;
    }();
    if (S == 1 && gx > resident) gx = resident;
    const int seg_env = knobs().segmented;
    // contiguous slab per workgroup: 1 = for one big problem, 2 = also inside every problem of a batch
    const bool seg = (seg_env == 1 && S == 1 && gx >= 64) || (seg_env == 2 && gx >= 2);
    const dim3 grid(gx, (unsigned)S), block(BLOCK);
    // The four-slot ring pays where the library leaves registers for it: d = 2 (every task the reference ships).  Mid-size
    // libraries run two slots, the largest ones (D*P > SGPR_XI_MAX) the row-per-wave kernel.
    constexpr bool TUNED = (Lib::D == 2) && (Lib::D * Lib::P <= SGPR_XI_MAX);
    constexpr bool ROWS = (Lib::D >= 2) && (Lib::D * Lib::P > SGPR_XI_MAX);
    const int rows_env = knobs().row_split;
    double* part = ws + WS_HEADER_DOUBLES;
    Finish fin = make_finish(ws, mask, inv_count, 2.0f * inv_count, loss, grad);
    if constexpr (ROWS) {
        if (rows_env != 0) {
            // a workgroup covers 64 points per step: give it as many steps as the thread-per-point form has
            loss_grad_rows_kernel<Lib><<<grid, dim3(Lib::D * WAVE), 0, st>>>(x, dx, n, xi, mask, part);
            SYMODE_LAUNCH_CHECK();
            fin.fused = 0;                              // D waves per workgroup: this form keeps the second launch
            return launch_finalize(fin, part, S, gx, NACC, st);
        }
    }
    loss_grad_kernel<Lib, TUNED><<<grid, block, 0, st>>>(x, dx, n, vec, xi, mask, part, fin, seg && TUNED);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, S, gx, NACC, st);
}

template <class Lib>
hipError_t launch_symreg_linear(const float* z, long n, const float* xi, const float* mask, const float* L, int n_gen,
                                float* loss, float* grad, double* ws, int gx, hipStream_t st) {
    constexpr int NACC = 1 + Lib::D * Lib::P;
    double* part = ws + WS_HEADER_DOUBLES;
    const Finish fin = make_finish(ws, mask, 1.0f, 2.0f, loss, grad);
    symreg_linear_kernel<Lib><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(z, n, vec_ok(z, n, Lib::D, 1), xi, mask, L, n_gen,
                                                                  part, fin);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, 1, gx, NACC, st);
}

template <class Lib>
hipError_t launch_symreg_reversed(const float* x, const float* dx, const float* gxp, const float* jgx, int n_g, long S, long n,
                                  const float* xi, const float* mask, float inv_count, float w_sym, float* loss, float* grad,
                                  double* ws, int gx, hipStream_t st) {
    constexpr int D = Lib::D;
    const bool mse = dx != nullptr;
    const int nacc = (mse ? 2 : 1) + D * Lib::P;
    double* part = ws + WS_HEADER_DOUBLES;
    Finish fin = make_finish(ws, mask, inv_count, 2.0f * inv_count, loss, grad);
    fin.n_loss = mse ? 2 : 1;
    // 16-byte vectors need every slab (problem, group element) to start on a 16-byte boundary
    const bool multi = S > 1 || n_g > 1;
    const bool vec = ((uintptr_t)x % 16 == 0) && ((uintptr_t)gxp % 16 == 0) && ((uintptr_t)jgx % 16 == 0) &&
                     (!mse || (uintptr_t)dx % 16 == 0) && (!multi || ((n * D) % 4 == 0 && (n * D * D) % 4 == 0));
    if (mse)
        symreg_reversed_kernel<Lib, true><<<dim3(gx, (unsigned)S), dim3(BLOCK), 0, st>>>(x, dx, gxp, jgx, n_g, n, vec, xi, mask, w_sym,
                                                                                       part, fin);
    else
        symreg_reversed_kernel<Lib, false><<<dim3(gx, (unsigned)S), dim3(BLOCK), 0, st>>>(x, nullptr, gxp, jgx, n_g, n, vec, xi, mask,
                                                                                        1.0f, part, fin);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, S, gx, nacc, st);
}

template <class Lib>
hipError_t launch_vjp(const float* x, const float* g, long n, const float* xi, const float* mask, float* grad_x,
                      float* grad_xi, double* ws, int gx, hipStream_t st) {
    constexpr int NACC = 1 + Lib::D * Lib::P;
    double* part = ws + WS_HEADER_DOUBLES;
    const Finish fin = make_finish(ws, mask, 0.0f, 1.0f, nullptr, grad_xi);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(g, n, Lib::D, 1) && (grad_x == nullptr || vec_ok(grad_x, n, Lib::D, 1));
    if (grad_x != nullptr)
        vjp_kernel<Lib, true><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(x, g, n, vec, xi, mask, grad_x, part, fin);
    else
        vjp_kernel<Lib, false><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(x, g, n, vec, xi, mask, nullptr, part, fin);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, 1, gx, NACC, st);
}

template <class Lib>
hipError_t launch_forward_jvp(const float* x, const float* v, long n, const float* xi, const float* mask, float* out,
                              float* jv, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int g = map_grid_for(n, Chunk<Lib::D>::PPT);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(v, n, Lib::D, 1) && vec_ok(jv, n, Lib::D, 1) &&
                     (out == nullptr || vec_ok(out, n, Lib::D, 1));
    forward_jvp_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x, v, n, vec, xi, mask, out, jv);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

template <class Lib>
hipError_t launch_jvp_vjp(const float* x, const float* v, const float* g_out, const float* g_jv, long n, const float* xi,
                          const float* mask, float* grad_x, float* grad_v, float* grad_xi, double* ws, int gx,
                          hipStream_t st) {
    constexpr int NACC = 1 + Lib::D * Lib::P;
    double* part = ws + WS_HEADER_DOUBLES;
    const Finish fin = make_finish(ws, mask, 0.0f, 1.0f, nullptr, grad_xi);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(v, n, Lib::D, 1) && vec_ok(g_jv, n, Lib::D, 1) &&
                     (g_out == nullptr || vec_ok(g_out, n, Lib::D, 1)) && vec_ok(grad_x, n, Lib::D, 1) && vec_ok(grad_v, n, Lib::D, 1);
    if (g_out != nullptr)
        jvp_vjp_kernel<Lib, true><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(x, v, g_out, g_jv, n, vec, xi, mask, grad_x, grad_v, part, fin);
    else
        jvp_vjp_kernel<Lib, false><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(x, v, nullptr, g_jv, n, vec, xi, mask, grad_x, grad_v, part, fin);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, 1, gx, NACC, st);
}

template <class Lib>
hipError_t launch_euler_jvp(const float* x, const float* v, long n, const float* xi, const float* mask, int n_steps,
                            float dt, float* x_out, float* t_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int g = map_grid_for(n, Chunk<Lib::D>::PPT);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(v, n, Lib::D, 1) && vec_ok(x_out, n, Lib::D, 1) && vec_ok(t_out, n, Lib::D, 1);
    euler_jvp_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x, v, n, vec, xi, mask, n_steps, dt, x_out, t_out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

template <class Lib>
hipError_t launch_euler_jvp_vjp(const float* x, const float* v, const float* g_x, const float* g_t, long n,
                                const float* xi, const float* mask, int n_steps, float dt, float* grad_x, float* grad_v,
                                float* grad_xi, double* ws, int gx, hipStream_t st) {
    constexpr int NACC = 1 + Lib::D * Lib::P;
    double* part = ws + WS_HEADER_DOUBLES;
    const Finish fin = make_finish(ws, mask, 0.0f, 1.0f, nullptr, grad_xi);
    const bool vec = vec_ok(x, n, Lib::D, 1) && vec_ok(v, n, Lib::D, 1) && vec_ok(g_x, n, Lib::D, 1) && vec_ok(g_t, n, Lib::D, 1) &&
                     vec_ok(grad_x, n, Lib::D, 1) && vec_ok(grad_v, n, Lib::D, 1);
    // per-thread LDS column for the K states while it fits (40 KB at K = 10, D = 2: four workgroups per CU now that the
    // reduction stages inside the column); SYMODE_EULER_STACK=0 forces the recompute form (A/B and the parity test)
    const size_t stack_bytes = (size_t)n_steps * 2 * Lib::D * BLOCK * sizeof(float);
    const size_t stage_bytes = reduce_lds_floats(BLOCK) * sizeof(float);        // the reduction reuses the column afterwards
    const bool use_stack = n_steps > 1 && stack_bytes <= 64 * 1024 && knobs().euler_stack != 0;
    if (use_stack)
        euler_jvp_vjp_kernel<Lib, true><<<dim3(gx, 1), dim3(BLOCK), stack_bytes > stage_bytes ? stack_bytes : stage_bytes, st>>>(x, v, g_x, g_t, n, vec, xi, mask, n_steps, dt,
                                                                                      grad_x, grad_v, part, fin);
    else
        euler_jvp_vjp_kernel<Lib, false><<<dim3(gx, 1), dim3(BLOCK), 0, st>>>(x, v, g_x, g_t, n, vec, xi, mask, n_steps, dt, grad_x,
                                                                             grad_v, part, fin);
    SYMODE_LAUNCH_CHECK();
    return launch_finalize(fin, part, 1, gx, NACC, st);
}

template <class Lib>
hipError_t launch_rk4_traj(const double* x0, long n_traj, const double* xi, int n_steps, double dt, int subsample,
                           float* x_out, float* dx_out, hipStream_t st) {
    if (n_traj == 0) return hipSuccess;
    const unsigned g = (unsigned)((n_traj + BLOCK - 1) / BLOCK);
    rk4_traj_kernel<Lib><<<dim3(g), dim3(BLOCK), 0, st>>>(x0, n_traj, xi, n_steps, dt, subsample, x_out, dx_out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace symode
