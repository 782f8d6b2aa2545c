// Weak-SINDy contraction fused with the library build (reference sindy.py:362-381):
//   G = V Theta(x)  (K, p)      b = -V' x  (K, d)
// for K test functions sampled on the T time points of ONE trajectory -- without writing Theta (48 B/point at p = 10)
// and re-reading it from a library GEMM.  One GEMM-shaped contraction over time, so it goes to the fp64 matrix
// cores like the Gram:  [V; -V'] (2K, T)  x  [Theta | x] (T, p + d), accumulated in fp64 (products of fp32 values are
// exact there), of which the host keeps the two blocks above.
//
// Mapping.  grid.y = 16-row tile of the stacked test functions, grid.x = slabs of time.  A wave takes 64 time points
// per step: thread-per-point builds the F = p + d features into a wave-private LDS slab [feature][point] (row stride 66,
// conflict-free both ways, as in gram.hpp); the B operand of v_mfma_f64_16x16x4_f64 -- lane l: feature 16 ct + (l & 15) of
// point 4 ks + (l >> 4) -- is read back from it, the A operand -- test function 16 rt + (l & 15) at that point -- straight
// from global memory (V, V' are a few MB and stay in L2; the trajectory is at most 10^4 points, the launch is
// latency-bound either way).  Every workgroup leaves one (CT x 256) fp64 partial per row tile; a second small kernel adds
// them in fixed order.
#pragma once
#include <hip/hip_runtime.h>

#include "gram.hpp"

namespace symode {

template <class Lib>
struct WeakShape {
    static constexpr int F = Lib::P + Lib::D;
    static constexpr int CT = (F + 15) / 16;           // 16-wide column tiles of [Theta | x]
    static constexpr int FT = CT * 16;
    static constexpr int PS = 66;
};

template <class Lib>
__global__ __launch_bounds__(BLOCK) void weak_gram_kernel(const float* __restrict__ x, long T, const float* __restrict__ V,
                                                          const float* __restrict__ Vd, int K, double* __restrict__ part) {
    using W = WeakShape<Lib>;
    constexpr int D = Lib::D, P = Lib::P, F = W::F, CT = W::CT, PS = W::PS;
    __shared__ float lds[(BLOCK / WAVE) * W::FT * PS];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    float* my = lds + wave * W::FT * PS;
    const int rt = blockIdx.y;
    const int row = 16 * rt + (lane & 15);               // row of [V; -V'] this lane feeds
    const float* arow = row < K ? V + (long)row * T : (row < 2 * K ? Vd + (long)(row - K) * T : nullptr);
    const float sign = row < K ? 1.0f : -1.0f;

    for (int f = F; f < W::FT; ++f) my[f * PS + lane] = 0.0f;        // padding features stay zero
    double4_t acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};

    const long n_slabs = (T + WAVE - 1) / WAVE;
    const long stride = (long)gridDim.x * (BLOCK / WAVE);
    for (long slab = (long)blockIdx.x * (BLOCK / WAVE) + wave; slab < n_slabs; slab += stride) {
        const long t = slab * WAVE + lane;
        float feat[F];
        if (t < T) {
            float xp[D], th[P];
            load_point<D>(x, t, xp);
            Lib::eval(xp, th);
#pragma unroll
            for (int k = 0; k < P; ++k) feat[k] = th[k];
#pragma unroll
            for (int j = 0; j < D; ++j) feat[P + j] = xp[j];
        } else {
#pragma unroll
            for (int k = 0; k < F; ++k) feat[k] = 0.0f;
        }
        // the slab's 16 A operands (test function `row` at 4 time points per MFMA step) are requested up front: at
        // T = 10^4 a wave sees one or two slabs, and sixteen dependent global loads in the MFMA loop WERE the launch
        // (24.6 us for 10^4 points x 50 test functions before, profiles/r03_weak_gram_kernel_stats.txt)
        float av[WAVE / 4];
#pragma unroll
        for (int ks = 0; ks < WAVE / 4; ++ks) {
            const long tt = slab * WAVE + 4 * ks + (lane >> 4);
            av[ks] = (arow != nullptr && tt < T) ? arow[tt] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();                 // the previous slab's operand reads are done (wave-private slab)
#pragma unroll
        for (int k = 0; k < F; ++k) my[k * PS + lane] = feat[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ks = 0; ks < WAVE / 4; ++ks) {
            const double a = (double)(sign * av[ks]);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const double b = (double)my[(16 * c + (lane & 15)) * PS + 4 * ks + (lane >> 4)];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
            }
        }
    }
    // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg; the four waves are combined in fixed order through LDS
    // (the operand slabs are free once every wave has left the loop)
    __syncthreads();
    double* comb = reinterpret_cast<double*>(lds);
    static_assert(sizeof(lds) >= sizeof(double) * (BLOCK / WAVE) * CT * 256, "LDS too small for the wave combine");
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) comb[(wave * CT + c) * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[c][r];
    __syncthreads();
    double* dst = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (CT * 256);
    for (int c = 0; c < CT; ++c) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) v += comb[(w * CT + c) * 256 + threadIdx.x];
        dst[c * 256 + threadIdx.x] = v;
    }
}

// out (RT*16, CT*16) row-major = sum over the gx partials of every row tile, fixed order.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void weak_gram_finalize_kernel(const double* __restrict__ part, int gx,
                                                                   double* __restrict__ out) {
    using W = WeakShape<Lib>;
    constexpr int CT = W::CT;
    const int rt = blockIdx.x, e = threadIdx.x, row = e >> 4, col = e & 15;
    for (int c = 0; c < CT; ++c) {
        const double* src = part + (long)rt * gx * (CT * 256) + c * 256 + e;
        double v = 0.0;
        int g = 0;
        for (; g + 8 <= gx; g += 8) {                    // eight partials in flight, added in order (40-64 of them at T >= 10^4)
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(long)(g + u) * (CT * 256)];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; g < gx; ++g) v += src[(long)g * (CT * 256)];
        out[(long)(16 * rt + row) * (CT * 16) + 16 * c + col] = v;
    }
}

template <class Lib>
hipError_t launch_weak_gram(const float* x, long T, const float* V, const float* Vd, int K, double* out, double* ws, int gx,
                            hipStream_t st) {
    const int RT = (2 * K + 15) / 16;
    double* part = ws + WS_HEADER_DOUBLES;
    weak_gram_kernel<Lib><<<dim3(gx, RT), dim3(BLOCK), 0, st>>>(x, T, V, Vd, K, part);
    SYMODE_LAUNCH_CHECK();
    weak_gram_finalize_kernel<Lib><<<dim3(RT), dim3(BLOCK), 0, st>>>(part, gx, out);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace symode
