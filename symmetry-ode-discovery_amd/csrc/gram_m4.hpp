// Augmented Gram matrix  S = [Theta | dx]^T [Theta | dx]  for 12 < F = p + d <= 24 (order 4-5 at d = 2, the d = 3 libraries
// up to 20 terms) on v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction.  (What the host solves from
// S replaces torch.linalg.lstsq on the ridge-augmented / block-diagonal system of sindy.py:250-315.)
//
// Why this shape.  fp64 products run on one set of units whoever issues them (tools/micro/mfma_f64_probe,
// profiles/r03_mfma_f64_probe.txt: v_fma_f64 33.9 T multiply-adds/s, 4x4x4 MFMA 33.9-34.4, both interleaved 36.9, and
// v_mfma_f64_16x16x4_f64 only 24.0), so a form's speed is the share of those slots it spends on distinct entries of the
// triangle and what it needs beside them:
//   gram.hpp, 16x16 tiles:  F = 23 pads to 32: 768 slots per point for 276 entries, at the slower instruction;
//   gram_valu.hpp, split:   276 slots per point, but four sibling workgroups each re-evaluate and convert the library
//                           (the 276 fp64 sums do not fit one thread): 66 G points/s at order 5;
//   here:                   4x4 tiles, F = 23 pads to 24: 21 lower-triangular tiles = 336 slots per point (82 % useful),
//                           and NOTHING else on the fp64 units but six conversions per lane and 16 points -- the fp32
//                           library, the operand selects and the loads go to the other pipes while the matrix core works.
//
// Mapping (found with one-hot operands by the probe): lane l of the instruction is block (l & 15) >> 2, k = l >> 4, and
// holds A[i = l & 3][k] and B[k][j = l & 3] of its block; D lane l = D[i = l >> 4][j = l & 3] of block (l & 15) >> 2.  A and
// B share one lane layout, so for S = A^T A one register serves as both.  All four blocks compute the SAME tile over
// DIFFERENT points: 16 points per instruction round; lane l works for the point of (block, k) and holds its features
// 4 t + (l & 3), t = 0 .. T-1, as T fp64 values; tile (ti, tj) += mfma(g[ti], g[tj]).
// A wave takes 64 points per pass: every lane evaluates the fp32 library of ONE point (no redundancy) and parks its F
// features in a wave-private LDS image [feature][column], row stride 66 floats; four rounds then read their operands
// back -- lane (block, k, i) of round r reads feature 4 t + i at column 8 block + 2 r + (k & 1) + 32 (k >> 1): the 32 lanes
// of a half wave (k >> 1 fixed) hit bank (2 i + 8 block + (k & 1) + const) mod 32, all distinct, and writes go lane ->
// consecutive column.  (The first version let the four lanes of a point each evaluate the whole library and select their
// columns: ~100 vector instructions per round beside 21 MFMAs, 61 G points/s -- the vector pipe, not the matrix core,
// was the limit.)  Accumulators: one fp64 per lane and tile (42 registers at F = 23).  At the end the four blocks of a
// tile are added across lanes, the four waves through LDS, and a workgroup leaves one partial of NT * 16 doubles.
// Same numbers as the other forms up to summation order: exact fp64 products of fp32 features, fp64 sums.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace symode {

template <class Lib>
struct GramM4Shape {
    static constexpr int F = Lib::P + Lib::D;
    static constexpr int T = (F + 3) / 4;                // 4-wide tiles per side
    static constexpr int NT = T * (T + 1) / 2;           // lower-triangular tile pairs (ti >= tj), p = ti (ti + 1) / 2 + tj
    static constexpr int PARTIAL = NT * 16;              // doubles per workgroup partial
    // F <= 12 stays on the vector pipe (gram_valu.hpp): measured with this kernel, order 3 (F = 12, 6 tiles, 81 % useful):
    // 2^26 points 317 against 275 us, 1024 x 125 000 574 against 514 us; small launches gain a little (64 x 50 000: 22 against
    // 25 us; config[3]'s index table 60 against 66 us), sparse index tables lose (64 x 62 500 of 8 M rows: 115 against 93 us)
    static constexpr bool OK = F > 12 && F <= 24;
};

// (waves_per_eu(4) for the d = 2 polynomial libraries: 128 registers -- left alone the compiler takes 70 VGPRs + 64 AGPRs = 3
//  waves per SIMD; four measure +1.5 %.  d = 3 and the sine / exp libraries would spill under that budget.)
template <class Lib, int R = 3>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((Lib::D == 2 && !Lib::SINE && !Lib::EXP) ? 4 : 1)))
void aug_gram_m4_kernel(const float* __restrict__ x, const float* __restrict__ dx, long N,
                                                            const int* __restrict__ idx, double* __restrict__ part) {
    using G = GramM4Shape<Lib>;
    constexpr int D = Lib::D, P = Lib::P, F = G::F, T = G::T, NT = G::NT, NV = (2 * D + 3) / 4, PS = 66, NW = BLOCK / WAVE;
    // operand image of the wave's 64 points, later the waves' tile sums
    constexpr int STAGE = 4 * T * PS, COMB = 2 * G::PARTIAL;                     // floats per wave
    __shared__ float lds[NW * (STAGE > COMB ? STAGE : COMB)];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    float* my = lds + wave * STAGE;
    // read side of a round: row offset of this lane's column of a tile, and its point column (+ 2 r per round)
    const int blk = (lane & 15) >> 2, k = lane >> 4;
    const float* rd = my + (lane & 3) * PS + 8 * blk + (k & 1) + 32 * (k >> 1);
    const long s = blockIdx.y;
    // idx == nullptr: problem s owns rows [s*N, (s+1)*N) of x / dx; else its row of the index table into ONE shared (x, dx)
    const float* xs = idx ? x : x + s * N * D;
    const float* ys = idx ? dx : dx + s * N * D;
    const int* is = idx ? idx + s * N : nullptr;
    double acc[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p) acc[p] = 0.0;
#pragma unroll
    for (int f = F; f < 4 * T; ++f) my[f * PS + lane] = 0.0f;                    // padding rows stay zero

    const long npass = (N + WAVE - 1) / WAVE;
    // R passes of (x, dx) in flight per lane (a pass is 84 MFMAs, ~0.6 us, at F = 23), and for index-table launches the
    // row indices one turn further ahead: the refill of a slot uses the index requested when the slot was filled last, so
    // no load waits for the load before it (as one dependent chain per refill, the 64 x 50 000-row table of config[3]
    // ran latency-bound)
    auto src_of = [&](long pass) -> long {
        pass = pass < npass ? pass : npass - 1;
        long n = pass * WAVE + lane;
        n = n < N ? n : N - 1;                                                   // ragged last pass: in bounds, zeroed below
        return is ? (long)is[n] : n;
    };
    auto fetch = [&](long src, float4 (&slot)[NV]) __attribute__((always_inline)) {
        float v[NV * 4];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            v[j] = xs[src * D + j];
            v[D + j] = ys[src * D + j];
        }
#pragma unroll
        for (int j = 2 * D; j < NV * 4; ++j) v[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) slot[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    };
    auto use = [&](long pass, const float4 (&slot)[NV]) __attribute__((always_inline)) {
        float v[NV * 4], xp[D], th[P];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[4 * i] = slot[i].x;
            v[4 * i + 1] = slot[i].y;
            v[4 * i + 2] = slot[i].z;
            v[4 * i + 3] = slot[i].w;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) xp[j] = v[j];
        Lib::eval(xp, th);
        if (pass * WAVE + WAVE <= N) {
#pragma unroll
            for (int j = 0; j < P; ++j) my[j * PS + lane] = th[j];
#pragma unroll
            for (int j = 0; j < D; ++j) my[(P + j) * PS + lane] = v[D + j];
        } else {
            const bool live = pass * WAVE + lane < N;
#pragma unroll
            for (int j = 0; j < P; ++j) my[j * PS + lane] = live ? th[j] : 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) my[(P + j) * PS + lane] = live ? v[D + j] : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double g[T];
#pragma unroll
            for (int t = 0; t < T; ++t) g[t] = (double)rd[4 * t * PS + 2 * r];
            int p = 0;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) {
                    acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(g[ti], g[tj], acc[p], 0, 0, 0);
                    ++p;
                }
        }
        __builtin_amdgcn_wave_barrier();                                         // the next pass overwrites the image (a wave's LDS operations run in order)
    };
    {
        const long stride = (long)gridDim.x * NW;
        long c = (long)blockIdx.x * NW + wave;
        if (c < npass) {
            float4 ring[R][NV];
            long nxt[R];
            each_point_static<0, R>([&](auto k) { fetch(src_of(c + k * stride), ring[k]); });
            each_point_static<0, R>([&](auto k) { nxt[k] = src_of(c + (R + k) * stride); });
            for (; c + (R - 1) * stride < npass; c += R * stride) {
                each_point_static<0, R>([&](auto k) {
                    use(c + k * stride, ring[k]);
                    fetch(nxt[k], ring[k]);
                    nxt[k] = src_of(c + (2 * R + k) * stride);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            each_point_static<0, R>([&](auto k) {
                if (c + k * stride < npass) use(c + k * stride, ring[k]);
            });
        }
    }

    // the four blocks of the instruction worked on different points: lanes l, l ^ 4, l ^ 8, l ^ 12 hold the same tile element
    __syncthreads();
    double* comb = reinterpret_cast<double*>(lds);                               // [wave][NT * 16]
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        double v = acc[p];
        v += __shfl_xor(v, 4, WAVE);
        v += __shfl_xor(v, 8, WAVE);
        if ((lane & 12) == 0) comb[wave * G::PARTIAL + p * 16 + (lane >> 4) * 4 + (lane & 3)] = v;    // element (row l >> 4, col l & 3)
    }
    __syncthreads();
    double* dst = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * G::PARTIAL;
    for (int e = threadIdx.x; e < G::PARTIAL; e += BLOCK) {
        double v = comb[e];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += comb[w * G::PARTIAL + e];
        dst[e] = v;
    }
}

// Sum the gx workgroup partials of problem s in fixed order and scatter into the dense symmetric (F, F) matrix.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void gram_m4_finalize_kernel(const double* __restrict__ part, int gx, double* __restrict__ gram) {
    using G = GramM4Shape<Lib>;
    constexpr int F = G::F;
    const long s = blockIdx.x;
    const double* src = part + s * (long)gx * G::PARTIAL;
    double* out = gram + s * (long)F * F;
    for (int e = threadIdx.x; e < G::PARTIAL; e += BLOCK) {
        double v = 0.0;
        int g = 0;
        for (; g + 8 <= gx; g += 8) {                          // 8 independent loads in flight, added in fixed order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(long)(g + u) * G::PARTIAL + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; g < gx; ++g) v += src[(long)g * G::PARTIAL + e];
        const int p = e >> 4;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= p) ++ti;
        const int tj = p - ti * (ti + 1) / 2;
        const int Rw = 4 * ti + ((e & 15) >> 2), Cl = 4 * tj + (e & 3);
        if (Rw < F && Cl < F) {
            out[Rw * F + Cl] = v;
            if (ti != tj) out[Cl * F + Rw] = v;
        }
    }
}

// The same for MANY partials (one large problem: gx up to 768): one workgroup per tile, 16 slices of the partials per
// element summed side by side (8 loads in flight each), the slices added in fixed order through LDS -- the one-block form
// above needs gx / 8 dependent round trips per element (768 partials: 190 us for a 260 us kernel).
template <class Lib>
__global__ __launch_bounds__(BLOCK) void gram_m4_finalize_wide_kernel(const double* __restrict__ part, int gx, double* __restrict__ gram) {
    using G = GramM4Shape<Lib>;
    constexpr int F = G::F, SL = BLOCK / 16;
    __shared__ double comb[SL][16];
    const long s = blockIdx.x;
    const int p = blockIdx.y, el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const double* src = part + s * (long)gx * G::PARTIAL + p * 16 + el;
    double v = 0.0;
    int g = slice;
    for (; g + 7 * SL < gx; g += 8 * SL) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(long)(g + u * SL) * G::PARTIAL];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; g < gx; g += SL) v += src[(long)g * G::PARTIAL];
    comb[slice][el] = v;
    __syncthreads();
    if (slice == 0) {
        double t = comb[0][el];
#pragma unroll
        for (int u = 1; u < SL; ++u) t += comb[u][el];
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= p) ++ti;
        const int tj = p - ti * (ti + 1) / 2;
        const int Rw = 4 * ti + (el >> 2), Cl = 4 * tj + (el & 3);
        double* out = gram + s * (long)F * F;
        if (Rw < F && Cl < F) {
            out[Rw * F + Cl] = t;
            if (ti != tj) out[Cl * F + Rw] = t;
        }
    }
}

template <class Lib>
hipError_t launch_aug_gram_m4(const float* x, const float* dx, long S, long n, const int* idx, double* gram, double* ws, int gx,
                              hipStream_t st) {
    double* part = ws + WS_HEADER_DOUBLES;
    aug_gram_m4_kernel<Lib><<<dim3(gx, (unsigned)S), dim3(BLOCK), 0, st>>>(x, dx, n, idx, part);
    SYMODE_LAUNCH_CHECK();
    if (gx > 16)
        gram_m4_finalize_wide_kernel<Lib><<<dim3((unsigned)S, GramM4Shape<Lib>::NT), dim3(BLOCK), 0, st>>>(part, gx, gram);
    else
        gram_m4_finalize_kernel<Lib><<<dim3((unsigned)S), dim3(BLOCK), 0, st>>>(part, gx, gram);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace symode
