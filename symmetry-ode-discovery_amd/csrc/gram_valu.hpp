// Augmented Gram matrix for SMALL libraries on the fp64 vector pipe.
//
// MI355X's fp64 matrix rate equals its fp64 vector rate (78.6 TFLOP/s each), so the matrix cores only win when their
// tiles are full.  For F = p + d <= 12 -- every library the reference ships at d = 2: order 2 (F = 8), order 2 + exp
// (F = 10), order 3 (F = 12) -- a 16x16x4 f64 MFMA spends 256 multiply-adds per point to produce the F(F+1)/2 <= 78
// distinct entries of a symmetric matrix (30 % useful at F = 12; gram.hpp keeps that form for F > 12, where the tiles fill
// up and a thread could not hold the accumulators).  Here a thread owns whole points: features in fp64 registers (exact
// conversions of the fp32 library, exact products), F(F+1)/2 fp64 accumulators, one v_fma_f64 per distinct entry per
// point -- 78 instead of 256 -- which moves the kernel from the MFMA roof (0.19 of HBM at F = 12) towards the HBM stream.
// Same numbers as gram.hpp up to summation order: fp64 end to end.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "gram.hpp"
#include "gram_m4.hpp"

namespace symode {

constexpr int GRAM_RING = 3;      // chunks of (x, dx) in flight per lane; 2, 3 and 4 measure alike, the two-chunk LOOP it replaced 8 % slower

template <class Lib>
struct GramValuShape {
    static constexpr int F = Lib::P + Lib::D;
    static constexpr int NPAIR = F * (F + 1) / 2;
    static constexpr bool OK = F <= 12;             // 78 fp64 accumulators + 12 fp64 features: ~200 VGPRs, two waves per SIMD
};

template <class Lib>
__global__ __launch_bounds__(BLOCK) void aug_gram_valu_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                              long N, bool vec, const int* __restrict__ idx,
                                                              double* __restrict__ part) {
    using G = GramValuShape<Lib>;
    constexpr int D = Lib::D, P = Lib::P, F = G::F, NPAIR = G::NPAIR, PPT = Chunk<D>::PPT, NV = Chunk<D>::NV;
    __shared__ double lds[reduce_lds_doubles(BLOCK)];
    const long s = blockIdx.y;
    const float* xs = idx ? x : x + s * N * D;
    const float* ys = idx ? dx : dx + s * N * D;
    const int* is = idx ? idx + s * N : nullptr;
    double acc[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) acc[q] = 0.0;

    auto one = [&](const float (&xp)[D], const float (&yp)[D]) {
        float th[P];
        Lib::eval(xp, th);
        double f[F];
#pragma unroll
        for (int k = 0; k < P; ++k) f[k] = (double)th[k];
#pragma unroll
        for (int j = 0; j < D; ++j) f[P + j] = (double)yp[j];
        int q = 0;
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int j = i; j < F; ++j) {
                acc[q] = fma(f[i], f[j], acc[q]);
                ++q;
            }
    };
    const long tid = (long)blockIdx.x * BLOCK + threadIdx.x, nthreads = (long)gridDim.x * BLOCK;
    if (is == nullptr && vec) {
        // register ring: GRAM_RING chunks of x and dx in flight per lane, a slot refilled as soon as it is consumed
        // (points.hpp, chunk_ring) -- at two waves per SIMD (the 78 fp64 sums) the wave's own run-ahead is what hides HBM:
        // order 3, 2^26 points 275 -> 253 us, 1024 x 125 000 495 -> 464 us, order 2 + exp 258 -> 236 us (same box)
        const long nchunks = N / PPT;
        chunk_ring<GRAM_RING, 2 * NV>(
            nchunks, tid, nthreads,
            [&](long q, float4 (&slot)[2 * NV]) {
                float4 tx[NV], ty[NV];
                load_chunk_raw<D, true>(xs, q, tx);
                load_chunk_raw<D, true>(ys, q, ty);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    slot[i] = tx[i];
                    slot[NV + i] = ty[i];
                }
            },
            [&](long, const float4 (&slot)[2 * NV]) {
                float4 tx[NV], ty[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    tx[i] = slot[i];
                    ty[i] = slot[NV + i];
                }
                float xa[PPT][D], ya[PPT][D];
                unpack_chunk<D>(tx, xa);
                unpack_chunk<D>(ty, ya);
#pragma unroll
                for (int i = 0; i < PPT; ++i) one(xa[i], ya[i]);
            });
        const long n = nchunks * PPT + tid;
        if (n < N) {
            float xp[D], yp[D];
            load_point<D>(xs, n, xp);
            load_point<D>(ys, n, yp);
            one(xp, yp);
        }
    } else {
        // index table (seed sweeps over subsamples of ONE data set) or unaligned rows: point by point, the next point's
        // operands requested before this point's arithmetic
        long n = tid;
        float xn[D], yn[D];
        if (n < N) {
            const long src = is ? (long)is[n] : n;
            load_point<D>(xs, src, xn);
            load_point<D>(ys, src, yn);
        }
        for (; n < N; n += nthreads) {
            float xp[D], yp[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                xp[j] = xn[j];
                yp[j] = yn[j];
            }
            if (n + nthreads < N) {
                const long src = is ? (long)is[n + nthreads] : n + nthreads;
                load_point<D>(xs, src, xn);
                load_point<D>(ys, src, yn);
            }
            one(xp, yp);
        }
    }
    double* dst = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * NPAIR;
    block_reduce_emit_lds_f64<NPAIR, BLOCK>(acc, lds, [&](int q, double v) { dst[q] = v; });
}

// Sum the gx block partials of problem s in fixed order and scatter into the dense symmetric (F, F) matrix.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void gram_valu_finalize_kernel(const double* __restrict__ part, int gx,
                                                                   double* __restrict__ gram) {
    using G = GramValuShape<Lib>;
    constexpr int F = G::F, NPAIR = G::NPAIR, LANES = 128, PARTS = BLOCK / LANES;   // NPAIR <= 78 < 128
    __shared__ double comb[BLOCK];
    const long s = blockIdx.x;
    const double* src = part + s * (long)gx * NPAIR;
    const int q = threadIdx.x % LANES, pt = threadIdx.x / LANES;
    double v = 0.0;
    if (q < NPAIR) {
        int g = pt;
        for (; g + 7 * PARTS < gx; g += 8 * PARTS) {                  // 8 independent loads in flight, added in fixed order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(long)(g + u * PARTS) * NPAIR + q];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; g < gx; g += PARTS) v += src[(long)g * NPAIR + q];
    }
    comb[threadIdx.x] = v;
    __syncthreads();
    if (pt == 0 && q < NPAIR) {
        double t = comb[q];
#pragma unroll
        for (int u = 1; u < PARTS; ++u) t += comb[u * LANES + q];
        // q -> (i, j), i <= j, row-major over the upper triangle
        int i = 0, rem = q;
        while (rem >= F - i) {
            rem -= F - i;
            ++i;
        }
        const int j = i + rem;
        double* out = gram + s * (long)F * F;
        out[i * F + j] = t;
        out[j * F + i] = t;
    }
}

// ---------------------------------------------------------------------------------------
// 12 < F <= 24 (order 4-5 at d = 2, the d = 3 / d = 4 libraries up to 20 terms): the F(F+1)/2 <= 300 sums do not fit one
// thread, and a 16x16x4 MFMA tiling spends 768 multiply-adds per point on them (36 % useful at F = 23; gram.hpp: 0.08 of
// the HBM roof).  The triangle is therefore cut into NPART <= 4 runs of NE <= 78 consecutive entries (row-major), and a
// WORKGROUP owns one run: every thread evaluates the whole library for its points (fp32, ~20 multiplies at order 5 -- cheap
// beside NE fp64 FMAs) and accumulates its run only; which run is wave-uniform (a switch over PART with one unrolled body
// each), so the feature registers are indexed at compile time.  The NPART workgroups that share a slab of points get
// consecutive-by-8 linear ids -- same XCD, same L2 -- so the slab comes from HBM once.
// ---------------------------------------------------------------------------------------
constexpr int pair_row(int F, int q) {
    int i = 0;
    while (q >= F - i) {
        q -= F - i;
        ++i;
    }
    return i;
}
constexpr int pair_col(int F, int q) {
    int i = 0;
    while (q >= F - i) {
        q -= F - i;
        ++i;
    }
    return i + q;
}

template <class Lib>
struct GramSplitShape {
    static constexpr int F = Lib::P + Lib::D;
    static constexpr int NPAIR = F * (F + 1) / 2;
    static constexpr bool OK = F > 12 && F <= 24;
    static constexpr int NPART = (NPAIR + 77) / 78;              // 2 .. 4
    static constexpr int NE = (NPAIR + NPART - 1) / NPART;       // entries per run (the last run may be shorter)
};

// one entry of a run: (row, col) are template-time constants, so f[] stays in registers
template <int F, int NPAIR, int Q0, int NE, int E>
__device__ __forceinline__ void gram_run_one(const double (&f)[F], double (&acc)[NE]) {
    if constexpr (Q0 + E < NPAIR) {
        constexpr int I = pair_row(F, Q0 + E), J = pair_col(F, Q0 + E);
        acc[E] = fma(f[I], f[J], acc[E]);
    }
}

template <int F, int NPAIR, int Q0, int NE, int... E>
__device__ __forceinline__ void gram_run_accumulate(const double (&f)[F], double (&acc)[NE], std::integer_sequence<int, E...>) {
    (gram_run_one<F, NPAIR, Q0, NE, E>(f, acc), ...);
}

template <class Lib, int PART>
__device__ __forceinline__ void gram_split_body(const float* __restrict__ xs, const float* __restrict__ ys, long N, bool vec,
                                                long tid, long nthreads, double* __restrict__ dst, double* lds) {
    using G = GramSplitShape<Lib>;
    constexpr int D = Lib::D, P = Lib::P, F = G::F, NE = G::NE, PPT = Chunk<D>::PPT, NV = Chunk<D>::NV;
    double acc[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) acc[q] = 0.0;
    auto one = [&](const float (&xp)[D], const float (&yp)[D]) {
        float th[P];
        Lib::eval(xp, th);
        double f[F];
#pragma unroll
        for (int k = 0; k < P; ++k) f[k] = (double)th[k];
#pragma unroll
        for (int j = 0; j < D; ++j) f[P + j] = (double)yp[j];
        gram_run_accumulate<F, G::NPAIR, PART * NE, NE>(f, acc, std::make_integer_sequence<int, NE>{});
    };
    if (vec) {
        // register ring of two chunks per lane, a slot refilled as soon as it is consumed (the run's NE fp64 sums and the F
        // fp64 features leave no room for more in flight: 2 waves per SIMD); plain loads: the sibling runs read the same
        // lines from L2
        const long nchunks = N / PPT;
        chunk_ring<2, 2 * NV>(
            nchunks, tid, nthreads,
            [&](long q, float4 (&slot)[2 * NV]) {
                float4 tx[NV], ty[NV];
                load_chunk_raw<D, false>(xs, q, tx);
                load_chunk_raw<D, false>(ys, q, ty);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    slot[i] = tx[i];
                    slot[NV + i] = ty[i];
                }
            },
            [&](long, const float4 (&slot)[2 * NV]) {
                float4 tx[NV], ty[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    tx[i] = slot[i];
                    ty[i] = slot[NV + i];
                }
                float xa[PPT][D], ya[PPT][D];
                unpack_chunk<D>(tx, xa);
                unpack_chunk<D>(ty, ya);
#pragma unroll 1
                for (int i = 0; i < PPT; ++i) one(xa[i], ya[i]);
            });
        const long n = nchunks * PPT + tid;
        if (n < N) {
            float xp[D], yp[D];
            load_point<D>(xs, n, xp);
            load_point<D>(ys, n, yp);
            one(xp, yp);
        }
    } else {
        for (long n = tid; n < N; n += nthreads) {
            float xp[D], yp[D];
            load_point<D>(xs, n, xp);
            load_point<D>(ys, n, yp);
            one(xp, yp);
        }
    }
    block_reduce_emit_lds_f64<NE, BLOCK>(acc, lds, [&](int q, double v) { dst[q] = v; });
}

// grid = (GX * NPART, S), GX a multiple of 8.  Linear id L -> slab column bx = (L % 8) + 8 (L / (8 NPART)), run
// part = (L / 8) % NPART: the NPART runs of a column sit 8 ids apart (same XCD under round-robin dispatch).
template <class Lib>
__global__ __launch_bounds__(BLOCK) void aug_gram_split_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                               long N, bool vec, int GX, double* __restrict__ part_ws) {
    using G = GramSplitShape<Lib>;
    constexpr int D = Lib::D, NPART = G::NPART, NE = G::NE;
    __shared__ double lds[reduce_lds_doubles(BLOCK)];
    const long s = blockIdx.y;
    const int L = blockIdx.x;
    const int bx = (L % 8) + 8 * (L / (8 * NPART)), part = (L / 8) % NPART;
    const float* xs = x + s * N * D;
    const float* ys = dx + s * N * D;
    const long tid = (long)bx * BLOCK + threadIdx.x, nthreads = (long)GX * BLOCK;
    double* dst = part_ws + (((long)s * GX + bx) * NPART + part) * NE;
    if (part == 0) gram_split_body<Lib, 0>(xs, ys, N, vec, tid, nthreads, dst, lds);
    if constexpr (NPART > 1) if (part == 1) gram_split_body<Lib, 1>(xs, ys, N, vec, tid, nthreads, dst, lds);
    if constexpr (NPART > 2) if (part == 2) gram_split_body<Lib, 2>(xs, ys, N, vec, tid, nthreads, dst, lds);
    if constexpr (NPART > 3) if (part == 3) gram_split_body<Lib, 3>(xs, ys, N, vec, tid, nthreads, dst, lds);
}

// (Round 3 tried the runs of a slab column as the WAVES of one workgroup instead: every wave evaluates and converts the
// features of its own 64 points once, parks them as fp64 in LDS, and after a barrier adds its run over all NPART groups
// -- 309 instead of 454 issue cycles per point and run on paper.  Measured 25 % SLOWER than this form on every shape
// (1024 x 125 000 points at order 5: 49.9 against 66.2 G points/s; profiles/r03_gram_forms.txt): with two waves per SIMD
// the 23 dependent ds_read_b64 per group and two barriers per step cost more than the library re-evaluation saves.  Removed.)
// Sum the GX partial runs of problem s in fixed order and scatter into the dense symmetric (F, F) matrix.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void gram_split_finalize_kernel(const double* __restrict__ part_ws, int GX,
                                                                    double* __restrict__ gram) {
    using G = GramSplitShape<Lib>;
    constexpr int F = G::F, NPAIR = G::NPAIR, NPART = G::NPART, NE = G::NE;
    const long s = blockIdx.x;
    const double* src = part_ws + s * (long)GX * NPART * NE;
    double* out = gram + s * (long)F * F;
    for (int q = threadIdx.x; q < NPAIR; q += BLOCK) {
        const int part = q / NE, e = q - part * NE;
        double v = 0.0;
        int g = 0;
        for (; g + 8 <= GX; g += 8) {                          // 8 independent loads in flight, added in fixed order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[((long)(g + u) * NPART + part) * NE + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; g < GX; ++g) v += src[((long)g * NPART + part) * NE + e];
        const int i = pair_row(F, q), j = pair_col(F, q);
        out[i * F + j] = v;
        out[j * F + i] = v;
    }
}

inline bool gram_split_enabled() { return knobs().gram_split != 0; }      // A-B knob: 0 keeps the MFMA form for 12 < F <= 24

// Index-table launches (seed sweeps over sorted subsamples): the vector-pipe form prefetches the next point's rows and,
// since the launch geometry gives every thread >= 32 points, beats the MFMA form here too (r02_gather_gram.txt: 64 x 500 000
// of 1 M rows 259 vs 410 us; 256 x 200 000 of 2 M rows 423 vs 634 us).  SYMODE_GRAM_VALU_GATHER=0 keeps the MFMA form.
inline bool gram_valu_gather_enabled() { return knobs().gram_valu_gather != 0; }

inline bool gram_valu_enabled() { return knobs().gram_valu != 0; }       // tuning / A-B knob: 0 forces the MFMA form for every library

template <class Lib>
hipError_t launch_aug_gram_any(const float* x, const float* dx, long S, long n, const int* idx, double* gram, double* ws,
                               int gx_mfma, int gx_valu, int gx_m4, hipStream_t st) {
    // (12 < F <= 24: the 4x4-tile matrix-core form for F >= 18 and for every index-table launch -- the split kernel has no
    //  gather path --, the split vector-pipe form for contiguous rows at F < 18)
    if constexpr (GramValuShape<Lib>::OK) {
        if (gram_valu_enabled() && (idx == nullptr || gram_valu_gather_enabled())) {
            double* part = ws + WS_HEADER_DOUBLES;
            const bool vec = vec_ok(x, n, Lib::D, S) && vec_ok(dx, n, Lib::D, S);
            aug_gram_valu_kernel<Lib><<<dim3(gx_valu, (unsigned)S), dim3(BLOCK), 0, st>>>(x, dx, n, vec, idx, part);
            SYMODE_LAUNCH_CHECK();
            gram_valu_finalize_kernel<Lib><<<dim3((unsigned)S), dim3(BLOCK), 0, st>>>(part, gx_valu, gram);
            SYMODE_LAUNCH_CHECK();
            return hipSuccess;
        }
    }
    if constexpr (GramM4Shape<Lib>::OK) {
        // 4x4 tiles on the matrix cores: contiguous rows and index tables alike (gram_m4.hpp); SYMODE_GRAM_M4=0 keeps the
        // split form / the 16x16 form below
        // (batches of contiguous rows at F < 18 stay with the split form below -- F = 17 fills 64 % of its 15 tiles: 512 x
        //  125 000 points 124 against 118 G points/s; one problem is faster here at every size: 16 M points 105 against 95)
        if (knobs().gram_m4 != 0 && (idx != nullptr || S == 1 || GramM4Shape<Lib>::F >= 18 || !gram_split_enabled()))
            return launch_aug_gram_m4<Lib>(x, dx, S, n, idx, gram, ws, gx_m4, st);
    }
    if constexpr (GramSplitShape<Lib>::OK) {
        if (gram_valu_enabled() && gram_split_enabled() && idx == nullptr) {
            double* part = ws + WS_HEADER_DOUBLES;
            const bool vec = vec_ok(x, n, Lib::D, S) && vec_ok(dx, n, Lib::D, S);
            // one problem: 128 columns x NPART runs = one resident round of the chip (2 workgroups per CU at 193 VGPRs):
            // 16 M points, order 5: 268 us against 334 us on 1024 columns (r02_gram_split.txt)
            int GX = (gx_valu + 7) / 8 * 8;
            if (S == 1 && GX > 128 && knobs().gram_valu_grid < 0) GX = 128;
            aug_gram_split_kernel<Lib><<<dim3(GX * GramSplitShape<Lib>::NPART, (unsigned)S), dim3(BLOCK), 0, st>>>(x, dx, n, vec, GX, part);
            SYMODE_LAUNCH_CHECK();
            gram_split_finalize_kernel<Lib><<<dim3((unsigned)S), dim3(BLOCK), 0, st>>>(part, GX, gram);
            SYMODE_LAUNCH_CHECK();
            return hipSuccess;
        }
    }
    return launch_aug_gram<Lib>(x, dx, S, n, idx, gram, ws, gx_mfma, st);
}

}  // namespace symode
