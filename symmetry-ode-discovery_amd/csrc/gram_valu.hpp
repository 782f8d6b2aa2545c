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

#include "gram.hpp"

namespace symode {

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
        const long nchunks = N / PPT;
        long c = tid;
        for (; c + nthreads < nchunks; c += 2 * nthreads) {
            float4 ax[NV], ay[NV], bx[NV], by[NV];
            load_chunk_raw<D, true>(xs, c, ax);
            load_chunk_raw<D, true>(ys, c, ay);
            load_chunk_raw<D, true>(xs, c + nthreads, bx);
            load_chunk_raw<D, true>(ys, c + nthreads, by);
            float xa[PPT][D], ya[PPT][D], xb[PPT][D], yb[PPT][D];
            unpack_chunk<D>(ax, xa);
            unpack_chunk<D>(ay, ya);
            unpack_chunk<D>(bx, xb);
            unpack_chunk<D>(by, yb);
#pragma unroll
            for (int i = 0; i < PPT; ++i) one(xa[i], ya[i]);
#pragma unroll
            for (int i = 0; i < PPT; ++i) one(xb[i], yb[i]);
        }
        if (c < nchunks) {
            float4 ax[NV], ay[NV];
            load_chunk_raw<D, true>(xs, c, ax);
            load_chunk_raw<D, true>(ys, c, ay);
            float xa[PPT][D], ya[PPT][D];
            unpack_chunk<D>(ax, xa);
            unpack_chunk<D>(ay, ya);
#pragma unroll
            for (int i = 0; i < PPT; ++i) one(xa[i], ya[i]);
        }
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

inline bool gram_valu_gather_enabled() {
    const char* e = getenv("SYMODE_GRAM_VALU_GATHER");
    return e && e[0] == '1';
}

inline bool gram_valu_enabled() {
    const char* e = getenv("SYMODE_GRAM_VALU");       // tuning / A-B knob: 0 forces the MFMA form for every library
    return !(e && e[0] == '0');
}

template <class Lib>
hipError_t launch_aug_gram_any(const float* x, const float* dx, long S, long n, const int* idx, double* gram, double* ws,
                               int gx_mfma, int gx_valu, hipStream_t st) {
    // index-table launches (random 8-byte rows) need more waves in flight than the register-bound vector form has:
    // they keep the MFMA form, whose thread-per-point feature build runs at 4+ waves per SIMD
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
    return launch_aug_gram<Lib>(x, dx, S, n, idx, gram, ws, gx_mfma, st);
}

}  // namespace symode
