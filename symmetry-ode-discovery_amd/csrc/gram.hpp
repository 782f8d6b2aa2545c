// Augmented Gram matrix  S = [Theta | dx]^T [Theta | dx]  on the fp64 matrix cores.
//
// The only GEMM-shaped contraction of the path (K = number of points) goes to MFMA:
// v_mfma_f64_16x16x4_f64.  fp32 features are converted to fp64 (exact), every product of
// two fp32 values is exact in fp64, and accumulation is fp64 end to end, so the normal
// equations the host solves from S are as accurate as an fp64 Gram of the fp32 library
// (cond(Theta)^2 amplification makes an fp32-accumulated Gram unusable
// at rtol 1e-5).
//
// Mapping.  A workgroup (4 waves) takes 256 points per step.  Thread-per-point builds the
// F = P + D features in fp32 registers and parks them in LDS as [feature][point] with a row
// stride of 66 floats: writes are conflict-free (consecutive lanes -> consecutive banks),
// and the MFMA operand read  lane l -> feature 16t + (l & 15) of point 4k + (l >> 4)
// hits bank (2 (l & 15) + (l >> 4) + const) mod 32: all 32 lanes of each half distinct.
// For the A^T A product the A and B operand of tile (ti, tj) are the SAME register layout
// (lane l holds feature row l & 15 of k-slice l >> 4), so one LDS read feeds both sides.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace symode {

typedef double double4_t __attribute__((ext_vector_type(4)));

template <class Lib>
struct GramShape {
    static constexpr int F = Lib::P + Lib::D;          // features incl. the dx columns
    static constexpr int T = (F + 15) / 16;            // 16-wide tiles per side
    static constexpr int NPAIR = T * (T + 1) / 2;      // upper-triangular tile pairs
    static constexpr int FT = T * 16;
    static constexpr int PS = 66;                      // LDS row stride (floats), = 2 mod 32
    static constexpr int LDS_PER_WAVE = FT * PS;       // floats
    static constexpr int PARTIAL = NPAIR * 256;        // doubles per wave partial
};

template <class Lib>
__global__ __launch_bounds__(BLOCK) void aug_gram_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                         long N, const int* __restrict__ idx,
                                                         double* __restrict__ ws) {
    using G = GramShape<Lib>;
    constexpr int D = Lib::D, P = Lib::P, F = G::F, T = G::T, PS = G::PS;
    __shared__ float lds[(BLOCK / WAVE) * G::LDS_PER_WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    float* my = lds + wave * G::LDS_PER_WAVE;
    const long s = blockIdx.y;
    // idx == nullptr: problem s owns rows [s*N, (s+1)*N) of x / dx.
    // idx != nullptr: every problem draws its N points from ONE shared (x, dx) through its row of the
    // index table (seed sweeps over random subsamples of the same data set, run_scripts/*.sh seed loops).
    const float* xs = idx ? x : x + s * N * D;
    const float* ys = idx ? dx : dx + s * N * D;
    const int* is = idx ? idx + s * N : nullptr;

    // zero the padding rows once (features F..FT-1 stay zero for the whole kernel)
    for (int f = F; f < G::FT; ++f) my[f * PS + lane] = 0.0f;

    double4_t acc[G::NPAIR];
#pragma unroll
    for (int i = 0; i < G::NPAIR; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};

    const long step = (long)gridDim.x * BLOCK;
    const long nsteps = (N + step - 1) / step;          // uniform across the grid: barriers are safe
    for (long it = 0; it < nsteps; ++it) {
        const long n = it * step + (long)blockIdx.x * BLOCK + threadIdx.x;
        float feat[F];
        if (n < N) {
            float xp[D], yp[D], th[P];
            const long src = is ? (long)is[n] : n;
            load_point<D>(xs, src, xp);
            load_point<D>(ys, src, yp);
            Lib::eval(xp, th);
#pragma unroll
            for (int k = 0; k < P; ++k) feat[k] = th[k];
#pragma unroll
            for (int j = 0; j < D; ++j) feat[P + j] = yp[j];
        } else {
#pragma unroll
            for (int k = 0; k < F; ++k) feat[k] = 0.0f;
        }
        __syncthreads();                                 // previous step's operand reads are done
#pragma unroll
        for (int k = 0; k < F; ++k) my[k * PS + lane] = feat[k];
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < WAVE / 4; ++ks) {
            double a[T];
#pragma unroll
            for (int t = 0; t < T; ++t) a[t] = (double)my[(16 * t + (lane & 15)) * PS + 4 * ks + (lane >> 4)];
            int pidx = 0;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = ti; tj < T; ++tj) {
                    acc[pidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], a[tj], acc[pidx], 0, 0, 0);
                    ++pidx;
                }
        }
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
    // The four waves' tiles are combined in fixed order through LDS (the operand staging area is free now),
    // so a workgroup leaves ONE partial: element e = row*16 + col of tile pair q at ws[block][q*256 + e].
    __syncthreads();
    double* comb = reinterpret_cast<double*>(lds);
    static_assert(sizeof(lds) >= sizeof(double) * (BLOCK / WAVE) * G::NPAIR * 256, "LDS too small for the wave combine");
#pragma unroll
    for (int q = 0; q < G::NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            comb[(wave * G::NPAIR + q) * 256 + row * 16 + col] = acc[q][r];
        }
    __syncthreads();
    double* dst = ws + ((long)blockIdx.y * gridDim.x + blockIdx.x) * G::PARTIAL;
    for (int q = 0; q < G::NPAIR; ++q) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) v += comb[(w * G::NPAIR + q) * 256 + threadIdx.x];
        dst[q * 256 + threadIdx.x] = v;
    }
}

// Sum the n_part per-wave partials of problem s in fixed order and scatter into the dense
// symmetric (F, F) matrix.
template <class Lib>
__global__ __launch_bounds__(BLOCK) void gram_finalize_kernel(const double* __restrict__ ws, int n_part,
                                                              double* __restrict__ gram) {
    using G = GramShape<Lib>;
    constexpr int F = G::F, T = G::T;
    const long s = blockIdx.x;
    const double* src = ws + s * (long)n_part * G::PARTIAL;
    double* out = gram + s * (long)F * F;
    const int e = threadIdx.x, row = e >> 4, col = e & 15;
    int q = 0;
    for (int ti = 0; ti < T; ++ti)
        for (int tj = ti; tj < T; ++tj, ++q) {
            double v = 0.0;
            int i = 0;
            for (; i + 8 <= n_part; i += 8) {                  // 8 independent loads in flight, added in fixed order
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(long)(i + u) * G::PARTIAL + q * 256 + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; i < n_part; ++i) v += src[(long)i * G::PARTIAL + q * 256 + e];
            const int R = 16 * ti + row, C = 16 * tj + col;
            if (R < F && C < F) {
                out[R * F + C] = v;
                if (ti != tj) out[C * F + R] = v;
            }
        }
}

template <class Lib>
hipError_t launch_aug_gram(const float* x, const float* dx, long S, long n, const int* idx, double* gram, double* ws,
                           int gx, hipStream_t st) {
    double* part = ws + WS_HEADER_DOUBLES;             // the header (magic, tickets) belongs to the one-launch reductions
    aug_gram_kernel<Lib><<<dim3(gx, (unsigned)S), dim3(BLOCK), 0, st>>>(x, dx, n, idx, part);
    SYMODE_LAUNCH_CHECK();
    gram_finalize_kernel<Lib><<<dim3((unsigned)S), dim3(BLOCK), 0, st>>>(part, gx, gram);
    SYMODE_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace symode
