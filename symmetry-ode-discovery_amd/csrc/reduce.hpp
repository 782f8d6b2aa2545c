// Wave64 / workgroup reduction helpers for the per-library-term sums.
#pragma once
#include <hip/hip_runtime.h>

namespace symode {

constexpr int WAVE = 64;

// Sum over the 64 lanes of a wave; every lane ends with the total (butterfly).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// Block-level reduction through an LDS transpose (no cross-lane network):
//   round r parks KC accumulators of all BLOCK threads as lds[kk][tid] (row stride BLOCK+1:
//   conflict-free writes and reads); thread (part, kk) adds BLOCK/PARTS consecutive partials in
//   fixed order; PARTS partial sums per value are combined by thread kk.  Deterministic.
template <int NACC, int BLOCK, typename Emit>
__device__ __forceinline__ void block_reduce_emit_lds(float (&acc)[NACC], float* lds, Emit emit) {
    constexpr int KC = 16;                      // values per round
    constexpr int PARTS = BLOCK / KC;           // 16 segments of BLOCK/PARTS = 16 threads' partials
    constexpr int SEG = BLOCK / PARTS;
    constexpr int STRIDE = BLOCK + 1;
    float* stage = lds;                          // [KC][STRIDE]
    float* part = lds + KC * STRIDE;             // [PARTS][KC]
    const int tid = threadIdx.x;
    const int kk = tid % KC, pp = tid / KC;
#pragma unroll
    for (int k0 = 0; k0 < NACC; k0 += KC) {
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (k0 + j < NACC) stage[j * STRIDE + tid] = acc[k0 + j];
        __syncthreads();
        float s = 0.0f;
        if (k0 + kk < NACC) {
#pragma unroll
            for (int i = 0; i < SEG; ++i) s += stage[kk * STRIDE + pp * SEG + i];
        }
        part[pp * KC + kk] = s;
        __syncthreads();
        if (tid < KC && k0 + tid < NACC) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) t += (double)part[q * KC + tid];
            emit(k0 + tid, t);
        }
    }
}

constexpr int reduce_lds_floats(int block) { return 16 * (block + 1) + (block / 16) * 16; }

// fp64 form of the same transpose reduction (per-thread fp64 partial sums: the VALU Gram kernel): KC = 8 values per
// round, thread (part, kk) adds BLOCK/PARTS consecutive threads' partials, thread kk combines the PARTS results.
template <int NACC, int BLOCK, typename Emit>
__device__ __forceinline__ void block_reduce_emit_lds_f64(double (&acc)[NACC], double* lds, Emit emit) {
    constexpr int KC = 8;
    constexpr int PARTS = BLOCK / KC;           // 32 segments of 8 threads' partials
    constexpr int SEG = BLOCK / PARTS;
    constexpr int STRIDE = BLOCK + 1;
    double* stage = lds;                         // [KC][STRIDE]
    double* part = lds + KC * STRIDE;            // [PARTS][KC]
    const int tid = threadIdx.x;
    const int kk = tid % KC, pp = tid / KC;
#pragma unroll
    for (int k0 = 0; k0 < NACC; k0 += KC) {
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (k0 + j < NACC) stage[j * STRIDE + tid] = acc[k0 + j];
        __syncthreads();
        double s = 0.0;
        if (k0 + kk < NACC) {
#pragma unroll
            for (int i = 0; i < SEG; ++i) s += stage[kk * STRIDE + pp * SEG + i];
        }
        part[pp * KC + kk] = s;
        __syncthreads();
        if (tid < KC && k0 + tid < NACC) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) t += part[q * KC + tid];
            emit(k0 + tid, t);
        }
    }
}

constexpr int reduce_lds_doubles(int block) { return 8 * (block + 1) + (block / 8) * 8; }

}  // namespace symode
