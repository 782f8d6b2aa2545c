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

// Block-level reduction of NACC per-thread accumulators.
// Stage 1: butterfly inside each wave.  Stage 2: lane 0 of every wave parks its NACC sums in
// LDS as [wave][NACC]; thread k < NACC then adds the waves in fixed order (deterministic)
// and hands the block's k-th sum to `emit(k, value)`.
template <int NACC, int BLOCK, typename Emit>
__device__ __forceinline__ void block_reduce_emit(float (&acc)[NACC], float* lds /* [BLOCK/64][NACC] */, Emit emit) {
    constexpr int NW = BLOCK / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) lds[wave * NACC + k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NACC; k += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += (double)lds[w * NACC + k];
        emit(k, s);
    }
}

}  // namespace symode
