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

// The same total, BIT-IDENTICAL to the butterfly above (same partners, same order: lane ^ 32, 16, 8, 4, 2, 1; fp addition
// commutes, so both partners of a step hold the same bits), without its six ds_bpermute round trips -- for the serial
// dot products of the L-BFGS recursion (lbfgs.hip), where the reduction IS the critical path:
//   lane ^ 32, lane ^ 16   gfx950's v_permlane32_swap / v_permlane16_swap with both operands = v: the two results are
//                          (lower half, lower half) / (upper half, upper half) resp. (even rows, odd rows), their sum is
//                          the butterfly step;
//   lane ^ 8               DPP row_ror:8;
//   lane ^ 4               DPP row_shl:4 into the quads with bit 2 clear, row_shr:4 into the others (bank masks);
//   lane ^ 2, lane ^ 1     DPP quad permutes.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK_MASK, false));
}

template <int CTRL>
__device__ __forceinline__ float dpp_full(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(h[0]) + __uint_as_float(h[1]);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    // (full-mask permutes are written with old = 0 / bound_ctrl so that the compiler folds them into ONE v_add_f32_dpp)
    v += dpp_full<0x128>(v);                                          // row_ror:8
    v += dpp_take<0x114, 0xa>(dpp_take<0x104, 0x5>(v, v), v);         // row_shl:4 -> quads 0, 2; row_shr:4 -> quads 1, 3
    v += dpp_full<0x4E>(v);                                           // quad_perm:[2,3,0,1]
    v += dpp_full<0xB1>(v);                                           // quad_perm:[1,0,3,2]
    return v;
}

// ... and as a scalar: every lane holds the total, the compiler just cannot know; handing it back through
// v_readfirstlane keeps everything computed from it (step coefficients, loop bounds, branch conditions) on the scalar side.
__device__ __forceinline__ float wave_sum_dpp_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_sum_dpp(v))));
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
