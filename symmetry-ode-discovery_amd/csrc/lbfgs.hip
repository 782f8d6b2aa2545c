// L-BFGS two-loop recursion for S independent small problems (seed sweeps): one wave per problem.
//
// torch.optim.LBFGS builds its search direction with ~4 tiny tensor ops per history pair
// (torch/optim/lbfgs.py, "compute the approximate (L-BFGS) inverse Hessian multiplied by the
// gradient"); for S problems of n <= 256 parameters that is hundreds of launches per iteration.
// Here problem s is one wavefront: lane l owns components l, l+64, l+128, l+192; the curvature pairs
// live in ring buffers (S, H, n) with per-problem head / count; every dot product is a wave reduction.
#include <hip/hip_runtime.h>

#include "../../include/symode.h"
#include "reduce.hpp"

namespace symode {

constexpr int LB_MAXC = 4;      // components per lane -> n <= 256
constexpr int LB_MAXH = 128;    // history slots

__global__ __launch_bounds__(WAVE) void lbfgs_direction_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ old_dirs,
                                                               const float* __restrict__ old_stps,
                                                               const float* __restrict__ ro,
                                                               const long* __restrict__ head,
                                                               const long* __restrict__ count,
                                                               const float* __restrict__ h_diag, int n, int H,
                                                               float* __restrict__ d_out) {
    __shared__ float al[LB_MAXH];
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    const int m = (int)count[s], h0 = (int)head[s];
    const float* Y = old_dirs + s * (long)H * n;
    const float* Sx = old_stps + s * (long)H * n;
    const float* R = ro + s * (long)H;
    float q[LB_MAXC];
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        q[c] = i < n ? -g[s * n + i] : 0.0f;
    }
    for (int k = m - 1; k >= 0; --k) {                       // newest -> oldest
        const int slot = (h0 + k) % H;
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) part = fmaf(Sx[slot * n + i], q[c], part);
        }
        const float a = wave_sum(part) * R[slot];
        if (lane == 0) al[k] = a;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) q[c] = fmaf(-a, Y[slot * n + i], q[c]);
        }
    }
    __syncthreads();
    const float hd = h_diag[s];
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) q[c] *= hd;            // r = q * H_diag
    for (int k = 0; k < m; ++k) {                            // oldest -> newest
        const int slot = (h0 + k) % H;
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) part = fmaf(Y[slot * n + i], q[c], part);
        }
        const float be = wave_sum(part) * R[slot];
        const float coef = al[k] - be;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) q[c] = fmaf(coef, Sx[slot * n + i], q[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) d_out[s * n + i] = q[c];
    }
}

}  // namespace symode

extern "C" int symode_lbfgs_direction(const float* g, const float* old_dirs, const float* old_stps, const float* ro,
                                      const long* head, const long* count, const float* h_diag, long n_problems, int n,
                                      int history, float* d_out, void* stream) {
    using namespace symode;
    if (n_problems < 1 || n < 1 || n > WAVE * LB_MAXC || history < 1 || history > LB_MAXH) return SYMODE_E_BADSIZE;
    if (!g || !old_dirs || !old_stps || !ro || !head || !count || !h_diag || !d_out) return SYMODE_E_NULLPTR;
    lbfgs_direction_kernel<<<dim3((unsigned)n_problems), dim3(WAVE), 0, (hipStream_t)stream>>>(
        g, old_dirs, old_stps, ro, head, count, h_diag, n, history, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}
