// Which pipe should carry the closure's per-point arithmetic on gfx950?  A synthetic point loop with the shape of the
// fused closure at order 5 (d = 2, p = 21): two "apply" products h = W th, hg = W thg (W wave-uniform) and the gradient
// accumulation acc[j][k] += a_j th_k - u_j thg_k, in four forms:
//   S   scalar fp32 VALU: 84 v_fmac with an SGPR coefficient + 84 with VGPR operands          (what kernels.hpp issues)
//   P   packed over the two equation rows: 42 + 42 v_pk_fma_f32 (same roundings, same order)
//   M   the applies as 42 v_mfma_f32_4x4x1_16b_f32 (A = column k of W in lanes l%4 < 2, B = th_k: row i of block b of the
//       result lands in lane 4b + j -- the lane's own point), gradient scalar
//   MP  applies on the matrix pipe, gradient packed
// Prints ns per point per lane-slot (1024 SIMDs x 64 lanes) at 2..4 resident waves per SIMD, and checks that the four
// forms compute the same h.  Tuning probe, not product code.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -o tools/micro/issue_probe tools/micro/issue_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int P = 21;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ f2 splat(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ w, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    float W[2][P];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < P; ++k) W[j][k] = w[j * P + k];                  // uniform -> SGPRs
    float th[P], thg[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        th[k] = 0.01f * (float)(threadIdx.x % 13) + 0.1f * k;
        thg[k] = 0.02f * (float)(threadIdx.x % 7) - 0.05f * k;
    }
    float acc[2][P];
    f2 acc2[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        acc[0][k] = acc[1][k] = 0.0f;
        acc2[k] = splat(0.0f);
    }
    // MFMA A operand: lane l supplies W[l % 4][k] for l % 4 < 2, else 0
    float Acol[P];
#pragma unroll
    for (int k = 0; k < P; ++k) Acol[k] = (lane & 3) < 2 ? w[(lane & 3) * P + k] : 0.0f;
    float hsum = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float h[2], hg[2];
        if constexpr (MODE == 0 || MODE == 1) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s = 0.0f, t = 0.0f;
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        s = fmaf(W[j][k], th[k], s);
                        t = fmaf(W[j][k], thg[k], t);
                    }
                    h[j] = s;
                    hg[j] = t;
                }
            } else {
                f2 s = splat(0.0f), t = splat(0.0f);
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const f2 w2 = f2{W[0][k], W[1][k]};
                    s = fma2(w2, splat(th[k]), s);
                    t = fma2(w2, splat(thg[k]), t);
                }
                h[0] = s.x; h[1] = s.y; hg[0] = t.x; hg[1] = t.y;
            }
        } else {
            f4 c = {0.f, 0.f, 0.f, 0.f}, cg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < P; ++k) {
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(Acol[k], th[k], c, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_4x4x1f32(Acol[k], thg[k], cg, 0, 0, 0);
            }
            h[0] = c.x; h[1] = c.y; hg[0] = cg.x; hg[1] = cg.y;
        }
        const float a0 = h[0] - hg[1], a1 = h[1] + hg[0], u0 = hg[0] * 0.5f, u1 = hg[1] * 0.25f;
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                acc[0][k] = fmaf(a0, th[k], fmaf(-u0, thg[k], acc[0][k]));
                acc[1][k] = fmaf(a1, th[k], fmaf(-u1, thg[k], acc[1][k]));
            }
        } else {
            const f2 a2 = {a0, a1}, u2 = {-u0, -u1};
#pragma unroll
            for (int k = 0; k < P; ++k) acc2[k] = fma2(a2, splat(th[k]), fma2(u2, splat(thg[k]), acc2[k]));
        }
        th[0] = fmaf(h[0], 1e-9f, th[0]);                                  // loop-carried: nothing hoists
        thg[0] = fmaf(hg[1], 1e-9f, thg[0]);
        hsum += h[0] + h[1];
    }
    float s = hsum;
#pragma unroll
    for (int k = 0; k < P; ++k) s += acc[0][k] + acc[1][k] + acc2[k].x + acc2[k].y;
    out[(long)blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x < 8 && iters == 1) out[(long)gridDim.x * 256 + threadIdx.x] = hsum;
}

template <int MODE>
static void run(const char* name, const float* w, float* out) {
    const int iters = 2000;
    for (int waves = 2; waves <= 4; ++waves) {
        const int blocks = 256 * waves;                                   // 4 waves per block, 1024 SIMDs
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        probe<MODE><<<blocks, 256>>>(w, out, iters);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(a));
            probe<MODE><<<blocks, 256>>>(w, out, iters);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        // a SIMD runs `waves` waves x iters points-per-lane: time per point per lane-slot
        printf("%-3s waves/SIMD=%d  %.3f ms  %.1f ns per point (per wave, per SIMD)\n", name, waves, best, best * 1e6 / (iters * waves));
    }
}

int main() {
    float *w, *out;
    CK(hipMalloc(&w, 2 * P * 4));
    CK(hipMalloc(&out, (256L * 4 * 256 + 64) * 4));
    float hw[2 * P];
    for (int i = 0; i < 2 * P; ++i) hw[i] = 0.05f * (i % 9) - 0.2f;
    CK(hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice));
    // same h from all four forms? (one iteration, first 8 lanes)
    float ref[8];
    for (int m = 0; m < 4; ++m) {
        if (m == 0) probe<0><<<1, 256>>>(w, out, 1);
        if (m == 1) probe<1><<<1, 256>>>(w, out, 1);
        if (m == 2) probe<2><<<1, 256>>>(w, out, 1);
        if (m == 3) probe<3><<<1, 256>>>(w, out, 1);
        CK(hipDeviceSynchronize());
        float got[8];
        CK(hipMemcpy(got, out + 256, sizeof(got), hipMemcpyDeviceToHost));
        if (m == 0) for (int i = 0; i < 8; ++i) ref[i] = got[i];
        float err = 0;
        for (int i = 0; i < 8; ++i) err = fmaxf(err, fabsf(got[i] - ref[i]));
        printf("# form %d: h sum lane 0..3 = %g %g %g %g, max |diff| vs scalar form %g\n", m, got[0], got[1], got[2], got[3], err);
    }
    run<0>("S", w, out);
    run<1>("P", w, out);
    run<2>("M", w, out);
    run<3>("MP", w, out);
    return 0;
}
