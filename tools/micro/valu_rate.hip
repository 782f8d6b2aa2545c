// VALU issue-rate probe for gfx950: wave-instructions per second per SIMD for v_fma_f32 and v_pk_fma_f32
// at 1..8 resident waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NA = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    if constexpr (MODE == 0) {
        float acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 1) {
        f2 acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
        const f2 a2 = {a, a + 1e-3f}, b2 = {b, b + 1e-3f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_elementwise_fma(acc[i], a2, b2);
        }
        f2 s = {0, 0};
#pragma unroll
        for (int i = 0; i < NA; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else if constexpr (MODE == 3) {   // the forward pattern: s = fma(w (SGPR), th[i] (VGPR), s (VGPR)) on 4 independent chains
        float th[NA], sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NA; ++i) th[i] = threadIdx.x * 3e-3f - i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) sacc[i & 3] = __builtin_fmaf(a + (float)i, th[i], sacc[i & 3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) th[i] += sacc[i] * 1e-9f;
        }
        out[blockIdx.x * 256 + threadIdx.x] = sacc[0] + sacc[1] + sacc[2] + sacc[3];
    } else {   // MODE 2: v_fmac with an SGPR operand and distinct VGPR sources (the loss/gradient pattern)
        float acc[NA], th[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) { acc[i] = threadIdx.x * 1e-3f + i; th[i] = threadIdx.x * 3e-3f - i; }
        float r = threadIdx.x * 1e-2f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_fmaf(r, th[i], acc[i]);
            r += a;
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int MODE>
void run(const char* name, int cus, float* out) {
    const int iters = 20000;
    for (int w = 1; w <= 8; ++w) {
        const int grid = cus * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double winstr = (double)iters * NA * w;                 // wave-instructions per SIMD
        const double flops = (double)iters * NA * (MODE == 1 ? 4 : 2) * 256.0 * grid;
        printf("%-14s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)  %.1f TFLOP/s\n", name, w, best,
               best * 1e6 / winstr, best * 1e6 / winstr * 2.4, flops / best / 1e9);
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    run<0>("v_fma_f32", cus, out);
    run<1>("v_pk_fma_f32", cus, out);
    run<2>("v_fmac(r,th)", cus, out);
    run<3>("v_fmac(s,th)", cus, out);
    return 0;
}
