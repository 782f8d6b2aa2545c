// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (tuning tool, not product code): (1) the lane -> element maps of A, B and
// C/D, found with one-hot operands; (2) its issue rate beside v_mfma_f64_16x16x4_f64 and v_fma_f64, at 1 / 2 / 4 waves
// per SIMD.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/micro/mfma_f64_probe tools/micro/mfma_f64_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void onehot_kernel(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[((long)la * 64 + lb) * 64 + lane] = d;
        }
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    double a = seed + lane, b = seed - lane;
    if constexpr (KIND == 0) {                           // 4x4x4, eight independent accumulators
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
        }
        double s = 0;
        for (int k = 0; k < 8; ++k) s += c[k];
        if (s == 12345.678) out[blockIdx.x] = s;
    } else if constexpr (KIND == 1) {                    // 16x16x4, four independent accumulators
        d4 c[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
        }
        double s = 0;
        for (int k = 0; k < 4; ++k) s += c[k].x + c[k].y + c[k].z + c[k].w;
        if (s == 12345.678) out[blockIdx.x] = s;
    } else if constexpr (KIND == 3) {                    // both pipes from one wave: 8 x 4x4x4 and 32 x v_fma_f64 per pass
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[16];
        for (int k = 0; k < 16; ++k) v[k] = k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
                v[2 * k] = __builtin_fma(a, b, v[2 * k]);
                v[2 * k + 1] = __builtin_fma(a, b, v[2 * k + 1]);
                v[(2 * k + 8) & 15] = __builtin_fma(b, a, v[(2 * k + 8) & 15]);
                v[(2 * k + 9) & 15] = __builtin_fma(b, a, v[(2 * k + 9) & 15]);
            }
        }
        double s = 0;
        for (int k = 0; k < 8; ++k) s += c[k];
        for (int k = 0; k < 16; ++k) s += v[k];
        if (s == 12345.678) out[blockIdx.x] = s;
    } else {                                             // v_fma_f64, sixteen independent accumulators
        double c[16];
        for (int k = 0; k < 16; ++k) c[k] = k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = __builtin_fma(a, b, c[k]);
        }
        double s = 0;
        for (int k = 0; k < 16; ++k) s += c[k];
        if (s == 12345.678) out[blockIdx.x] = s;
    }
}

template <int KIND>
static void rate(const char* what, int per_iter, double fma_per_instr, double* out) {
    const int iters = 20000;
    for (int wg_per_cu : {1, 2, 4}) {                    // 256 threads = 1 wave per SIMD per workgroup
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int grid = 256 * wg_per_cu;
        rate_kernel<KIND><<<grid, 256>>>(out, 10, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        rate_kernel<KIND><<<grid, 256>>>(out, iters, 1.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * per_iter * wg_per_cu;          // 1024 SIMDs, one wave each per workgroup
        const double ns_per_instr = ms * 1e6 / instr_per_simd;
        const double tfma = instr_per_simd * 1024 * fma_per_instr / (ms * 1e-3) * 1e-12;
        printf("%-28s %d wave(s)/SIMD: %.2f ns per instruction and SIMD (%.1f cycles at 2.4 GHz), %.1f T fp64 FMA/s\n", what, wg_per_cu,
               ns_per_instr, ns_per_instr * 2.4, tfma);
    }
}

int main() {
    double* out;
    CK(hipMalloc(&out, 64L * 64 * 64 * 8));
    onehot_kernel<<<1, 64>>>(out);
    CK(hipDeviceSynchronize());
    std::vector<double> h(64L * 64 * 64);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    // for every output lane: which (la, lb) pairs reach it
    printf("# v_mfma_f64_4x4x4_4b_f64: output lane <- list of (A lane, B lane) pairs whose product it sums\n");
    for (int lo = 0; lo < 64; ++lo) {
        printf("D lane %2d:", lo);
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (h[((long)la * 64 + lb) * 64 + lo] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    rate<0>("v_mfma_f64_4x4x4_4b_f64", 8, 256, out);
    rate<1>("v_mfma_f64_16x16x4_f64", 4, 1024, out);
    rate<2>("v_fma_f64", 16, 64, out);
    // one "instruction" here = one 4x4x4 (256 FMAs) + four v_fma_f64 (256 FMAs): 512 FMAs
    rate<3>("4x4x4 + 4 v_fma_f64 interleaved", 8, 512, out);
    return 0;
}
