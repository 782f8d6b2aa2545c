// A/B harness for the closure kernels (tuning runs, not product code): loss_grad and the fused residual + reversed
// regulariser closure straight from csrc/kernels.hpp, one big problem (S = 1, 2^26 points) and the bench's batched shape
// (8192 x 125 000 points), over launch widths and ring depths -- one process, so variants see the same box.  (The run
// kept as profiles/r03_closure_ab.txt also held the two-chunk form the ring replaced: ring=0 there.)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I symmetry-ode-discovery_amd/csrc
//              -o tools/micro/closure_ab tools/micro/closure_ab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"
using namespace symode;
using L3 = Library<2, 3, 0>;
using L5 = Library<2, 5, 0>;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ void fill(float* a, long n, unsigned seed, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        a[i] = scale * ((float)(h & 0xffffff) / 8388608.0f - 1.0f);
    }
}

template <typename F>
static double time_us(F launch, int reps, int rounds = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<double> t;
    for (int r = 0; r < rounds; ++r) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3 / reps);
    }
    return *std::min_element(t.begin(), t.end());
}

int main(int argc, char** argv) {
    const long S = 8192, NB = 125000, N1 = 1L << 26;
    const long NT = S * NB;                       // 1.024e9 points >= 2^26
    constexpr int D = 2;
    float *x, *dx, *gx, *jgx, *xi3, *xi5, *loss, *grad;
    double* ws;
    CK(hipMalloc(&x, NT * D * 4));
    CK(hipMalloc(&dx, NT * D * 4));
    CK(hipMalloc(&gx, NT * D * 4));
    CK(hipMalloc(&jgx, NT * D * D * 4));
    CK(hipMalloc(&xi3, S * D * L3::P * 4));
    CK(hipMalloc(&xi5, S * D * L5::P * 4));
    CK(hipMalloc(&loss, S * 2 * 4));
    CK(hipMalloc(&grad, S * D * L5::P * 4));
    const long max_rows = 4 * S;                  // partial rows of the widest launch
    CK(hipMalloc(&ws, (WS_HEADER_DOUBLES + max_rows * (2 + D * L5::P)) * 8));
    fill<<<4096, 256>>>(x, NT * D, 1u, 0.7f);
    fill<<<4096, 256>>>(dx, NT * D, 2u, 0.5f);
    fill<<<4096, 256>>>(gx, NT * D, 3u, 0.7f);
    fill<<<4096, 256>>>(jgx, NT * D * D, 4u, 1.0f);
    fill<<<64, 256>>>(xi3, S * D * L3::P, 5u, 0.3f);
    fill<<<64, 256>>>(xi5, S * D * L5::P, 6u, 0.3f);
    workspace_init_kernel<0><<<dim3((WS_HEADER_DOUBLES + BLOCK - 1) / BLOCK), dim3(BLOCK)>>>((unsigned long long*)ws, WS_HEADER_DOUBLES);
    CK(hipDeviceSynchronize());
    double* part = ws + WS_HEADER_DOUBLES;
    Finish fin{(unsigned long long*)ws, nullptr, 1.0f, 2.0f, loss, grad, 1, 1};
    Finish fin2 = fin;
    fin2.n_loss = 2;

    printf("# one problem, N = %ld points (d = 2)\n", N1);
    const int grids[] = {128, 256, 512, 768, 1024, 2048};
    for (int g : grids) {
        double us = time_us([&] { loss_grad_kernel<L3, true><<<dim3(g, 1), dim3(BLOCK)>>>(x, dx, N1, true, xi3, nullptr, part, fin, true); }, 10);
        printf("loss_grad o3 ring4 seg grid=%d: %.1f us %.0f GB/s\n", g, us, N1 * 16.0 / us * 1e-3);
        us = time_us([&] { loss_grad_kernel<L5, true><<<dim3(g, 1), dim3(BLOCK)>>>(x, dx, N1, true, xi5, nullptr, part, fin, true); }, 10);
        printf("loss_grad o5 ring4 seg grid=%d: %.1f us %.0f GB/s\n", g, us, N1 * 16.0 / us * 1e-3);
    }
#define REV1(LIB, MSEF, RG, BPP)                                                                                               \
    for (int g : grids) {                                                                                                      \
        const double us = time_us([&] {                                                                                        \
            symreg_reversed_kernel<LIB, MSEF, RG><<<dim3(g, 1), dim3(BLOCK)>>>(x, MSEF ? dx : nullptr, gx, jgx, 1, N1, true,    \
                                                                                (LIB::P == 10 ? xi3 : xi5), nullptr, 0.1f, part, \
                                                                                MSEF ? fin2 : fin);                              \
        }, 10);                                                                                                                 \
        printf("symreg_reversed p=%d mse=%d ring=%d grid=%d: %.1f us %.0f GB/s\n", LIB::P, (int)MSEF, RG, g, us, N1 * BPP / us * 1e-3); \
    }
    REV1(L3, false, 2, 32.0)
    REV1(L3, false, 3, 32.0)
    REV1(L5, true, 2, 40.0)
    REV1(L5, true, 3, 40.0)

    printf("# batched: %ld problems x %ld points\n", S, NB);
    const int gxs[] = {1, 2, 4};
#define REVB(RG)                                                                                                                \
    for (int g : gxs) {                                                                                                         \
        const double us = time_us([&] {                                                                                         \
            symreg_reversed_kernel<L5, true, RG><<<dim3(g, (unsigned)S), dim3(BLOCK)>>>(x, dx, gx, jgx, 1, NB, true, xi5, nullptr, \
                                                                                       0.1f, part, fin2);                         \
        }, 5);                                                                                                                   \
        printf("closure o5 batched ring=%d grid.x=%d: %.1f us %.0f GB/s (%.3f of 8 TB/s)\n", RG, g, us, NT * 40.0 / us * 1e-3,   \
               NT * 40.0 / us * 1e-3 / 8000);                                                                                   \
    }
    REVB(2)
    REVB(3)
    // Xi of the order-5 closure in VGPRs instead of SGPRs (42 more registers: 2 waves per SIMD instead of 3)
    for (int g : gxs) {
        const double us = time_us([&] {
            symreg_reversed_kernel<L5, true, 2, 1000><<<dim3(g, (unsigned)S), dim3(BLOCK)>>>(x, dx, gx, jgx, 1, NB, true, xi5, nullptr,
                                                                                          0.1f, part, fin2);
        }, 5);
        printf("closure o5 batched ring=2 Xi in VGPRs grid.x=%d: %.1f us %.0f GB/s (%.3f of 8 TB/s)\n", g, us, NT * 40.0 / us * 1e-3,
               NT * 40.0 / us * 1e-3 / 8000);
    }
    for (int g : gxs) {
        const double us = time_us([&] { loss_grad_kernel<L5, true><<<dim3(g, (unsigned)S), dim3(BLOCK)>>>(x, dx, NB, true, xi5, nullptr, part, fin, false); }, 5);
        printf("loss_grad o5 batched ring4 grid.x=%d: %.1f us %.0f GB/s\n", g, us, NT * 16.0 / us * 1e-3);
    }
    CK(hipDeviceSynchronize());
    float hl;
    CK(hipMemcpy(&hl, grad, 4, hipMemcpyDeviceToHost));
    printf("# done (grad[0] = %g)\n", hl);
    return 0;
}
