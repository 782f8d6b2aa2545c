// A/B harness for the streaming reduction kernels (tuning runs, not product code): instantiates variants of
// symreg_linear_kernel and vjp_kernel straight from csrc/kernels.hpp and times them with HIP events in ONE process, so
// that variants are compared on the same box in the same minute (boxes differ by 5 % and more).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I symmetry-ode-discovery_amd/csrc
//              -o tools/micro/stream_ab tools/micro/stream_ab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"
using namespace symode;
using Lib = Library<2, 3, 0>;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <typename F>
static double time_us(F launch, int reps = 10, int rounds = 5) {
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
    const long N = argc > 1 ? atol(argv[1]) : (1L << 26);
    constexpr int D = 2, P = Lib::P, NACC = 1 + D * P;
    float *x, *g, *gx, *xi, *L, *loss, *grad;
    double* ws;
    const long max_grid = 8192;
    CK(hipMalloc(&x, N * D * 4));
    CK(hipMalloc(&g, N * D * 4));
    CK(hipMalloc(&gx, N * D * 4));
    CK(hipMalloc(&xi, D * P * 4));
    CK(hipMalloc(&L, 4 * 4));
    CK(hipMalloc(&loss, 4));
    CK(hipMalloc(&grad, D * P * 4));
    CK(hipMalloc(&ws, (WS_HEADER_DOUBLES + max_grid * NACC) * 8));
    std::vector<float> h(N * D);
    for (long i = 0; i < N * D; ++i) h[i] = 0.5f * (float)((i * 2654435761u) & 0xffff) / 65536.0f - 0.25f;
    CK(hipMemcpy(x, h.data(), N * D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, h.data(), N * D * 4, hipMemcpyHostToDevice));
    float hxi[D * P];
    for (int i = 0; i < D * P; ++i) hxi[i] = 0.1f * (i % 7) - 0.3f;
    CK(hipMemcpy(xi, hxi, sizeof(hxi), hipMemcpyHostToDevice));
    const float hL[4] = {0.f, 1.f, -1.f, 0.f};
    CK(hipMemcpy(L, hL, sizeof(hL), hipMemcpyHostToDevice));
    workspace_init_kernel<0><<<dim3((WS_HEADER_DOUBLES + BLOCK - 1) / BLOCK), dim3(BLOCK)>>>((unsigned long long*)ws, WS_HEADER_DOUBLES);
    CK(hipDeviceSynchronize());
    double* part = ws + WS_HEADER_DOUBLES;
    Finish fin{(unsigned long long*)ws, nullptr, 1.0f, 2.0f, loss, grad, 1, 1};
    Finish fin_v = fin;
    fin_v.loss = nullptr;

    float *v, *gv;
    CK(hipMalloc(&v, N * D * 4));
    CK(hipMalloc(&gv, N * D * 4));
    CK(hipMemcpy(v, h.data(), N * D * 4, hipMemcpyHostToDevice));
    const int grids[] = {64, 128, 256, 384, 512, 768, 1024};
    const long sizes[] = {125000, 1000000, 4000000, 16000000, N};
    for (long n : sizes) {
        if (n > N) continue;
        printf("# n = %ld points, d = 2, order 3 (p = 10); us per launch (min of 5 rounds x 10)\n", n);
#define ROW(NAME, BYTES, LAUNCH)                                                  \
    {                                                                             \
        printf("%-28s", NAME);                                                    \
        double best = 1e30;                                                       \
        int bg = 0;                                                               \
        for (int gsz : grids) {                                                   \
            const double us = time_us([&] { LAUNCH; });                           \
            printf(" %8.1f", us);                                                 \
            if (us < best) { best = us; bg = gsz; }                               \
        }                                                                         \
        printf("   best grid %d: %.0f GB/s\n", bg, n * (BYTES) / best * 1e-3);    \
    }
        printf("%-28s", "grid");
        for (int gsz : grids) printf(" %8d", gsz);
        printf("\n");
        ROW("symreg_linear R=4", 8.0, (symreg_linear_kernel<Lib, 4><<<dim3(gsz), dim3(BLOCK)>>>(x, n, true, xi, nullptr, L, 1, part, fin)))
        ROW("symreg_linear R=2", 8.0, (symreg_linear_kernel<Lib, 2><<<dim3(gsz), dim3(BLOCK)>>>(x, n, true, xi, nullptr, L, 1, part, fin)))
        ROW("vjp grad_x R=3", 24.0, (vjp_kernel<Lib, true, 3><<<dim3(gsz), dim3(BLOCK)>>>(x, g, n, true, xi, nullptr, gx, part, fin_v)))
        ROW("vjp grad_x R=4", 24.0, (vjp_kernel<Lib, true, 4><<<dim3(gsz), dim3(BLOCK)>>>(x, g, n, true, xi, nullptr, gx, part, fin_v)))
        ROW("vjp no grad_x R=3", 16.0, (vjp_kernel<Lib, false, 3><<<dim3(gsz), dim3(BLOCK)>>>(x, g, n, true, xi, nullptr, nullptr, part, fin_v)))
        ROW("vjp no grad_x R=4", 16.0, (vjp_kernel<Lib, false, 4><<<dim3(gsz), dim3(BLOCK)>>>(x, g, n, true, xi, nullptr, nullptr, part, fin_v)))
        ROW("jvp_vjp g_out R=1", 48.0, (jvp_vjp_kernel<Lib, true, 1><<<dim3(gsz), dim3(BLOCK)>>>(x, v, g, g, n, true, xi, nullptr, gx, gv, part, fin_v)))
        ROW("jvp_vjp g_out R=2", 48.0, (jvp_vjp_kernel<Lib, true, 2><<<dim3(gsz), dim3(BLOCK)>>>(x, v, g, g, n, true, xi, nullptr, gx, gv, part, fin_v)))
        ROW("jvp_vjp g_out R=3", 48.0, (jvp_vjp_kernel<Lib, true, 3><<<dim3(gsz), dim3(BLOCK)>>>(x, v, g, g, n, true, xi, nullptr, gx, gv, part, fin_v)))
        ROW("jvp_vjp no g_out R=2", 40.0, (jvp_vjp_kernel<Lib, false, 2><<<dim3(gsz), dim3(BLOCK)>>>(x, v, nullptr, g, n, true, xi, nullptr, gx, gv, part, fin_v)))
        ROW("jvp_vjp no g_out R=3", 40.0, (jvp_vjp_kernel<Lib, false, 3><<<dim3(gsz), dim3(BLOCK)>>>(x, v, nullptr, g, n, true, xi, nullptr, gx, gv, part, fin_v)))
    }
    CK(hipDeviceSynchronize());
    float hl;
    CK(hipMemcpy(&hl, grad, 4, hipMemcpyDeviceToHost));
    printf("# done (grad[0] = %g)\n", hl);
    return 0;
}
