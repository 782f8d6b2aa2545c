// A/B harness for the closure kernels (tuning runs, not product code): loss_grad and the fused residual + reversed
// regulariser closure straight from csrc/kernels.hpp, one big problem (S = 1, 2^26 points) and the bench's batched shape
// (8192 x 125 000 points), over launch widths and ring depths -- one process, so variants see the same box.  (The run
// kept as profiles/r03_closure_ab.txt also held the two-chunk form the ring replaced: ring=0 there; the problem-walking
// form of the closure measured in profiles/r03_closure_walk.txt lives in commit becc803 only.)
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

// The fused closure's memory side alone: the same four streams (x, dx, g(x) 8 B/point each, J_g 16 B/point), the same
// 16-byte non-temporal chunk loads in the same two-slot ring on the same (grid.x, problems) launch -- and six adds per
// chunk instead of ~880 vector instructions.  What this reaches is the HBM ceiling of the closure's access pattern.
__global__ __launch_bounds__(BLOCK) void stream_only_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                            const float* __restrict__ gx, const float* __restrict__ jgx, long N,
                                                            float* __restrict__ out) {
    const long s = blockIdx.y;
    const float* xs = x + s * N * 2;
    const float* ys = dx + s * N * 2;
    const float* gs = gx + s * N * 2;
    const float* js = jgx + s * N * 4;
    const long tid = (long)blockIdx.x * BLOCK + threadIdx.x, nthreads = (long)gridDim.x * BLOCK;
    float acc = 0.0f;
    chunk_ring<2, 5>(
        N / 2, tid, nthreads,
        [&](long q, float4 (&slot)[5]) {
            float4 a[1], b[1], c[1];
            load_chunk_raw<2, true>(xs, q, a);
            load_chunk_raw<2, true>(ys, q, b);
            load_chunk_raw<2, true>(gs, q, c);
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v* jq = reinterpret_cast<const f4v*>(js) + q * 2;
            const f4v j0 = __builtin_nontemporal_load(jq), j1 = __builtin_nontemporal_load(jq + 1);
            slot[0] = a[0];
            slot[1] = b[0];
            slot[2] = c[0];
            slot[3] = make_float4(j0.x, j0.y, j0.z, j0.w);
            slot[4] = make_float4(j1.x, j1.y, j1.z, j1.w);
        },
        [&](long, const float4 (&slot)[5]) {
#pragma unroll
            for (int i = 0; i < 5; ++i) acc += slot[i].x + slot[i].y + slot[i].z + slot[i].w;
        });
    if (acc == 123456.789f) out[s * gridDim.x + blockIdx.x] = acc;      // (never true: keeps the loads alive without a store stream)
}

// ... and the same bytes as ONE flat stream over all problems (they are contiguous): G long-lived workgroups, ring depth R,
// SLAB = true: every workgroup walks its own contiguous slab instead of striding through the whole array.
template <int R, bool SLAB>
__global__ __launch_bounds__(BLOCK) void stream_flat_kernel(const float* __restrict__ x, const float* __restrict__ dx,
                                                            const float* __restrict__ gx, const float* __restrict__ jgx, long NT_,
                                                            float* __restrict__ out) {
    const long nchunks_all = NT_ / 2;
    long nchunks = nchunks_all, c0 = (long)blockIdx.x * BLOCK + threadIdx.x, stride = (long)gridDim.x * BLOCK;
    if (SLAB) {
        const long per = (nchunks_all + gridDim.x - 1) / gridDim.x, lo = (long)blockIdx.x * per;
        nchunks = lo + per < nchunks_all ? lo + per : nchunks_all;
        c0 = lo + threadIdx.x;
        stride = BLOCK;
    }
    float acc = 0.0f;
    chunk_ring<R, 5>(
        nchunks, c0, stride,
        [&](long q, float4 (&slot)[5]) {
            float4 a[1], b[1], c[1];
            load_chunk_raw<2, true>(x, q, a);
            load_chunk_raw<2, true>(dx, q, b);
            load_chunk_raw<2, true>(gx, q, c);
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v* jq = reinterpret_cast<const f4v*>(jgx) + q * 2;
            const f4v j0 = __builtin_nontemporal_load(jq), j1 = __builtin_nontemporal_load(jq + 1);
            slot[0] = a[0];
            slot[1] = b[0];
            slot[2] = c[0];
            slot[3] = make_float4(j0.x, j0.y, j0.z, j0.w);
            slot[4] = make_float4(j1.x, j1.y, j1.z, j1.w);
        },
        [&](long, const float4 (&slot)[5]) {
#pragma unroll
            for (int i = 0; i < 5; ++i) acc += slot[i].x + slot[i].y + slot[i].z + slot[i].w;
        });
    if (acc == 123456.789f) out[blockIdx.x] = acc;
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
        const double us = time_us([&] { stream_only_kernel<<<dim3(g, (unsigned)S), dim3(BLOCK)>>>(x, dx, gx, jgx, NB, loss); }, 5);
        printf("the closure's four streams alone (ring=2, six adds per chunk) grid.x=%d: %.1f us %.0f GB/s (%.3f of 8 TB/s)\n", g, us,
               NT * 40.0 / us * 1e-3, NT * 40.0 / us * 1e-3 / 8000);
    }
    const int flat_grids[] = {256, 512, 768, 1024, 2048, 4096};
#define FLAT(RR, SL)                                                                                                              \
    for (int g : flat_grids) {                                                                                                    \
        const double us = time_us([&] { stream_flat_kernel<RR, SL><<<dim3(g), dim3(BLOCK)>>>(x, dx, gx, jgx, NT, loss); }, 5);     \
        printf("four streams, ONE flat pass, %s, ring=%d, %d workgroups: %.1f us %.0f GB/s (%.3f of 8 TB/s)\n",                      \
               SL ? "slab per workgroup" : "grid-stride", RR, g, us, NT * 40.0 / us * 1e-3, NT * 40.0 / us * 1e-3 / 8000);            \
    }
    FLAT(2, false)
    FLAT(4, false)
    FLAT(2, true)
    FLAT(4, true)
    // the same with the closure's residency: 52 KB of (unused) LDS per workgroup = 3 workgroups per CU = 3 waves per SIMD
    {
        const int grids2[] = {768, 1024, 4096, 8192, 16384};
        CK(hipFuncSetAttribute((const void*)stream_flat_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 53248));
        CK(hipFuncSetAttribute((const void*)stream_flat_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 53248));
        for (int g : grids2) {
            double us = time_us([&] { stream_flat_kernel<2, true><<<dim3(g), dim3(BLOCK), 53248>>>(x, dx, gx, jgx, NT, loss); }, 5);
            printf("four streams, flat, slab, 3 waves/SIMD, ring=2, %d workgroups: %.1f us %.0f GB/s (%.3f)\n", g, us, NT * 40.0 / us * 1e-3,
                   NT * 40.0 / us * 1e-3 / 8000);
            us = time_us([&] { stream_flat_kernel<4, true><<<dim3(g), dim3(BLOCK), 53248>>>(x, dx, gx, jgx, NT, loss); }, 5);
            printf("four streams, flat, slab, 3 waves/SIMD, ring=4, %d workgroups: %.1f us %.0f GB/s (%.3f)\n", g, us, NT * 40.0 / us * 1e-3,
                   NT * 40.0 / us * 1e-3 / 8000);
            us = time_us([&] { stream_flat_kernel<2, true><<<dim3(g), dim3(BLOCK)>>>(x, dx, gx, jgx, NT, loss); }, 5);
            printf("four streams, flat, slab, 8 waves/SIMD, ring=2, %d workgroups: %.1f us %.0f GB/s (%.3f)\n", g, us, NT * 40.0 / us * 1e-3,
                   NT * 40.0 / us * 1e-3 / 8000);
        }
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
