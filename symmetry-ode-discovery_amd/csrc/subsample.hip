// Seeded subsamples of a data set for the seed sweeps: for every seed the m-subset of range(n) that holds the m smallest of
// n keys key(seed, row) -- a counter-based hash, so a seed's subsample depends on that seed alone (not on the other seeds
// of the sweep, the world size or the device), rows ascending.  One workgroup per seed: three radix-select passes over the
// 32-bit keys (LDS histograms of 11 + 11 + 10 bits) find the m-th smallest key T and how many keys equal to T still
// belong to the subset; one ordered compaction pass (eight consecutive rows per thread, two block scans per 8192 rows)
// writes the rows with key < T (and would take ties in row order: the present key has none).  Keys are recomputed in every
// pass instead of being stored.
// replaces: the first batch of DataLoader(train_dataset, batch_size=int(len * lbfgs_subsample), shuffle=True) per seed
// (main.py:36-38) as an index table; the torch form it replaced (one generator launch per seed, a batched top-k and a
// sort) took 1.1-1.2 ms for 64 seeds of 10^5 rows, three quarters of the sequential-threshold sweep's wall time.
#include <hip/hip_runtime.h>

#include "../../include/symode.h"

namespace {

constexpr int SB = 1024;                                  // threads per workgroup: 16 waves

// key(seed, row) = fmix32((row ^ a) * 0x9E3779B1 + b), (a, b) = the two halves of a 64-bit mix of the seed: three 32-bit
// multiplies per row (the kernel recomputes every key in each of its four passes; a 64-bit mixer per row -- twelve
// quarter-rate multiplies -- made the hash most of its time: 103 us for 64 seeds of 10^5 rows).  Every step is a bijection
// of the 32-bit word, so for n <= 2^32 rows no two rows of a seed share a key.
__device__ __forceinline__ unsigned long long seed_words(unsigned long long seed) {
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ unsigned subsample_key(unsigned a, unsigned b, unsigned row) {
    unsigned h = (row ^ a) * 0x9E3779B1u + b;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(SB) void seeded_subsample_kernel(long n, long m, const long long* __restrict__ seeds,
                                                              int* __restrict__ out) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wave_tot[2][SB / 64];
    __shared__ unsigned sel_bin;
    __shared__ long sel_need;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long sw = seed_words((unsigned long long)seeds[blockIdx.x]);
    const unsigned ka = (unsigned)sw, kb = (unsigned)(sw >> 32);
    int* dst = out + (long)blockIdx.x * m;

    // radix select: after pass p the top bits of T are known (`prefix`), `need` = rank of T among the keys that share them
    unsigned prefix = 0;
    long need = m;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass], nb = 1 << widths[pass], hi = sh + widths[pass];
        for (int b = tid; b < 2048; b += SB) hist[b] = 0;
        __syncthreads();
        for (long i = tid; i < n; i += SB) {
            const unsigned k = subsample_key(ka, kb, (unsigned)i);
            if (hi >= 32 || (k >> hi) == (prefix >> hi)) atomicAdd(&hist[(k >> sh) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns bins [l * per, (l + 1) * per): its total, the wave's exclusive prefix, then the bin that holds rank `need`
            const int per = nb / 64;
            unsigned mine = 0;
            for (int j = 0; j < per; ++j) mine += hist[lane * per + j];
            unsigned incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            unsigned long long cum = incl - mine;                        // keys in bins before this lane's
            if ((long)cum < need && need <= (long)(cum + mine)) {        // exactly one lane
                int b = lane * per;
                while ((long)(cum + hist[b]) < need) {
                    cum += hist[b];
                    ++b;
                }
                sel_bin = (unsigned)b;
                sel_need = need - (long)cum;
            }
        }
        __syncthreads();
        prefix |= sel_bin << sh;
        need = sel_need;
        __syncthreads();
    }
    const unsigned T = prefix;                                            // the m-th smallest key; `need` keys equal to T belong

    // ordered compaction: RPT consecutive rows per thread and step (8192 rows per step), positions from a block scan of the
    // per-thread counts (wave scan by shuffles + the 16 wave totals through LDS) -- first of the ties, which must be taken
    // in row order, then of the chosen rows
    constexpr int RPT = 8;
    auto block_scan = [&](unsigned v, int buf, unsigned& total) -> unsigned {        // exclusive prefix of v over the workgroup
        unsigned incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wave_tot[buf][wave] = incl;
        __syncthreads();
        unsigned before = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < SB / 64; ++w) {
            const unsigned t = wave_tot[buf][w];
            before += w < wave ? t : 0u;
            total += t;
        }
        return before + incl - v;
    };
    long base = 0, ties_before = 0;
    for (long c0 = 0; c0 < n; c0 += (long)SB * RPT) {
        const long i0 = c0 + (long)tid * RPT;
        unsigned k[RPT];
        unsigned n_eq = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            k[r] = i0 + r < n ? subsample_key(ka, kb, (unsigned)(i0 + r)) : 0xFFFFFFFFu;
            n_eq += (i0 + r < n && k[r] == T) ? 1u : 0u;
        }
        unsigned eq_all, take_all;
        long eq_before = ties_before + block_scan(n_eq, 0, eq_all);
        bool take[RPT];
        unsigned n_take = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool in = i0 + r < n, eq = in && k[r] == T;
            take[r] = in && (k[r] < T || (eq && eq_before < need));
            eq_before += eq ? 1 : 0;
            n_take += take[r] ? 1u : 0u;
        }
        long pos = base + block_scan(n_take, 1, take_all);
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (take[r]) dst[pos++] = (int)(i0 + r);
        base += take_all;
        ties_before += eq_all;
        __syncthreads();                                                  // wave_tot is rewritten by the next step
    }
}

}  // namespace

extern "C" int symode_seeded_subsamples(long n, long m, const long long* seeds, int n_seeds, int* idx_out, void* stream) {
    if (n < 1 || n > 2147483647L || m < 1 || m > n || n_seeds < 1) return SYMODE_E_BADSIZE;
    if (!seeds || !idx_out) return SYMODE_E_NULLPTR;
    if (((uintptr_t)seeds % 8) != 0 || ((uintptr_t)idx_out % 4) != 0) return SYMODE_E_ALIGN;
    seeded_subsample_kernel<<<dim3((unsigned)n_seeds), dim3(SB), 0, (hipStream_t)stream>>>(n, m, seeds, idx_out);
    return (int)hipGetLastError();
}
