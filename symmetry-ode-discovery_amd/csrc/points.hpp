// Coalesced access to (N, D) row-major fp32 trajectory arrays.
//
// The flattened trajectory tensor (n_ics*n_steps, D) (reference dataset.py:193-194) is read
// as a stream of 16-byte vectors: one "chunk" per lane per step holds PPT whole points
// (D=1: 4 points in one dwordx4, D=2: 2 points, D=3: 4 points in three dwordx4, D=4: 1).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace symode {

// The points of a chunk, one after the other: body(i) with i a COMPILE-TIME constant in each of the PPT copies of the body.
// Template recursion, not `#pragma unroll`: the pragma is a request, and around bodies with inlined sinf / expf the
// optimiser declines it beyond its size threshold ("loop not unrolled", -Wpass-failed) -- the loop variable then indexes
// the per-chunk point arrays at run time and they land in scratch (100-600 bytes per lane in the D = 3 sine libraries).
// (Tried and dropped for those libraries: ONE copy of the body in a real loop with the point arrays rotated by a row per
//  pass -- no scratch, half the registers, but hipcc 7.2 emitted the loop-carried row copies as KILL pseudo-ops and every
//  point of a chunk came out wrong; the same loop with row `pass` swapped in and out behind uniform branches was right
//  and back in scratch.  Those libraries take the per-point path instead: kernels.hpp, chunked_stream.)
template <int I, int N, typename F>
__device__ __forceinline__ void each_point_static(F&& body) {
    if constexpr (I < N) {
        body(std::integral_constant<int, I>{});
        each_point_static<I + 1, N>(body);
    }
}

template <int PPT, typename F>
__device__ __forceinline__ void each_point(F&& body) {
    each_point_static<0, PPT>(body);
}

template <int D>
struct Chunk {
    static constexpr int PPT = (D == 1) ? 4 : (D == 2) ? 2 : (D == 3) ? 4 : 1;   // points per chunk
    static constexpr int NV = PPT * D / 4;                                         // dwordx4 per chunk
    static_assert(PPT * D == NV * 4, "chunk must be a whole number of 16-byte vectors");
};

// D = 3: a point is 12 bytes, so chunk c of lane l (vectors 3c .. 3c+2) makes every 16-byte lane load 48-byte strided --
// three requests each using a third of the lines they touch (measured 4.3 TB/s where d = 2 and d = 4 stream at 6.4-6.8).
// A full wave instead fetches its 192-vector tile (256 points) coalesced -- lane l takes vectors l, 64+l, 128+l of the
// tile -- and redistributes through a wave-private LDS slab: lane l reads back vectors 3l, 3l+1, 3l+2, its own chunk.
// Preconditions (checked by the caller): all 64 lanes active, lane l holds chunk c0 + l.
// Written for any chunk of NVEC 16-byte vectors per lane (the regulariser's J_g at D = 3: 9 vectors = 144 bytes per lane,
// which as per-lane loads touch ~9x the lines they use and thrash the 32 KB L1 with 8-12 waves per CU): the wave fetches
// its 64 NVEC-vector tile coalesced -- lane l takes vectors l, 64 + l, ... -- and reads its own chunk back from a
// wave-private slab of 64 NVEC vectors.  ds_read_b128 at a lane stride of NVEC vectors is conflict-free for odd NVEC.
template <int NVEC, bool NT>
__device__ __forceinline__ void load_tile_raw(const float* __restrict__ a, long c0, int lane, float4 (&t)[NVEC]) {
    const float4* q = reinterpret_cast<const float4*>(a) + c0 * NVEC + lane;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        if constexpr (NT) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v u = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q + 64 * i));
            t[i] = make_float4(u.x, u.y, u.z, u.w);
        } else {
            t[i] = q[64 * i];
        }
    }
}

template <int NVEC>
__device__ __forceinline__ void exchange_tile(const float4 (&t)[NVEC], float4 (&v)[NVEC], float4* slab, int lane) {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) slab[64 * i + lane] = t[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < NVEC; ++i) v[i] = slab[NVEC * lane + i];
    __builtin_amdgcn_wave_barrier();                     // the next exchange reuses the slab (a wave's LDS operations run in order)
}

// Reverse direction for stores: v (this lane's chunk) -> t (tile order).  Plain vector VALUES in and out: with float4
// arrays by reference the result array stayed a stack object and went through scratch around the wave barrier
// (64 bytes per lane in every D = 3 kernel that stores points).
typedef float f4x __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void exchange_tile3_out(f4x v0, f4x v1, f4x v2, f4x& t0, f4x& t1, f4x& t2, float4* slab, int lane) {
    f4x* s = reinterpret_cast<f4x*>(slab);
    s[3 * lane + 0] = v0;
    s[3 * lane + 1] = v1;
    s[3 * lane + 2] = v2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    t0 = s[lane];
    t1 = s[64 + lane];
    t2 = s[128 + lane];
    __builtin_amdgcn_wave_barrier();
}

// Wave-private LDS slab for the D = 3 tile exchange of a 256-thread workgroup (one per call site that needs it).
#define SYMODE_TILE3_SLAB(name) __shared__ float4 name[4][192]

template <int D>
__device__ __forceinline__ void load_chunk(const float* __restrict__ a, long c, float (&p)[Chunk<D>::PPT][D]) {
    constexpr int NV = Chunk<D>::NV;
    typedef float f4v __attribute__((ext_vector_type(4)));
    float f[NV * 4];
    if constexpr (D == 3) {
        // callers hand consecutive chunks to consecutive lanes (chunk c of lane l = base + l): a whole wave takes the coalesced tile
        if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {
            SYMODE_TILE3_SLAB(slab);
            const int lane = threadIdx.x & 63;
            float4 t[3], v[3];
            load_tile_raw<3, true>(a, c - lane, lane, t);
            exchange_tile<3>(t, v, slab[threadIdx.x >> 6], lane);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f[4 * i + 0] = v[i].x;
                f[4 * i + 1] = v[i].y;
                f[4 * i + 2] = v[i].z;
                f[4 * i + 3] = v[i].w;
            }
#pragma unroll
            for (int i = 0; i < NV * 4; ++i) p[i / D][i % D] = f[i];
            return;
        }
    }
    const f4v* q = reinterpret_cast<const f4v*>(a) + c * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f4v v = __builtin_nontemporal_load(q + i);          // streamed once
        f[4 * i + 0] = v.x;
        f[4 * i + 1] = v.y;
        f[4 * i + 2] = v.z;
        f[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NV * 4; ++i) p[i / D][i % D] = f[i];
}

// Raw 16-byte vectors of chunk c (optionally non-temporal: the stream is read exactly once).
template <int D, bool NT>
__device__ __forceinline__ void load_chunk_raw(const float* __restrict__ a, long c, float4 (&v)[Chunk<D>::NV]) {
    constexpr int NV = Chunk<D>::NV;
    const float4* q = reinterpret_cast<const float4*>(a) + c * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (NT) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q + i));
            v[i] = make_float4(t.x, t.y, t.z, t.w);
        } else {
            v[i] = q[i];
        }
    }
}

template <int D>
__device__ __forceinline__ void unpack_chunk(const float4 (&v)[Chunk<D>::NV], float (&p)[Chunk<D>::PPT][D]) {
    constexpr int NV = Chunk<D>::NV;
    float f[NV * 4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f[4 * i + 0] = v[i].x;
        f[4 * i + 1] = v[i].y;
        f[4 * i + 2] = v[i].z;
        f[4 * i + 3] = v[i].w;
    }
#pragma unroll
    for (int i = 0; i < NV * 4; ++i) p[i / D][i % D] = f[i];
}

template <int D>
__device__ __forceinline__ void store_chunk(float* __restrict__ a, long c, const float (&p)[Chunk<D>::PPT][D]) {
    constexpr int NV = Chunk<D>::NV;
    float4* q = reinterpret_cast<float4*>(a) + c * NV;
    float f[NV * 4];
#pragma unroll
    for (int i = 0; i < NV * 4; ++i) f[i] = p[i / D][i % D];
    if constexpr (D == 3) {
        if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {           // whole wave: coalesced tile stores (see load_chunk)
            SYMODE_TILE3_SLAB(slab);
            const int lane = threadIdx.x & 63;
            f4x t0, t1, t2;
            exchange_tile3_out(f4x{f[0], f[1], f[2], f[3]}, f4x{f[4], f[5], f[6], f[7]}, f4x{f[8], f[9], f[10], f[11]}, t0, t1, t2,
                               slab[threadIdx.x >> 6], lane);
            f4x* tile = reinterpret_cast<f4x*>(a) + (c - lane) * 3 + lane;
            tile[0] = t0;
            tile[64] = t1;
            tile[128] = t2;
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) q[i] = make_float4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
}

// Same, as non-temporal stores: outputs of the streaming maps are written once and not read back by the kernel.
template <int D>
__device__ __forceinline__ void store_chunk_nt(float* __restrict__ a, long c, const float (&p)[Chunk<D>::PPT][D]) {
    if constexpr (D == 3) {
        store_chunk<D>(a, c, p);                                   // coalesced tile stores through the wave's LDS slab
    } else {
        constexpr int NV = Chunk<D>::NV;
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v* q = reinterpret_cast<f4v*>(a) + c * NV;
        float f[NV * 4];
#pragma unroll
        for (int i = 0; i < NV * 4; ++i) f[i] = p[i / D][i % D];
#pragma unroll
        for (int i = 0; i < NV; ++i) __builtin_nontemporal_store(f4v{f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]}, q + i);
    }
}

template <int D>
__device__ __forceinline__ void load_point(const float* __restrict__ a, long n, float (&p)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) p[i] = a[n * D + i];
}

template <int D>
__device__ __forceinline__ void store_point(float* __restrict__ a, long n, const float (&p)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) a[n * D + i] = p[i];
}

// Two-chunk software pipeline over one (N, D) problem: per step the operands of two chunks are requested (load(c, ops):
// 16-byte non-temporal vectors through load_chunk) before either chunk is computed (compute(c, ops)), so a lane keeps
// two chunks of every operand in flight; ragged tails and unaligned bases go point by point.  Ops is the caller's bundle
// of per-chunk operand registers.
//   ROUNDS = 1: chunks c and c + nthreads with a grid-wide stride (a map launches the whole index space as workgroups,
//               so this is one chunk per lane);
//   ROUNDS > 1: every workgroup owns contiguous slabs of ROUNDS * BLOCK chunks (chunks c and c + BLOCK per step): the
//               launch still walks memory in address order, with ROUNDS chunks per lane -- for the order 4-5 libraries,
//               whose 30-42 coefficient loads and mask products per workgroup cost as much issue time as two points of
//               work (kernels.hpp, map_rounds).
template <int D, int BLOCK, class Ops, int ROUNDS = 1, typename Load, typename Compute, typename PointBody>
__device__ __forceinline__ void for_each_chunk2(long N, bool vec, Load load, Compute compute, PointBody point_body) {
    constexpr int PPT = Chunk<D>::PPT;
    const long tid = (long)blockIdx.x * BLOCK + threadIdx.x;
    const long nthreads = (long)gridDim.x * BLOCK;
    if (vec) {
        const long nchunks = N / PPT;
        if constexpr (ROUNDS == 1) {
            long c = tid;
            for (; c + nthreads < nchunks; c += 2 * nthreads) {
                Ops a, b;
                load(c, a);
                load(c + nthreads, b);
                compute(c, a);
                compute(c + nthreads, b);
            }
            if (c < nchunks) {
                Ops a;
                load(c, a);
                compute(c, a);
            }
        } else {
            constexpr long PER = (long)ROUNDS * BLOCK;
            for (long lo = (long)blockIdx.x * PER; lo < nchunks; lo += (long)gridDim.x * PER) {
                const long hi = lo + PER < nchunks ? lo + PER : nchunks;
                long c = lo + threadIdx.x;
#pragma unroll 1                                          // (unrolled, forward_jvp keeps four chunks of operands live and loses occupancy)
                for (; c + BLOCK < hi; c += 2 * BLOCK) {
                    Ops a, b;
                    load(c, a);
                    load(c + BLOCK, b);
                    compute(c, a);
                    compute(c + BLOCK, b);
                }
                if (c < hi) {
                    Ops a;
                    load(c, a);
                    compute(c, a);
                }
            }
        }
        const long n = nchunks * PPT + tid;
        if (n < N) point_body(n);
    } else {
        for (long n = tid; n < N; n += nthreads) point_body(n);
    }
}

// The ring itself, for callers whose chunk is more than NA equal arrays (the regulariser's J_g): `slot` = NVTOT 16-byte
// registers per chunk, load(q, slot) issues chunk q's loads into them (q already clamped to the last chunk),
// use(c, slot) consumes chunk c.  Chunks c0, c0 + stride, ... < nchunks, in that order.
template <int R, int NVTOT, typename Load, typename Use>
__device__ __forceinline__ void chunk_ring(long nchunks, long c0, long stride, Load load, Use use) {
    if (nchunks <= 0) return;
    const long lastc = nchunks - 1;
    float4 ring[R][NVTOT];
    long c = c0;
    // (slot loops by template recursion: `#pragma unroll` is declined around large bodies, and a ring indexed at run time
    //  is a ring in scratch)
    each_point_static<0, R>([&](auto k) {
        const long cc = c + k * stride;
        load(cc < lastc ? cc : lastc, ring[k]);
    });
    for (; c + (R - 1) * stride < nchunks; c += R * stride) {
        each_point_static<0, R>([&](auto k) {
            use(c + k * stride, ring[k]);
            const long cc = c + (R + k) * stride;
            load(cc < lastc ? cc : lastc, ring[k]);
            __builtin_amdgcn_sched_barrier(0);           // keep the refill here: the scheduler would sink all R to the loop end
        });
    }
    each_point_static<0, R>([&](auto k) {
        if (c + k * stride < nchunks) use(c + k * stride, ring[k]);
    });
}

// Register-ring pipeline over NA operand arrays of one (N, D) problem: every lane keeps R chunks of every operand in
// flight -- the slot a chunk has just been consumed from is refilled at once with the chunk R rounds ahead (16-byte
// non-temporal loads; indices past the end are clamped to the last chunk, so the tail over-reads in bounds and the
// vmcnt arithmetic stays uniform).  compute(c, p) receives the unpacked points p[a][i][:] of chunk c of operand a.
// Why: a reduction launch is a few long-lived workgroups (its partial rows must stay few), so unlike the index-space
// maps nothing but the wave's own run-ahead hides the HBM latency; with the two-chunk form a wave of the 8-byte-per-point
// regulariser issued one 16-byte load per ~230 vector instructions and waited out every miss (0.55 of the issue slots).
// D = 3 keeps the two-chunk form: its 12-byte points come through the wave's coalesced-tile exchange (load_chunk).
template <int D, int BLOCK, int R, int NA, typename Compute, typename PointBody>
__device__ __forceinline__ void for_each_chunk_ring(long N, bool vec, const float* const (&arr)[NA], Compute compute,
                                                    PointBody point_body) {
    constexpr int PPT = Chunk<D>::PPT, NV = Chunk<D>::NV;
    const long tid = (long)blockIdx.x * BLOCK + threadIdx.x;
    const long nthreads = (long)gridDim.x * BLOCK;
    if (!vec) {
        for (long n = tid; n < N; n += nthreads) point_body(n);
        return;
    }
    const long nchunks = N / PPT;
    if constexpr (D == 3) {
        // whole waves: the tiles of ALL operands of two chunks are requested (coalesced) before the first of them goes
        // through the wave's LDS slab; a form that fetched and exchanged operand by operand waited out every miss in
        // turn (loss_grad, d = 3 order 2: 135 -> 231 us at 2^25 points)
        __shared__ float4 slab3[BLOCK / 64][3 * 64];
        const int lane = threadIdx.x & 63;
        float4* slab = slab3[threadIdx.x >> 6];
        long c = tid;
        for (; c + nthreads < nchunks; c += 2 * nthreads) {
            float a[NA][PPT][D], b[NA][PPT][D];
            if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {
                float4 ta[NA][NV], tb[NA][NV], va[NV];
#pragma unroll
                for (int q = 0; q < NA; ++q) load_tile_raw<NV, true>(arr[q], c - lane, lane, ta[q]);
#pragma unroll
                for (int q = 0; q < NA; ++q) load_tile_raw<NV, true>(arr[q], c + nthreads - lane, lane, tb[q]);
#pragma unroll
                for (int q = 0; q < NA; ++q) {
                    exchange_tile<NV>(ta[q], va, slab, lane);
                    unpack_chunk<D>(va, a[q]);
                }
#pragma unroll
                for (int q = 0; q < NA; ++q) {
                    exchange_tile<NV>(tb[q], va, slab, lane);
                    unpack_chunk<D>(va, b[q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NA; ++q) load_chunk<D>(arr[q], c, a[q]);
#pragma unroll
                for (int q = 0; q < NA; ++q) load_chunk<D>(arr[q], c + nthreads, b[q]);
            }
            compute(c, a);
            compute(c + nthreads, b);
        }
        if (c < nchunks) {
            float a[NA][PPT][D];
#pragma unroll
            for (int q = 0; q < NA; ++q) load_chunk<D>(arr[q], c, a[q]);
            compute(c, a);
        }
    } else {
        chunk_ring<R, NA * NV>(
            nchunks, tid, nthreads,
            [&](long q, float4 (&slot)[NA * NV]) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    float4 v[NV];
                    load_chunk_raw<D, true>(arr[a], q, v);
#pragma unroll
                    for (int i = 0; i < NV; ++i) slot[a * NV + i] = v[i];
                }
            },
            [&](long cc, const float4 (&slot)[NA * NV]) {
                float pts[NA][PPT][D];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    float4 v[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) v[i] = slot[a * NV + i];
                    unpack_chunk<D>(v, pts[a]);
                }
                compute(cc, pts);
            });
    }
    const long n = nchunks * PPT + tid;
    if (n < N) point_body(n);
}

}  // namespace symode
