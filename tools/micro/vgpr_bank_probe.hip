// Does the VGPR bank of the operands set the issue rate of a three-operand v_fmac_f32 on gfx950?
// acc[i] += r * th[i] written in inline asm with FIXED registers: acc = v[32+i], th = v[64+OFF+i], r = v100 (i = 0..15), so that
// OFF shifts th's bank (register number mod 4) against acc's.  Prints ns per wave-instruction per SIMD at 2 / 4 / 8 resident
// waves for OFF = 0 (same bank), 1, 2, 3, and for r moved to another bank.  Tuning probe, not product code.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/vgpr_bank_probe tools/micro/vgpr_bank_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define FM(a, t, r) "v_fmac_f32 v" #a ", v" #r ", v" #t "\n"
#define BLOCK16(o0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10, o11, o12, o13, o14, o15, r)                                        \
    FM(32, o0, r) FM(33, o1, r) FM(34, o2, r) FM(35, o3, r) FM(36, o4, r) FM(37, o5, r) FM(38, o6, r) FM(39, o7, r) FM(40, o8, r) \
        FM(41, o9, r) FM(42, o10, r) FM(43, o11, r) FM(44, o12, r) FM(45, o13, r) FM(46, o14, r) FM(47, o15, r)

#define CLOB                                                                                                                    \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v64", "v65", \
        "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82",    \
        "v100", "v101"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    // fill the fixed registers with harmless values (small th, r so that nothing overflows)
    asm volatile(
        "v_mov_b32 v100, 0x3a83126f\n v_mov_b32 v101, 0x3a83126f\n"
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n"
        "v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"
        "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 1.0\n v_mov_b32 v67, 1.0\n v_mov_b32 v68, 1.0\n v_mov_b32 v69, 1.0\n"
        "v_mov_b32 v70, 1.0\n v_mov_b32 v71, 1.0\n v_mov_b32 v72, 1.0\n v_mov_b32 v73, 1.0\n v_mov_b32 v74, 1.0\n v_mov_b32 v75, 1.0\n"
        "v_mov_b32 v76, 1.0\n v_mov_b32 v77, 1.0\n v_mov_b32 v78, 1.0\n v_mov_b32 v79, 1.0\n v_mov_b32 v80, 1.0\n v_mov_b32 v81, 1.0\n v_mov_b32 v82, 1.0\n" ::
            : CLOB);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0)        // th bank == acc bank, r in bank 0
            asm volatile(BLOCK16(64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 100)::: CLOB);
        else if constexpr (MODE == 1)   // th bank = acc bank + 1
            asm volatile(BLOCK16(65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 100)::: CLOB);
        else if constexpr (MODE == 2)   // th bank = acc bank + 2
            asm volatile(BLOCK16(66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 100)::: CLOB);
        else if constexpr (MODE == 3)   // th bank = acc bank + 3
            asm volatile(BLOCK16(67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 100)::: CLOB);
        else if constexpr (MODE == 4)   // th bank = acc bank + 1, r in bank 1
            asm volatile(BLOCK16(65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 101)::: CLOB);
        else                            // th == one fixed register for all (broadcast-like: th bank fixed = 0), r bank 0
            asm volatile(BLOCK16(64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 100)::: CLOB);
    }
    float s;
    asm volatile("v_add_f32 %0, v32, v47" : "=v"(s)::CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 20000;
    for (int w : {2, 4, 8}) {
        const int grid = 256 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<MODE><<<grid, 256>>>(out, iters);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<MODE><<<grid, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double winstr = (double)iters * 16 * w;
        printf("%-46s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, w, best, best * 1e6 / winstr,
               best * 1e6 / winstr * 2.4);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    run<0>("acc v32+i, th v64+i (same bank), r v100", out);
    run<1>("acc v32+i, th v65+i (bank +1), r v100", out);
    run<2>("acc v32+i, th v66+i (bank +2), r v100", out);
    run<3>("acc v32+i, th v67+i (bank +3), r v100", out);
    run<4>("acc v32+i, th v65+i (bank +1), r v101", out);
    run<5>("acc v32+i, th v64 for all, r v100", out);
    return 0;
}
