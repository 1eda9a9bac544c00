// valu_rate.hip -- how many cycles does one SIMD of gfx950 need per wave64 VALU instruction, as a function of the
// waves resident on it and of the instruction kind? (MI355X_MICROARCH.md says 2 for v_fma_f32 with several waves,
// the traversal kernel's counters look like 4.)  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 1) {
                asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 2) {
                asm volatile("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %8, %6, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %7, %9 op_sel_hi:[1,0,0]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 3) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if (KIND == 4) {
                asm volatile("v_min3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                             "v_min3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 10) {  // packed fp16 FMA: the reference's shipped slab test (AMD_F16_BVH, mathlib.glsl:144-164)
                asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n"
                             "v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 11) {  // packed fp16 min / max
                asm volatile("v_pk_min_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_min_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n"
                             "v_pk_min_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_min_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 12) {  // fp16 -> fp32 conversions (low half, and high half through SDWA)
                asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_f16 %2, %9\n v_cvt_f32_f16_sdwa %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_f16 %6, %9\n v_cvt_f32_f16_sdwa %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 13) {  // the fp16 slab step's VALU mix for both children: 6 pk_fma + 10 pk_min/max + 4 cvt (20 instructions)
                asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n"
                             "v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n"
                             "v_pk_min_f16 %6, %0, %3\n v_pk_max_f16 %7, %0, %3\n v_pk_min_f16 %0, %1, %4\n v_pk_max_f16 %3, %1, %4\n"
                             "v_pk_min_f16 %1, %2, %5\n v_pk_max_f16 %4, %2, %5\n"
                             "v_pk_max_f16 %6, %6, %0\n v_pk_max_f16 %6, %6, %1\n v_pk_min_f16 %7, %7, %3\n v_pk_min_f16 %7, %7, %4\n"
                             "v_cvt_f32_f16 %2, %6\n v_cvt_f32_f16_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_f16 %0, %7\n v_cvt_f32_f16_sdwa %1, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 14) {  // the fp32 slab step's VALU mix for both children as rt_traverse issues it: 12 fma_mix + 12 min/max + 4 min3/max3 (28)
                asm volatile("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n"
                             "v_min_f32 %6, %0, %3\n v_max_f32 %7, %0, %3\n v_min_f32 %0, %1, %4\n v_max_f32 %3, %1, %4\n v_min_f32 %1, %2, %5\n v_max_f32 %4, %2, %5\n"
                             "v_max3_f32 %6, %6, %0, %1\n v_min3_f32 %7, %7, %3, %4\n"
                             "v_fma_mix_f32 %0, %8, %6, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %7, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n"
                             "v_min_f32 %6, %0, %3\n v_max_f32 %7, %0, %3\n v_min_f32 %0, %1, %4\n v_max_f32 %3, %1, %4\n v_min_f32 %1, %2, %5\n v_max_f32 %4, %2, %5\n"
                             "v_max3_f32 %6, %6, %0, %1\n v_min3_f32 %7, %7, %3, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 7) {   // 8 half-rate VALU + 8 SALU mask operations (independent of the VALU ones) per group
                asm volatile("v_min_f32 %0, %0, %8\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_max_f32 %1, %1, %8\n s_or_b64 s[26:27], s[20:21], s[24:25]\n"
                             "v_min_f32 %2, %2, %8\n s_andn2_b64 s[22:23], s[26:27], s[24:25]\n v_max_f32 %3, %3, %8\n s_xor_b64 s[20:21], s[22:23], s[26:27]\n"
                             "v_min_f32 %4, %4, %8\n s_and_b64 s[26:27], s[20:21], s[24:25]\n v_max_f32 %5, %5, %8\n s_or_b64 s[22:23], s[26:27], s[24:25]\n"
                             "v_min_f32 %6, %6, %8\n s_andn2_b64 s[20:21], s[22:23], s[24:25]\n v_max_f32 %7, %7, %8\n s_xor_b64 s[26:27], s[20:21], s[22:23]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
            } else if (KIND == 8) {   // 8 half-rate VALU + 16 SALU
                asm volatile("v_min_f32 %0, %0, %8\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_bcnt1_i32_b64 s28, s[20:21]\n v_max_f32 %1, %1, %8\n s_or_b64 s[26:27], s[20:21], s[24:25]\n s_cmp_gt_u32 s28, 7\n"
                             "v_min_f32 %2, %2, %8\n s_andn2_b64 s[22:23], s[26:27], s[24:25]\n s_cselect_b64 s[30:31], -1, 0\n v_max_f32 %3, %3, %8\n s_xor_b64 s[20:21], s[22:23], s[26:27]\n s_mov_b64 s[32:33], s[30:31]\n"
                             "v_min_f32 %4, %4, %8\n s_and_b64 s[26:27], s[20:21], s[24:25]\n s_bcnt1_i32_b64 s28, s[26:27]\n v_max_f32 %5, %5, %8\n s_or_b64 s[22:23], s[26:27], s[24:25]\n s_cmp_gt_u32 s28, 9\n"
                             "v_min_f32 %6, %6, %8\n s_andn2_b64 s[20:21], s[22:23], s[24:25]\n s_cselect_b64 s[30:31], -1, 0\n v_max_f32 %7, %7, %8\n s_xor_b64 s[26:27], s[20:21], s[22:23]\n s_mov_b64 s[32:33], s[30:31]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s30", "s31", "s32", "s33", "scc");
            } else if (KIND == 9) {   // 8 half-rate VALU, each consuming a mask a SALU instruction has just produced (v_cndmask on an SGPR pair)
                asm volatile("s_and_b64 s[20:21], s[22:23], s[24:25]\n v_cndmask_b32 %0, %0, %8, s[20:21]\n s_or_b64 s[26:27], s[20:21], s[24:25]\n v_cndmask_b32 %1, %1, %8, s[26:27]\n"
                             "s_andn2_b64 s[22:23], s[26:27], s[24:25]\n v_cndmask_b32 %2, %2, %8, s[22:23]\n s_xor_b64 s[20:21], s[22:23], s[26:27]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                             "s_and_b64 s[26:27], s[20:21], s[24:25]\n v_cndmask_b32 %4, %4, %8, s[26:27]\n s_or_b64 s[22:23], s[26:27], s[24:25]\n v_cndmask_b32 %5, %5, %8, s[22:23]\n"
                             "s_andn2_b64 s[20:21], s[22:23], s[24:25]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n s_xor_b64 s[26:27], s[20:21], s[22:23]\n v_cndmask_b32 %7, %7, %8, s[26:27]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
            } else {
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int vinst_per_group) {
    const int iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4 * 2);
    hipMalloc(&cyc, 256 * 4 * 8 * 8 * 2);
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {       // waves per SIMD: grid = 256 CUs x 4 SIMDs x wps one-wave blocks
        int grid = 256 * 4 * wps;
        k<KIND><<<grid, 64>>>(out, cyc, 10);  // warm
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND><<<grid, 64>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        double per_wave = sum / grid;                                   // memtime ticks (100 MHz? or shader clock) per wave
        double insts = (double)iters * 8 * vinst_per_group;            // VALU instructions per wave
        // wall-clock based: SIMD-cycles per instruction at 2.4 GHz = ms * 2.4e6 / (insts * wps)
        printf("  wps %d: %.2f cyc/inst/SIMD @2.4GHz (%.3f ms, memtime/inst %.2f)", wps, ms * 2.4e6 / (insts * wps), ms, per_wave / insts);
    }
    printf("\n");
}

int main() {
    run<0>("v_fma_f32", 8);
    run<6>("v_mul/v_add_f32", 8);
    run<1>("v_min/v_max_f32", 8);
    run<2>("v_fma_mix_f32", 8);
    run<4>("v_min3/v_max3_f32", 8);
    run<3>("v_cmp(vcc)+v_cndmask", 8);
    // does scalar work cost VALU issue? (per VALU instruction, as above)
    run<7>("v_min/max + 1 SALU each", 8);
    run<8>("v_min/max + 2 SALU each", 8);
    run<9>("s_op -> v_cndmask(sgpr)", 8);
    // the reference's shipped slab arithmetic (AMD_F16_BVH: both children at once in packed fp16) against the fp32 one
    run<10>("v_pk_fma_f16", 8);
    run<11>("v_pk_min/v_pk_max_f16", 8);
    run<12>("v_cvt_f32_f16 (lo / sdwa hi)", 8);
    run<13>("fp16 slab mix (20 inst)", 20);
    run<14>("fp32 slab mix (28 inst)", 28);
    return 0;
}
