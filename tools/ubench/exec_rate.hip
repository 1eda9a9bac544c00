// exec_rate.hip -- does a wave64 VALU instruction of gfx950 cost less when only part of EXEC is set? The SIMD is 16 lanes wide and takes a
// wave64 in four passes; if a pass whose sixteen lanes are all off were skipped, packing a wave's live rays into its low lanes would make
// the tails of the traversal cheaper (half of every issued vector instruction works on idle lanes).
// Build: make -C tools/ubench exec_rate       Run on the GPU box: tools/ubench/exec_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long mask, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // (the mask through an ordinary divergent branch, so that the compiler knows what EXEC is: written into EXEC behind its back, the
    // address arithmetic it scheduled into the region ran on the masked lanes only and the store after it faulted)
    if ((mask >> threadIdx.x) & 1ull)
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0)
                asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            else if (KIND == 1)
                asm volatile("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %8, %6, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %7, %9 op_sel_hi:[1,0,0]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            else
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && cyc) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name) {
    const int iters = 2000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    hipMalloc(&cyc, 256 * 4 * 8 * 8);
    struct { const char* what; unsigned long long m; } masks[] = {
        {"all 64", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"low 24", 0xFFFFFFull}, {"low 20", 0xFFFFFull}, {"low 17", 0x1FFFFull}, {"low 16", 0xFFFFull}, {"low 12", 0xFFFull},
        {"low 9", 0x1FFull}, {"low 8", 0xFFull}, {"low 4", 0xFull}, {"lane 0", 1ull}, {"every 2nd", 0x5555555555555555ull}, {"every 4th", 0x1111111111111111ull},
        {"every 8th", 0x0101010101010101ull}, {"every 16th", 0x0001000100010001ull}, {"8 in each half", 0x000000FF000000FFull}};
    for (int wps : {1, 8}) {
        printf("%-14s %d wave(s) per SIMD; per mask: wall-clock SIMD cycles per instruction at 2.4 GHz / s_memtime ticks per instruction of a wave:\n   ", name, wps);
        for (auto& mk : masks) {
            int grid = 256 * 4 * wps;
            k<KIND><<<grid, 64>>>(out, mk.m, 10, nullptr);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<KIND><<<grid, 64>>>(out, mk.m, iters, cyc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (auto v : h) sum += (double)v;
            printf("  %s %.2f/%.2f", mk.what, ms * 2.4e6 / ((double)iters * 64 * wps), sum / grid / ((double)iters * 64));
        }
        printf("\n");
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_min/v_max");
    run<1>("v_fma_mix_f32");
    run<2>("v_fma_f32");
    return 0;
}
