// gather_rate.hip -- how many dependent 32-byte record fetches per second does gfx950 sustain when every lane of a
// wave64 chases its own pointer, the way the BVH traversal loop does (two global_load_dwordx4 per record, the next
// index depends on the loaded data)?  Varies: table size (L2 / Infinity Cache / HBM resident), live lanes per wave,
// index distribution (uniform, or "tree": a level drawn uniformly, an index uniform within the level, so the upper
// levels are hot like a BVH's), and VALU padding per step (`pad` x 8 v_min/v_max in four independent chains, on top of
// the ~20 VALU instructions of the chase itself).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int LOADS, bool TREE, bool PAIR = false>
__global__ __launch_bounds__(128, 8) void chase(const uint4* __restrict__ tab, int log2n, int steps, int live, int pad,
                                                uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    if (lane >= live) return;
    uint32_t gid = blockIdx.x * 128 + threadIdx.x;
    uint32_t r = hash32(gid * 2654435761u + 12345u);
    const uint32_t mask = (1u << log2n) - 1u;
    uint32_t idx = r & mask;
    float acc = (float)lane, acc1 = acc + 1.f, acc2 = acc + 2.f, acc3 = acc + 3.f;
    for (int s = 0; s < steps; s++) {
        uint4 a = tab[(size_t)idx * 2];
        uint4 b = LOADS == 2 ? tab[(size_t)idx * 2 + 1] : a;
        r = hash32(r ^ a.x ^ a.y ^ a.w ^ b.x ^ b.y ^ b.z);   // every component is used: the loads stay two dwordx4
        // half-rate VALU work (v_min / v_max, like the slab test): `pad` x 8 instructions in four independent chains, as
        // inline assembly so that the count is what it says (fminf / fmaxf would add a canonicalising v_max each)
        const float x0 = __uint_as_float(a.z & 0x3fffffffu), x1 = __uint_as_float(b.w & 0x3fffffffu);
        for (int p = 0; p < pad; p++)
            asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_min_f32 %2, %2, %5\n v_max_f32 %3, %3, %4\n"
                         "v_min_f32 %0, %0, %5\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %5\n"
                         : "+v"(acc), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(x0), "v"(x1));
        if (PAIR && (s & 1) == 0) {
            idx ^= 1u + (r & 2u);   // every other fetch stays in the 128-byte line of the one before (a record next to it)
        } else if (TREE) {
            uint32_t lvl = ((r >> 24) * (uint32_t)(log2n + 1)) >> 8;   // level 0..log2n, each (almost) equally likely; no division
            uint32_t base = (lvl == 0) ? 0u : ((1u << lvl) - 1u);
            idx = (base + ((r >> 2) & ((1u << lvl) - 1u))) & mask;
        } else {
            idx = r & mask;
        }
    }
    out[gid] = r ^ __float_as_uint(acc) ^ __float_as_uint(acc1) ^ __float_as_uint(acc2) ^ __float_as_uint(acc3);
}

template <int LOADS, bool TREE, bool PAIR = false>
static double run(const uint4* tab, int log2n, int live, int pad, uint32_t* out) {
    const int grid = 256 * 16, steps = 256;   // 16 workgroups of 128 per CU = 8 waves per SIMD, one generation
    chase<LOADS, TREE, PAIR><<<grid, 128>>>(tab, log2n, 8, live, pad, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chase<LOADS, TREE, PAIR><<<grid, 128>>>(tab, log2n, steps, live, pad, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    double recs = (double)grid * 2 * live * steps;
    return recs / (ms * 1e-3) * 1e-9;   // G records per second
}

int main(int argc, char** argv) {
    const int maxlog = 24;   // 16 Mi records x 32 B = 512 MiB
    std::vector<uint32_t> h((size_t)8 << maxlog);
    uint32_t s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    uint4* tab; uint32_t* out;
    hipMalloc(&tab, h.size() * 4);
    hipMalloc(&out, 256 * 16 * 128 * 4);
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("G records/s (32-byte record = 2 x dwordx4 unless noted); 8 waves per SIMD, dependent chain of 256 fetches per lane\n");
    if (argc > 1 && !strcmp(argv[1], "slab")) {
        // round 3: what the packed-fp16 slab test (the reference's AMD_F16_BVH build) would buy the traversal loop. The fp32
        // step is +96 half-rate VALU per fetch here (12 x 8); the fp16 step replaces 28 of its instructions by 20 (valu_rate:
        // "fp16 slab mix" against "fp32 slab mix"), i.e. +88 at equal rates and +80 / +72 if the packed instructions ran
        // at twice / four times the rate of the ones they replace
        for (int log2n : {18, 24}) {
            for (int pad : {12, 11, 10, 9}) {
                printf("table %4d MiB tree    +%3d VALU:", (32 << log2n) >> 20, 8 * pad);
                for (int live : {16, 24, 32, 48, 64}) printf("  live %2d: %6.1f", live, run<2, true>(tab, log2n, live, pad, out));
                printf("\n");
            }
        }
        return 0;
    }
    for (int log2n : {15, 18, 20, 24}) {   // 1 MiB, 8 MiB, 32 MiB, 512 MiB tables
        for (int tree = 0; tree < 2; tree++) {
            for (int pad : {0, 6, 12, 18}) {
                printf("table %4d MiB %-7s +%3d VALU:", (32 << log2n) >> 20, tree ? "tree" : "uniform", 8 * pad);
                for (int live : {8, 16, 32, 48, 64}) {
                    double g = tree ? run<2, true>(tab, log2n, live, pad, out) : run<2, false>(tab, log2n, live, pad, out);
                    printf("  live %2d: %6.1f", live, g);
                }
                if (pad == 0) {
                    double g1 = tree ? run<1, true>(tab, log2n, 64, 0, out) : run<1, false>(tab, log2n, 64, 0, out);
                    printf("  | one dwordx4 per record, live 64: %6.1f", g1);
                }
                printf("\n");
            }
        }
        // what a layout buys in which every other step finds its record in the line the step before has fetched
        for (int pad : {0, 12}) {
            printf("table %4d MiB tree, every other fetch in the previous line, +%3d VALU:", (32 << log2n) >> 20, 8 * pad);
            for (int live : {16, 32, 48, 64}) printf("  live %2d: %6.1f", live, run<2, true, true>(tab, log2n, live, pad, out));
            printf("   (independent lines:");
            for (int live : {32, 48}) printf(" live %2d: %6.1f", live, run<2, true, false>(tab, log2n, live, pad, out));
            printf(")\n");
        }
    }
    return 0;
}
