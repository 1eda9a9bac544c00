// exact_math.hip -- are there shorter instruction sequences than the compiler's IEEE expansions of sqrtf(x) and 1.0f / x that
// give the SAME bits? (study for rt_shade: 13 normalize3 per shaded ray = 13 x (sqrt ~14 + divide ~10 instructions).)
// Every float bit pattern in [lo, hi] is pushed through the candidates and compared with the compiler's own result.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/ubench/exact_math.hip -o exact_math && ./exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// sqrt candidates
__device__ __forceinline__ float sqrt_s1(float x) {   // rsq, one Newton step on the root
    float g = __builtin_amdgcn_rsqf(x), s = x * g, h = 0.5f * g;
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s2(float x) {   // ... and a second one
    float g = __builtin_amdgcn_rsqf(x), s = x * g, h = 0.5f * g;
    float r = fmaf(-s, s, x);
    s = fmaf(r, h, s);
    r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s3(float x) {   // hardware sqrt + one step with 0.5 * rcp(s)
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
// reciprocal candidates
__device__ __forceinline__ float rcp_r1(float s) {
    float r = __builtin_amdgcn_rcpf(s);
    float e = fmaf(-s, r, 1.0f);
    return fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_r2(float s) {
    float r = __builtin_amdgcn_rcpf(s);
    float e = fmaf(-s, r, 1.0f);
    r = fmaf(e, r, r);
    e = fmaf(-s, r, 1.0f);
    return fmaf(e, r, r);
}
// the pair as normalize3 uses it: 1 / sqrt(d)
__device__ __forceinline__ float inv_len_fast(float d) { return rcp_r2(sqrt_s2(d)); }

__global__ void sweep(uint32_t lo, uint32_t n, unsigned long long* bad, uint32_t* first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = u2f(lo + i);
    const float sq = sqrtf(x), rc = 1.0f / x, il = 1.0f / sqrtf(x);
    const float c[6] = {sqrt_s1(x), sqrt_s2(x), sqrt_s3(x), rcp_r1(x), rcp_r2(x), inv_len_fast(x)};
    const float w[6] = {sq, sq, sq, rc, rc, il};
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (f2u(c[k]) != f2u(w[k])) {
            if (atomicAdd(&bad[k], 1ull) == 0ull) first[k] = lo + i;
        }
}

int main() {
    unsigned long long* d_bad; uint32_t* d_first;
    hipMalloc(&d_bad, 6 * sizeof(unsigned long long)); hipMalloc(&d_first, 6 * 4);
    const char* names[6] = {"sqrt: rsq + 1 step", "sqrt: rsq + 2 steps", "sqrt: v_sqrt + 1 step (rcp)", "rcp: v_rcp + 1 step", "rcp: v_rcp + 2 steps", "1/sqrt: s2 then r2"};
    // ranges of positive floats by exponent: [2^-126, 2^-64), [2^-64, 2^64), [2^64, 2^127]
    const uint32_t edges[4] = {0x00800000u, 0x1F800000u, 0x5F800000u, 0x7F000000u};
    for (int r = 0; r < 3; r++) {
        hipMemset(d_bad, 0, 6 * sizeof(unsigned long long)); hipMemset(d_first, 0, 24);
        const uint32_t lo = edges[r], hi = edges[r + 1];
        for (uint64_t base = lo; base < hi; base += (1u << 28)) {
            const uint32_t n = (uint32_t)((hi - base) < (1u << 28) ? (hi - base) : (1u << 28));
            sweep<<<(n + 255) / 256, 256>>>((uint32_t)base, n, d_bad, d_first);
        }
        hipDeviceSynchronize();
        unsigned long long bad[6]; uint32_t first[6];
        hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost); hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost);
        printf("floats 0x%08x .. 0x%08x (%u patterns)\n", lo, hi, hi - lo);
        for (int k = 0; k < 6; k++) printf("  %-30s mismatches %12llu  first at 0x%08x\n", names[k], bad[k], first[k]);
    }
    return 0;
}
