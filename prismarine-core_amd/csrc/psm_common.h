// psm_common.h -- wave64 / workgroup primitives shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psm_math.h"

namespace psm {

constexpr int WAVE = 64;

PSM_D int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
PSM_D uint64_t lanemask_lt() {
    int l = lane_id();
    return l == 0 ? 0ull : (~0ull >> (64 - l));
}

// inclusive scan across the 64 lanes of a wave
PSM_D uint32_t wave_scan_incl(uint32_t v) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane_id() >= d) v += t;
    }
    return v;
}
PSM_D uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// exclusive scan over a workgroup of NT threads (NT multiple of 64, <= 1024).
// `tmp` is an LDS array of at least NT/64+1 uint32. Returns the exclusive prefix; *total = block sum.
template <int NT>
PSM_D uint32_t block_scan_excl(uint32_t v, uint32_t* tmp, uint32_t* total) {
    constexpr int NW = NT / WAVE;
    int w = threadIdx.x / WAVE, l = lane_id();
    uint32_t inc = wave_scan_incl(v);
    if (l == WAVE - 1) tmp[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t s = (l < NW) ? tmp[l] : 0u;
        uint32_t si = wave_scan_incl(s);
        if (l < NW) tmp[l] = si - s;
        if (l == NW - 1) tmp[NW] = si;
    }
    __syncthreads();
    uint32_t base = tmp[w];
    if (total) *total = tmp[NW];
    __syncthreads();
    return base + inc - v;
}

}  // namespace psm
