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

// inclusive scan across the 64 lanes of a wave: DPP moves inside the vector unit -- row_shr:1,2,4,8 scan each row of 16 lanes,
// row_bcast:15 carries a row's total into rows 1 and 3, row_bcast:31 the lower half's into rows 2 and 3 -- six v_add_u32 with a DPP
// operand. (Round 5: the __shfl_up this used to be is ds_bpermute_b32, a trip through the LDS crossbar per step: six of them were a
// quarter of radix_local's prefix phase, tools/sort_log.py.)
PSM_D uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source read 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
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

// exclusive scan of in[0 .. n) into out[0 .. n) by ONE workgroup of NT threads (in == out allowed); returns the total to
// every thread. Strips of 4 NT elements: a thread reads and writes four consecutive ones (16-byte accesses, coalesced over
// the wave), a running carry between the strips. `tmp`: 33 words of LDS.
template <int NT>
PSM_D uint32_t block_scan_array(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp) {
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 4u * NT) {
        const uint32_t i = base + 4u * tid;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i + 3u < n) v = *(const uint4*)(in + i);
        else {
            if (i < n) v.x = in[i];
            if (i + 1u < n) v.y = in[i + 1u];
            if (i + 2u < n) v.z = in[i + 2u];
        }
        uint32_t strip;
        const uint32_t ex = carry + block_scan_excl<NT>(v.x + v.y + v.z + v.w, tmp, &strip);
        const uint4 o = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        if (i + 3u < n) *(uint4*)(out + i) = o;
        else {
            if (i < n) out[i] = o.x;
            if (i + 1u < n) out[i + 1u] = o.y;
            if (i + 2u < n) out[i + 2u] = o.z;
        }
        carry += strip;
    }
    return carry;
}
PSM_D uint32_t block_scan_array_1024(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp) { return block_scan_array<1024>(in, out, n, tmp); }

// ---- segmented ray queue -------------------------------------------------------------------------------------------
// A ray queue is a sequence of segments of QUEUE_SEG slots: segment b holds rays bases[b] .. bases[b+1]-1 of the queue
// (dense numbering = the canonical queue order) in its first slots; bases[nb] = total. The shading kernel writes each
// workgroup's output rays into the workgroup's own segment (ordered inside it) and a scan of the counts gives `bases`:
// the next round reads the rays THROUGH this map, so the compacting copy of reloadQueuedRays (Pipeline.inl:325-359:
// three buffer copies per round there, one gather + scatter of every ray here in round 1) is gone. A queue written
// densely (camera, upload) is one segment: nb = 1.
#ifndef PSM_SHADE_BLOCK
#define PSM_SHADE_BLOCK 256   // (128: equal to 0.7 % faster with frames in flight, 4 % slower alone (0.332 -> 0.347 ms of shading per C3 frame); 64: equal in flight, slower alone)
#endif
constexpr int SHADE_BLOCK = PSM_SHADE_BLOCK;           // rays a shading workgroup takes from the queue
constexpr uint32_t QUEUE_SEG = 4u * SHADE_BLOCK;       // ... and the slots of its output segment: at most 4 rays out per ray in

// Accesses to data that is written once per launch and read by a LATER launch (ray queues, hit records, hand-over state,
// images): plain loads and plain stores. Round 3 put the non-temporal hint on them (hipcc: `global_load_dwordx4 ... nt`,
// `global_store_dwordx4 ... nt`; nt = 1, sc0 = sc1 = 0), which keeps ~1.3 GB per C3 frame from displacing node and triangle
// records in L2 -- worth 2.5 % with the hint on loads and stores, 1.2 % on stores only. It was taken off the LOADS after an `nt`
// load of the hand-over state returned, for some slots, what the slot had held two launches earlier (the two state buffers
// alternate, so a resume wave reads addresses it read -- or another wave of its XCD wrote -- then), and in
// round 4 off the STORES as well: the 8 XCDs' L2s are not coherent with each other and a launch boundary is what makes one
// XCD's stores visible to another's loads; what that boundary does to lines allocated under the `nt` policy could not be
// established from the documents at hand (the failure above says: not always what it does to the others), an `nt` store
// leaves its line in the storing XCD's L2 exactly as a plain store does (MI355X_MICROARCH.md, "stores of each flavour"), and
// "it never showed with stores" is an observation, not an argument. The two names stay: they mark the once-per-launch streams.
// (PSM_EXP_NT_STORES=1: the round-3 stores, an experiment build -- never the product; round 3's A/B of the hint is profiles/r03_cache_policy_ab.txt.)
#ifndef PSM_EXP_NT_STORES
#define PSM_EXP_NT_STORES 0
#endif
typedef float psm_f4v __attribute__((ext_vector_type(4)));
PSM_D float4 ld_stream(const float4* p) { return *p; }
PSM_D uint32_t ld_stream(const uint32_t* p) { return *p; }
PSM_D void st_stream(float4* p, float4 v) {
    if (PSM_EXP_NT_STORES) {
        psm_f4v w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, (psm_f4v*)p);
    } else {
        *p = v;
    }
}
PSM_D void st_stream(uint32_t* p, uint32_t v) {
    if (PSM_EXP_NT_STORES) __builtin_nontemporal_store(v, p);
    else *p = v;
}

struct RayQueue {
    const float4 *A, *B, *C;   // origin|texel, direct|bitfield, color|pkey
    const uint32_t* bases;     // nb + 1 entries
    uint32_t nb;
};

// slot of ray i (i < total): interpolation guess, gallop, binary search for the b with bases[b] <= i < bases[b+1]
PSM_D uint32_t queue_loc(const uint32_t* __restrict__ bases, uint32_t nb, uint32_t total, uint32_t i) {
    if (nb <= 1u) return i;
    uint32_t lo = 0, hi = nb;  // bases[lo] <= i < bases[hi]
    // a first guess only (the search below is exact from any start): i * nb / total in floats -- the 64-bit integer quotient
    // this used to be is ~100 scalar and vector instructions per wave in every prologue (hipcc expands it in software)
    uint32_t g = (uint32_t)((float)i * ((float)nb / (float)total));
    g = g < nb ? g : nb - 1u;
    if (bases[g] <= i) {
        lo = g;
        uint32_t step = 1;
        while (lo + step < hi && bases[lo + step] <= i) { lo += step; step <<= 1; }
        if (lo + step < hi) hi = lo + step;
    } else {
        hi = g;
        uint32_t step = 1;
        while (hi > lo + step && bases[hi - step] > i) { hi -= step; step <<= 1; }
        if (hi > lo + step) lo = hi - step;
    }
    while (hi - lo > 1u) {
        uint32_t mid = (lo + hi) >> 1;
        if (bases[mid] <= i) lo = mid; else hi = mid;
    }
    return lo * QUEUE_SEG + (i - bases[lo]);
}

}  // namespace psm
