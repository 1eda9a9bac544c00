// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// Replaces psm::RadixSort::sort (Include/Prismarine/Radix.hpp:47-74) and the shaders it drives
// (ShadersSDK/radix/{histogram,pfx-work,permute}.comp): 8 passes x 8 bits, ascending, stable,
// result back in the input buffers.  The reference runs 32 workgroups and sorts each 256-key
// block with eight 1-bit ballot splits; here every tile of 256*ITEMS keys is ranked with wave64
// match-any ballots, staged through LDS in digit order and written out as coalesced runs.
//
// Per pass: radix_hist (LDS digit histogram per tile) -> radix_scan (one workgroup per digit scans
// its row of tile counts) -> radix_scatter (rank + LDS staging + coalesced bucket writes).
#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

template <int ITEMS>
__global__ __launch_bounds__(256) void radix_hist(const uint64_t* __restrict__ keys, uint32_t* __restrict__ ghist,
                                                  uint32_t numTiles, uint32_t n_max,
                                                  const uint32_t* __restrict__ d_n, int shift) {
    constexpr uint32_t TILE = 256 * ITEMS;
    __shared__ uint32_t h[256];
    uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t base = tile * TILE;
    h[tid] = 0;
    __syncthreads();
    if (base < n) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            uint32_t idx = base + i * 256 + tid;
            if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    ghist[tid * numTiles + tile] = h[tid];
}

// One workgroup per digit: exclusive scan of that digit's row of per-tile counts in place
// (row-major ghist[digit][tile]); the row total goes to totals[digit]. The cross-digit base is a
// 256-wide scan that every scatter workgroup redoes from `totals` (pfx-work.comp:34-70 did both
// scans in ONE workgroup for the whole grid).
__global__ __launch_bounds__(256) void radix_scan(uint32_t* __restrict__ g, uint32_t numTiles,
                                                  uint32_t* __restrict__ totals) {
    __shared__ uint32_t tmp[8];
    uint32_t* row = g + (size_t)blockIdx.x * numTiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < numTiles; base += 256) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < numTiles ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_scan_excl<256>(v, tmp, &total);
        if (i < numTiles) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int ITEMS>
__global__ __launch_bounds__(256) void radix_scatter(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                     uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                     const uint32_t* __restrict__ gscan,
                                                     const uint32_t* __restrict__ totals, uint32_t numTiles,
                                                     uint32_t n_max, const uint32_t* __restrict__ d_n, int shift) {
    constexpr uint32_t TILE = 256 * ITEMS;
    __shared__ uint64_t sk[TILE];
    __shared__ uint32_t sv[TILE];
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t tstart[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t tmp[8];
    uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t base = tile * TILE;
    if (base >= n) return;
    uint32_t w = tid >> 6;
    int l = lane_id();
    uint64_t lt = lanemask_lt();
#pragma unroll
    for (int q = 0; q < 4; q++) wcount[q][tid] = 0;
    __syncthreads();

    uint64_t k[ITEMS];
    uint32_t v[ITEMS], r[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        bool valid = idx < n;
        k[i] = valid ? kin[idx] : ~0ull;
        v[i] = valid ? vin[idx] : 0u;
    }
    volatile uint32_t* wc = &wcount[w][0];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        bool valid = idx < n;
        uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t before = (uint32_t)__popcll(peers & lt);
        uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t old = valid ? wc[d] : 0u;
        if (valid && before == 0) wc[d] = old + cnt;
        r[i] = old + before;
    }
    __syncthreads();
    {
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t t = wcount[q][tid];
            wcount[q][tid] = run;
            run += t;
        }
        uint32_t ts = block_scan_excl<256>(run, tmp, nullptr);
        tstart[tid] = ts;
        uint32_t dbase = block_scan_excl<256>(totals[tid], tmp, nullptr);  // keys with a smaller digit
        gbase[tid] = dbase + gscan[tid * numTiles + tile];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        if (idx < n) {
            uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
            uint32_t pos = tstart[d] + wcount[w][d] + r[i];
            sk[pos] = k[i];
            sv[pos] = v[i];
        }
    }
    __syncthreads();
    uint32_t tileN = min(TILE, n - base);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t j = i * 256 + tid;
        if (j < tileN) {
            uint64_t key = sk[j];
            uint32_t d = (uint32_t)(key >> shift) & 255u;
            uint32_t gp = gbase[d] + (j - tstart[d]);
            kout[gp] = key;
            vout[gp] = sv[j];
        }
    }
}

template <int ITEMS>
static int sort_passes(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n) {
    constexpr uint32_t TILE = 256 * ITEMS;
    uint32_t numTiles = (uint32_t)((n_max + TILE - 1) / TILE);
    size_t E = (size_t)256 * numTiles + 256;  // per-tile counts + 256 digit totals
    if (c->sort_cap < n_max) {
        if (c->sort_keys_tmp) (void)hipFree(c->sort_keys_tmp);
        if (c->sort_vals_tmp) (void)hipFree(c->sort_vals_tmp);
        c->sort_keys_tmp = nullptr; c->sort_vals_tmp = nullptr; c->sort_cap = 0;
        PSM_HIP(c, hipMalloc(&c->sort_keys_tmp, n_max * sizeof(uint64_t)));
        PSM_HIP(c, hipMalloc(&c->sort_vals_tmp, n_max * sizeof(uint32_t)));
        c->sort_cap = n_max;
    }
    if (c->sort_hist_cap < E) {
        if (c->sort_hist) (void)hipFree(c->sort_hist);
        c->sort_hist = nullptr; c->sort_hist_cap = 0;
        PSM_HIP(c, hipMalloc(&c->sort_hist, E * sizeof(uint32_t)));
        c->sort_hist_cap = E;
    }
    uint64_t* kin = d_keys; uint32_t* vin = d_vals;
    uint64_t* kout = c->sort_keys_tmp; uint32_t* vout = c->sort_vals_tmp;
    for (int pass = 0; pass < 8; pass++) {  // Radix.hpp:57: 64-bit keys, 8 passes
        int shift = pass * 8;
        radix_hist<ITEMS><<<numTiles, 256, 0, c->stream>>>(kin, c->sort_hist, numTiles, (uint32_t)n_max, d_n, shift);
        uint32_t* totals = c->sort_hist + (size_t)256 * numTiles;
        radix_scan<<<256, 256, 0, c->stream>>>(c->sort_hist, numTiles, totals);
        radix_scatter<ITEMS><<<numTiles, 256, 0, c->stream>>>(kin, vin, kout, vout, c->sort_hist, totals, numTiles,
                                                              (uint32_t)n_max, d_n, shift);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_sort(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n) {
    if (n_max == 0) return PSM_OK;
    if (n_max > 0xFFFFFFF0ull) return set_err(c, PSM_ERR_CAPACITY, "sort: n exceeds 32-bit indexing");
    TimedScope ts(c, CAT_SORT);
    if (n_max <= (1u << 21)) return sort_passes<4>(c, d_keys, d_vals, n_max, d_n);
    return sort_passes<16>(c, d_keys, d_vals, n_max, d_n);
}

}  // namespace psm
