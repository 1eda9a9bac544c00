// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// Replaces psm::RadixSort::sort (Include/Prismarine/Radix.hpp:47-74) and the shaders it drives
// (ShadersSDK/radix/{histogram,pfx-work,permute}.comp): 8 passes x 8 bits, ascending, stable,
// result back in the input buffers.  The reference runs 32 workgroups, three dispatches per pass, and
// sorts each 256-key block with eight 1-bit ballot splits.  Here:
//   radix_hist_all   ONE sweep over the keys builds all eight 256-bin digit histograms (8 KB of LDS per
//                    workgroup, one atomic per non-empty bin at the end): 8 B/key read once, not per pass
//   radix_onesweep   one launch per pass: a tile of 256*ITEMS keys is ranked with wave64 match-any ballots,
//                    learns where its digits start by a decoupled look-back over the tiles before it
//                    (one 32-bit {count | flag} word per tile and digit), is staged through LDS in digit
//                    order and written out as coalesced runs: 12 B/key read + 12 B/key written per pass
// = 200 B/key and 10 launches, against 256 B/key and 24 launches for the three-kernel passes
// (radix_hist / radix_scan / radix_scatter below). Measured on MI355X (tools/sort_bench.py) the
// three-kernel passes are FASTER at every size -- 0.113 vs 0.146 ms for 262 267 keys, 0.21 vs 0.53 ms for 2 M,
// 0.77 vs 0.93 ms for 10 M (round 3's tile shapes, pass_tile below): a look-back hop is a ~1 us round trip through L2 across XCDs and the tiles of a pass
// all start together (the chip holds as many tiles as a pass has), so the look-back chain costs more than the
// two extra launches and the 8 B/key it saves. The three-kernel pass is therefore the default;
// psm_sort_set_algorithm(ctx, 1) selects the one-sweep sort, and the parity tests run both.
//
// Look-back protocol (cdna_hip_programming.md Guideline 16, "R2 granule"): a status word carries value and
// flag together and is written by ONE relaxed agent-scope atomic store and read by relaxed agent-scope atomic
// loads (sc1: served by L2, never by a stale L1 line), so no fence is needed. Tile numbers are tickets of an
// atomic counter, not blockIdx: a tile's predecessors have all started before it, and none of them waits for
// a later tile, so every spin ends; spins are bounded all the same and a timeout raises an error word the
// host checks (psm_sort fails loudly rather than hang).
#include <algorithm>

#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

constexpr uint32_t ST_AGGREGATE = 1u << 30;  // the tile's own digit count is in the low 30 bits
constexpr uint32_t ST_INCLUSIVE = 2u << 30;  // count of this digit in tiles 0..t
constexpr uint32_t ST_VALUE = (1u << 30) - 1u;
constexpr uint32_t LOOKBACK_SPINS = 1u << 22;

// control block of one sort, zeroed by one memset: [0..2047] global digit histograms [pass][digit],
// [2048..2055] tile tickets per pass, [2056] error word
constexpr uint32_t CB_TICKET = 2048, CB_ERROR = 2056, CB_WORDS = 2064;

__global__ __launch_bounds__(256) void radix_hist_all(const uint64_t* __restrict__ keys, uint32_t* __restrict__ cb,
                                                      uint32_t n_max, const uint32_t* __restrict__ d_n, uint32_t per_block) {
    __shared__ uint32_t h[8][256];
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 8; p++) h[p][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * per_block;
    const uint32_t end = min(base + per_block, n);
    for (uint32_t i = base + tid; i < end; i += 256) {
        const uint64_t k = keys[i];
#pragma unroll
        for (int p = 0; p < 8; p++) atomicAdd(&h[p][(uint32_t)(k >> (8 * p)) & 255u], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const uint32_t v = h[p][tid];
        if (v) atomicAdd(&cb[p * 256 + tid], v);
    }
}

template <int ITEMS>
__global__ __launch_bounds__(256) void radix_onesweep(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                      uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                      uint32_t* __restrict__ cb, uint32_t* __restrict__ status,
                                                      uint32_t n_max, const uint32_t* __restrict__ d_n, int pass) {
    constexpr uint32_t TILE = 256 * ITEMS;
    __shared__ uint64_t sk[TILE];
    __shared__ uint32_t sv[TILE];
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t tstart[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t tmp[8];
    __shared__ uint32_t s_tile;
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t tid = threadIdx.x;
    const int shift = pass * 8;
    if (tid == 0) s_tile = atomicAdd(&cb[CB_TICKET + pass], 1u);
#pragma unroll
    for (int q = 0; q < 4; q++) wcount[q][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * TILE;
    if (base >= n) return;  // the launch is sized for n_max; tiles past the device-side count have nothing to do
    const uint32_t w = tid >> 6;
    const int l = lane_id();
    const uint64_t lt = lanemask_lt();

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
    {   // thread tid owns digit tid
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t t = wcount[q][tid];
            wcount[q][tid] = run;
            run += t;
        }
        uint32_t* mine = status + (size_t)tile * 256 + tid;
        __hip_atomic_store(mine, (run & ST_VALUE) | (tile == 0 ? ST_INCLUSIVE : ST_AGGREGATE), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        // decoupled look-back: keys with this digit in the tiles before this one. A window of LB predecessors is
        // loaded at once (independent loads in flight) and consumed nearest first, so a long walk over tiles that
        // have only published their own counts costs one memory latency per LB tiles, not per tile.
        uint32_t excl = 0;
        if (tile != 0) {
            constexpr int LB = 8;
            uint32_t t = tile - 1, spins = 0;
            bool done = false;
            while (!done) {
                uint32_t sw[LB];
#pragma unroll
                for (int j = 0; j < LB; j++)
                    sw[j] = (t >= (uint32_t)j) ? __hip_atomic_load(status + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : 0u;
#pragma unroll
                for (int j = 0; j < LB; j++) {
                    if (done || t == 0xFFFFFFFFu) break;
                    const uint32_t flag = sw[j] & ~ST_VALUE;
                    if (flag == 0u) break;          // not published yet: poll again from this tile
                    excl += sw[j] & ST_VALUE;
                    if (flag == ST_INCLUSIVE || t == 0) done = true;
                    else t--;
                }
                if (!done) {
                    if (++spins > LOOKBACK_SPINS) { atomicOr(&cb[CB_ERROR], 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(mine, ((excl + run) & ST_VALUE) | ST_INCLUSIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t ts = block_scan_excl<256>(run, tmp, nullptr);
        tstart[tid] = ts;
        uint32_t dbase = block_scan_excl<256>(cb[pass * 256 + tid], tmp, nullptr);  // keys with a smaller digit
        gbase[tid] = dbase + excl;
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

// ---- the three-kernel pass (default, psm_sort_set_algorithm(ctx, 0)): per pass a per-tile histogram, a scan of the
// tile counts per digit, and the scatter ----

template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void radix_hist(const uint64_t* __restrict__ keys, uint32_t* __restrict__ ghist,
                                                      uint32_t numTiles, uint32_t n_max,
                                                      const uint32_t* __restrict__ d_n, int shift) {
    constexpr uint32_t TILE = THREADS * ITEMS;
    __shared__ uint32_t h[256];
    uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t base = tile * TILE;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    if (base < n) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            uint32_t idx = base + i * THREADS + tid;
            if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) ghist[tid * numTiles + tile] = h[tid];
}

// One workgroup per digit: exclusive scan of that digit's row of per-tile counts in place
// (row-major ghist[digit][tile]); the row total goes to totals[digit]. The cross-digit base is a
// 256-wide scan that every scatter workgroup redoes from `totals` (pfx-work.comp:34-70 did both
// scans in ONE workgroup for the whole grid).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void radix_scan(uint32_t* __restrict__ g, uint32_t numTiles,
                                                      uint32_t* __restrict__ totals) {
    __shared__ uint32_t tmp[THREADS / 64 + 1];
    uint32_t* row = g + (size_t)blockIdx.x * numTiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < numTiles; base += THREADS) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < numTiles ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_scan_excl<THREADS>(v, tmp, &total);
        if (i < numTiles) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void radix_scatter(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                         uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                         const uint32_t* __restrict__ gscan,
                                                         const uint32_t* __restrict__ totals, uint32_t numTiles,
                                                         uint32_t n_max, const uint32_t* __restrict__ d_n, int shift) {
    constexpr uint32_t TILE = THREADS * ITEMS;
    constexpr int NW = THREADS / 64;
    __shared__ uint64_t sk[TILE];
    __shared__ uint32_t sv[TILE];
    __shared__ uint32_t wcount[NW][256];
    __shared__ uint32_t tstart[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t tmp[THREADS / 64 + 1];
    uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t base = tile * TILE;
    if (base >= n) return;
    uint32_t w = tid >> 6;
    int l = lane_id();
    uint64_t lt = lanemask_lt();
    for (uint32_t q = tid; q < NW * 256u; q += THREADS) (&wcount[0][0])[q] = 0;
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
    {   // thread tid < 256 owns digit tid (the other threads of a wider workgroup only take part in the scans)
        uint32_t run = 0;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < NW; q++) {
                uint32_t t = wcount[q][tid];
                wcount[q][tid] = run;
                run += t;
            }
        }
        uint32_t ts = block_scan_excl<THREADS>(run, tmp, nullptr);
        uint32_t dbase = block_scan_excl<THREADS>(tid < 256 ? totals[tid] : 0u, tmp, nullptr);  // keys with a smaller digit
        if (tid < 256) {
            tstart[tid] = ts;
            gbase[tid] = dbase + gscan[tid * numTiles + tile];
        }
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
        uint32_t j = i * THREADS + tid;
        if (j < tileN) {
            uint64_t key = sk[j];
            uint32_t d = (uint32_t)(key >> shift) & 255u;
            uint32_t gp = gbase[d] + (j - tstart[d]);
            kout[gp] = key;
            vout[gp] = sv[j];
        }
    }
}

// ping-pong buffers and histogram / status words of the sort, grown on demand. Every growth bumps sort_gen: a captured
// build graph (psm_bvh_build) holds these pointers and is re-captured when the generation has moved on.
static int sort_buffers(psm_ctx* c, size_t n_max, size_t E) {
    if (c->sort_cap < n_max) {
        if (c->sort_keys_tmp) (void)hipFree(c->sort_keys_tmp);
        if (c->sort_vals_tmp) (void)hipFree(c->sort_vals_tmp);
        c->sort_keys_tmp = nullptr; c->sort_vals_tmp = nullptr; c->sort_cap = 0;
        c->sort_gen++;
        PSM_HIP(c, hipMalloc(&c->sort_keys_tmp, n_max * sizeof(uint64_t)));
        PSM_HIP(c, hipMalloc(&c->sort_vals_tmp, n_max * sizeof(uint32_t)));
        c->sort_cap = n_max;
    }
    if (c->sort_hist_cap < E) {
        if (c->sort_hist) (void)hipFree(c->sort_hist);
        c->sort_hist = nullptr; c->sort_hist_cap = 0;
        c->sort_gen++;
        PSM_HIP(c, hipMalloc(&c->sort_hist, E * sizeof(uint32_t)));
        c->sort_hist_cap = E;
    }
    return PSM_OK;
}

static bool sort_uses_passes(const psm_ctx* c, size_t n_max) { return c->sort_algorithm != 1 || n_max >= (1u << 30); }
// keys per tile of the three-kernel pass, measured on MI355X (round 3, tools/sort_bench.py, profiles/r03_sort_bench.txt):
// the wider the workgroup over a tile the better -- 1024 threads x 4 keys against round 2's 256 x 16 over the same 4096
// keys: 0.79 against 0.88 ms at 10 M keys (16 waves per CU hide the scatter's latency where 54 KB of LDS per tile allowed
// 8), 0.21 against 0.32 ms at 2 M (which ran 256 x 4); small sorts are launch-bound and want enough tiles to fill the chip
static uint32_t pass_tile(size_t n_max) {
    // (C3's 262 267 keys: 1024 threads x 2 sort in 0.113 ms against 0.122 alone on the chip -- but next to other frames' one-wave
    // traversal workgroups a 16-wave workgroup waits for room: the emulated 1/8 tile's frame is 3 % faster with 256 x 4, the
    // full frame equal, profiles/r03_tile_emulation.txt)
    if (n_max <= (1u << 19)) return 1024u;   // 256 threads x 4 keys
    return 4096u;                            // 1024 threads x 4
}
static size_t sort_words(const psm_ctx* c, size_t n_max) {
    uint32_t tile = sort_uses_passes(c, n_max) ? pass_tile(n_max) : 256u * (n_max <= (1u << 21) ? 4u : 16u);
    size_t numTiles = (n_max + tile - 1) / tile;
    return sort_uses_passes(c, n_max) ? (size_t)256 * numTiles + 256 : (size_t)CB_WORDS + (size_t)8 * numTiles * 256;
}

// allocate what launch_sort(c, ., ., n_max, .) will need, without launching anything (no allocation may happen while a
// stream is being captured)
int sort_reserve(psm_ctx* c, size_t n_max) {
    if (n_max == 0) return PSM_OK;
    if (n_max > 0xFFFFFFF0ull) return set_err(c, PSM_ERR_CAPACITY, "sort: n exceeds 32-bit indexing");
    return sort_buffers(c, n_max, sort_words(c, n_max));
}

template <int ITEMS, int THREADS>
static int sort_passes(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n) {
    constexpr uint32_t TILE = THREADS * ITEMS;
    uint32_t numTiles = (uint32_t)((n_max + TILE - 1) / TILE);
    size_t E = (size_t)256 * numTiles + 256;  // per-tile counts + 256 digit totals
    { int rc = sort_buffers(c, n_max, E); if (rc != PSM_OK) return rc; }
    uint64_t* kin = d_keys; uint32_t* vin = d_vals;
    uint64_t* kout = c->sort_keys_tmp; uint32_t* vout = c->sort_vals_tmp;
    for (int pass = 0; pass < 8; pass++) {  // Radix.hpp:57: 64-bit keys, 8 passes
        int shift = pass * 8;
        radix_hist<ITEMS, THREADS><<<numTiles, THREADS, 0, c->stream>>>(kin, c->sort_hist, numTiles, (uint32_t)n_max, d_n, shift);
        uint32_t* totals = c->sort_hist + (size_t)256 * numTiles;
        if (numTiles > 512u) radix_scan<1024><<<256, 1024, 0, c->stream>>>(c->sort_hist, numTiles, totals);   // (10 M keys: 2442 tiles, 3 strips instead of 10)
        else radix_scan<256><<<256, 256, 0, c->stream>>>(c->sort_hist, numTiles, totals);
        radix_scatter<ITEMS, THREADS><<<numTiles, THREADS, 0, c->stream>>>(kin, vin, kout, vout, c->sort_hist, totals, numTiles,
                                                              (uint32_t)n_max, d_n, shift);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

template <int ITEMS>
static int sort_onesweep(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n) {
    constexpr uint32_t TILE = 256 * ITEMS;
    uint32_t numTiles = (uint32_t)((n_max + TILE - 1) / TILE);
    size_t E = (size_t)CB_WORDS + (size_t)8 * numTiles * 256;  // control block + status words of the 8 passes
    { int rc = sort_buffers(c, n_max, E); if (rc != PSM_OK) return rc; }
    uint32_t* cb = c->sort_hist;
    PSM_HIP(c, hipMemsetAsync(cb, 0, E * sizeof(uint32_t), c->stream));
    // histogram blocks: enough to fill the chip, at least 4096 keys each
    uint32_t hb = (uint32_t)std::min<size_t>((n_max + 4095) / 4096, 2048);
    uint32_t per_block = (uint32_t)((n_max + hb - 1) / hb);
    per_block = (per_block + 255u) & ~255u;
    hb = (uint32_t)((n_max + per_block - 1) / per_block);
    radix_hist_all<<<hb, 256, 0, c->stream>>>(d_keys, cb, (uint32_t)n_max, d_n, per_block);
    uint64_t* kin = d_keys; uint32_t* vin = d_vals;
    uint64_t* kout = c->sort_keys_tmp; uint32_t* vout = c->sort_vals_tmp;
    for (int pass = 0; pass < 8; pass++) {  // Radix.hpp:57: 64-bit keys, 8 passes
        radix_onesweep<ITEMS><<<numTiles, 256, 0, c->stream>>>(kin, vin, kout, vout, cb, cb + CB_WORDS + (size_t)pass * numTiles * 256,
                                                               (uint32_t)n_max, d_n, pass);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    PSM_HIP(c, hipGetLastError());
    c->sort_error_word = cb + CB_ERROR;  // checked where the caller synchronises anyway (psm_sort_check)
    return PSM_OK;
}

int launch_sort(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n) {
    if (n_max == 0) return PSM_OK;
    if (n_max > 0xFFFFFFF0ull) return set_err(c, PSM_ERR_CAPACITY, "sort: n exceeds 32-bit indexing");
    TimedScope ts(c, CAT_SORT);
    if (sort_uses_passes(c, n_max)) {  // (the one-sweep status words hold 30-bit counts)
        c->sort_error_word = nullptr;  // no look-back, nothing to time out (and the buffer it pointed into is reused)
        if (n_max <= (1u << 19)) return sort_passes<4, 256>(c, d_keys, d_vals, n_max, d_n);
        return sort_passes<4, 1024>(c, d_keys, d_vals, n_max, d_n);
    }
    if (n_max <= (1u << 21)) return sort_onesweep<4>(c, d_keys, d_vals, n_max, d_n);
    return sort_onesweep<16>(c, d_keys, d_vals, n_max, d_n);
}

// the look-back's timeout word of the last sort on this context; synchronises. PSM_OK or PSM_ERR_STATE.
int sort_check(psm_ctx* c) {
    if (!c->sort_error_word) return PSM_OK;
    uint32_t e = 0;
    PSM_HIP(c, hipMemcpyAsync(&e, c->sort_error_word, 4, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (e) return set_err(c, PSM_ERR_STATE, "radix sort: a look-back spin timed out (a predecessor tile never published)");
    return PSM_OK;
}

}  // namespace psm
