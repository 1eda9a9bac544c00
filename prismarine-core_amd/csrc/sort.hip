// sort.hip -- stable radix sort of (u64 key, u32 value) pairs for gfx950.
//
// Replaces psm::RadixSort::sort (Include/Prismarine/Radix.hpp:47-74) and the shaders it drives
// (ShadersSDK/radix/{histogram,pfx-work,permute}.comp): 8 passes x 8 bits, ascending, stable,
// result back in the input buffers.  The reference runs 32 workgroups, three dispatches per pass, and
// sorts each 256-key block with eight 1-bit ballot splits.  Three implementations, one result (psm_sort_set_algorithm):
//   2 (default, round 5)  HYBRID: the LSD passes of the top sixteen key bits (radix_hist / radix_scan / radix_scatter, twice),
//                    then radix_local sorts chunks of whole sixteen-bit bins by the remaining digits in LDS and writes them
//                    back once -- 7 launches, a key moves through HBM three times (see "the hybrid sort" below)
//   0                radix_hist / radix_scan / radix_scatter for all eight passes: 24 launches, 256 B/key
//   1                radix_hist_all: ONE sweep over the keys builds all eight 256-bin digit histograms (8 KB of LDS per
//                    workgroup, one atomic per non-empty bin at the end): 8 B/key read once, not per pass; then per pass
//                    radix_onesweep: a tile of 256*ITEMS keys is ranked with wave64 match-any ballots, learns where its
//                    digits start by a decoupled look-back over the tiles before it (one 32-bit {count | flag} word per tile
//                    and digit), is staged through LDS in digit order and written out as coalesced runs -- 200 B/key and
//                    10 launches, and SLOWER than 0 at every size: a look-back hop is a ~1 us round trip through
//                    L2 across XCDs and the tiles of a pass all start together (the chip holds as many tiles as a pass has)
// tools/sort_bench.py on MI355X (profiles/r05_sort_bench.txt): 262 267 Morton codes 0.049 / 0.104 / 0.135 ms (hybrid / 0 / 1),
// 2 M keys 0.104 / 0.178 / 0.515, 10 M Morton codes 0.434 / 0.681 / 0.900. The parity tests run all three.
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

// Stable ranks of one wave round (64 keys, one per lane): r = keys of this wave with the same digit in earlier rounds or in
// lower lanes. `row`: the wave's own 256 digit counts in LDS, updated by the lowest lane of each group of equal digits.
// Match-any by eight ballots; per bit the mask of lanes that agree with this lane is ~(ballot ^ s), s = the lane's bit spread
// over a word, so a bit costs v_bfe_i32, v_cmp and one three-input v_bitop3_b32 per mask half: four instructions, by builtins (hipcc made
// eleven of `peers &= bit ? m : ~m` and six of the same thing written with xor and and-not); `before` is v_mbcnt of the final mask. The counts are read and
// written as LDS (address space 3) -- through a generic `volatile` pointer hipcc emitted flat_load / flat_store with sc0 sc1
// and a full wait each (round 5: 8 of them per tile in radix_scatter).
template <typename CT>
PSM_D uint32_t wave_rank(uint32_t d, bool valid, CT* row) {
    typedef __attribute__((address_space(3))) volatile CT lds_ct;
    lds_ct* wc = (lds_ct*)row;
    const uint64_t vm = __ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);   // the lane's bit b, spread over the word
        const uint64_t m = __builtin_amdgcn_uicmp(s, 0u, 33);                // ballot(s != 0) (33: ICMP_NE)
        plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, s, 0x90);        // plo & ~(m ^ s)
        phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), s, 0x90);
    }
    const uint32_t before = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    const uint32_t cnt = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
    const uint32_t old = wc[d];
    if (valid && before == 0u) wc[d] = (CT)(old + cnt);
    return old + before;
}

// (Round 5 also tried the match through LDS -- every lane ORs its lane bit into a per-wave mask word of its digit with ds_or_b64 and
// reads the word back: 12 vector instructions a round instead of 45 -- and it LOST: C5's sort 0.52 against 0.41 ms, C3's 0.052
// against 0.049, profiles/r05_sort_sweep.txt: LDS atomics are slower than the ballots they replace. Not kept.)

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

    uint64_t k[ITEMS];
    uint32_t v[ITEMS], r[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        bool valid = idx < n;
        k[i] = valid ? kin[idx] : ~0ull;
        v[i] = valid ? vin[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        r[i] = wave_rank((uint32_t)(k[i] >> shift) & 255u, idx < n, &wcount[w][0]);
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
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        uint32_t idx = base + w * (64 * ITEMS) + i * 64 + l;
        r[i] = wave_rank((uint32_t)(k[i] >> shift) & 255u, idx < n, &wcount[w][0]);
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

// ---- the hybrid sort (default, psm_sort_set_algorithm(ctx, 2)): two global passes, the rest in LDS ------------------------
// The LSD sort above moves every key through HBM eight times and takes 24 dependent launches (0.12 ms for C3's 262 267 keys,
// all of it launch latency; 0.73 ms for C5's 10 M, 0.34 of the HBM peak of its 200 B/key). The hybrid sort moves a key three
// times in seven launches:
//   1. the LSD passes of the TOP sixteen key bits first -- digit [pshift, pshift + 8) then [pshift + 8, pshift + 16), with the
//      three kernels above, stable on the input order: the array is then partitioned into up to 65 536 bins by its top bits,
//      bins in ascending order, input order inside a bin;
//   2. radix_local: workgroup c takes the bins that START in [c S, (c + 1) S) -- whole bins, so chunks are ordered among
//      themselves -- and sorts them by every remaining digit in LDS (stable LSD passes over the digits in which the chunk's
//      keys differ at all, wave64 match-any ranks as in radix_scatter), then writes the chunk back where it lay.
// A chunk is at most S keys plus the tail of its last bin; it fits the workgroup's LDS (CAP keys) as long as no bin is longer
// than CAP - S. Morton codes of meshes spread well over sixteen bits (C3: longest bin 1 876 of 262 267 keys, C5: 723 of 10 M,
// bits 47..62); for a bin that does not, the workgroup sorts its chunk through global memory on its own (local_slow: correct
// for every input, slow) and raises `overflow`, a pinned host word launch_sort looks at: a context whose keys overflowed
// goes back to the eight-pass sort for good. The result is the stable ascending order -- the same bits as the LSD sorts.
//
// In place: a workgroup reads the window [c S - 1, c S + CAP) of the keys -- beyond its own chunk on both sides, to find the
// bin boundaries -- while its neighbours may already be writing their sorted chunks into the same array. What it looks at in
// a neighbour's keys is the bin they belong to (key >> pshift), and sorting a chunk never moves a key out of the positions of
// its bin: whichever version of a neighbour's key a load returns, its bin is the same (pshift >= 32: the bin bits lie in the
// key's high dword, so even a load torn between two 32-bit halves would agree).

#ifndef PSM_EXP_SORTLOG
#define PSM_EXP_SORTLOG 0   // 1: an experiment build (make sortlog) whose radix_local workgroups log s_memtime at their phase boundaries (tests/studies/sort_log.py)
#endif
#if PSM_EXP_SORTLOG
__device__ unsigned long long* g_sortlog = nullptr;   // 32 stamps per workgroup (set by psm_sortlog_set, an export of the experiment build only)
#define SORT_STAMP(k) do { if (g_sortlog && threadIdx.x == 0 && (k) < 32) g_sortlog[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SORT_STAMP(k) do { } while (0)
#endif

PSM_D uint32_t key_bin(uint64_t k, int pshift) { return pshift >= 64 ? 0u : (uint32_t)(k >> pshift); }

template <int CAP, int THREADS>
struct LocalLds {
    static constexpr int NW = THREADS / 64;
    uint64_t sk[CAP];
    uint32_t sv[CAP];
    uint16_t wcount[NW][256];   // per wave and digit: count, then exclusive prefix over the waves (a tile has at most CAP < 65 536 keys)
    uint32_t tpart[256];        // exclusive digit prefix inside each group of 64 digits
    uint32_t wtot[4];           // totals of the four groups
    uint32_t gbase[256];        // local_slow: where the next key of each digit goes
    uint32_t diff[2];
    uint32_t lo, hi;
};

// the chunk does not fit LDS: stable LSD passes over [gs, gs + m) through global memory, this workgroup alone, the scratch
// arrays' own [gs, gs + m) as the other buffer
template <int CAP, int THREADS>
PSM_D void local_slow(LocalLds<CAP, THREADS>& S, uint64_t* keys, uint32_t* vals, uint64_t* altk, uint32_t* altv, uint32_t gs, uint32_t m) {
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int l = lane_id();
    {
        const uint64_t k0 = keys[gs];
        uint64_t dif = 0;
        for (uint32_t j = tid; j < m; j += THREADS) dif |= keys[gs + j] ^ k0;
        if (dif) { atomicOr(&S.diff[0], (uint32_t)dif); atomicOr(&S.diff[1], (uint32_t)(dif >> 32)); }
    }
    __syncthreads();
    const uint64_t diff = ((uint64_t)S.diff[1] << 32) | S.diff[0];
    uint64_t* sk = keys; uint32_t* sv = vals;
    uint64_t* dk = altk; uint32_t* dv = altv;
    for (int p = 0; p < 8; p++) {
        const int shift = 8 * p;
        if (((diff >> shift) & 255ull) == 0ull) continue;
        if (tid < 256) S.gbase[tid] = 0;
        __syncthreads();
        for (uint32_t j = tid; j < m; j += THREADS) atomicAdd(&S.gbase[(uint32_t)(sk[gs + j] >> shift) & 255u], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 digit counts
            uint32_t c = tid < 256 ? S.gbase[tid] : 0u, inc = 0;
            if (tid < 256) {
                inc = wave_scan_incl(c);
                if (l == 63) S.wtot[w] = inc;
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t pre = 0;
                for (uint32_t q = 0; q < w; q++) pre += S.wtot[q];
                S.gbase[tid] = pre + inc - c;
            }
            __syncthreads();
        }
        for (uint32_t t0 = 0; t0 < m; t0 += CAP) {
            const uint32_t tileN = min((uint32_t)CAP, m - t0);
            uint64_t k[ITEMS];
            uint32_t v[ITEMS], r[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const uint32_t q = w * (64 * ITEMS) + i * 64 + l;
                const bool valid = q < tileN;
                k[i] = valid ? sk[gs + t0 + q] : 0ull;
                v[i] = valid ? sv[gs + t0 + q] : 0u;
            }
            for (int j = l; j < 256; j += 64) S.wcount[w][j] = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const uint32_t q = w * (64 * ITEMS) + i * 64 + l;
                r[i] = wave_rank((uint32_t)(k[i] >> shift) & 255u, q < tileN, &S.wcount[w][0]);
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t run = 0;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    const uint32_t t = S.wcount[q][tid];
                    S.wcount[q][tid] = (uint16_t)run;
                    run += t;
                }
                const uint32_t base = S.gbase[tid];
                S.tpart[tid] = base;
                S.gbase[tid] = base + run;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const uint32_t q = w * (64 * ITEMS) + i * 64 + l;
                if (q < tileN) {
                    const uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
                    const uint32_t pos = gs + S.tpart[d] + S.wcount[w][d] + r[i];
                    dk[pos] = k[i];
                    dv[pos] = v[i];
                }
            }
            __syncthreads();
        }
        // the next pass reads what other waves of this workgroup have just written: stores out to L2, no stale line in L1
        __threadfence();
        __syncthreads();
        __threadfence();
        { uint64_t* t = sk; sk = dk; dk = t; }
        { uint32_t* t = sv; sv = dv; dv = t; }
    }
    if (sk != keys) {
        for (uint32_t j = tid; j < m; j += THREADS) { keys[gs + j] = sk[gs + j]; vals[gs + j] = sv[gs + j]; }
    }
}

// (second launch bound: waves per SIMD such that two workgroups share a CU while their LDS -- 160 KB a CU -- allows it)
template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS, (2 * sizeof(LocalLds<CAP, THREADS>) <= 160 * 1024 ? 2 : 1) * THREADS / 256) void radix_local(uint64_t* keys, uint32_t* vals, uint64_t* altk, uint32_t* altv,
                                                       uint32_t n_max, const uint32_t* __restrict__ d_n, uint32_t S, int pshift,
                                                       uint32_t* overflow) {
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    static_assert(CAP % THREADS == 0 && THREADS % 64 == 0 && THREADS >= 256 && CAP < 65536, "tile shape");
    __shared__ LocalLds<CAP, THREADS> L;
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t a = blockIdx.x * S;
    if (a >= n) return;
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int l = lane_id();
    SORT_STAMP(0);
    const uint32_t winN = min((uint32_t)CAP, n - a);   // the window: keys [a, a + winN)
    // ... of which the first S + 1024 are loaded at once and the rest only when the chunk's last bin turns out to be longer than
    // 1024 keys (C5: never; the second half of the window would be 8 B/key more of HBM reads for nothing)
    const uint32_t win1 = min(winN, S + 1024u);
    if (tid == 0) { L.lo = 0xFFFFFFFFu; L.hi = 0xFFFFFFFFu; L.diff[0] = 0; L.diff[1] = 0; }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const uint32_t j = i * THREADS + tid;
        if (j < win1) L.sk[j] = keys[a + j];
    }
    const uint32_t bin_before = a > 0 ? key_bin(keys[a - 1], pshift) : 0u;
    __syncthreads();
    // the chunk: from the first bin boundary at or after a to the first one at or after a + S (a boundary: position 0, or a key
    // whose bin differs from its predecessor's)
    {
        uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t j = i * THREADS + tid;
            if (j < win1) {
                const bool edge = (a + j == 0u) || key_bin(L.sk[j], pshift) != (j == 0u ? bin_before : key_bin(L.sk[j - 1], pshift));
                if (edge) {
                    lo = min(lo, j);
                    if (j >= S) hi = min(hi, j);
                }
            }
        }
        if (lo != 0xFFFFFFFFu) atomicMin(&L.lo, lo);
        if (hi != 0xFFFFFFFFu) atomicMin(&L.hi, hi);
    }
    __syncthreads();
    if (L.hi == 0xFFFFFFFFu && win1 < winN && L.lo < S) {   // (the same for every thread) the rest of the window
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t j = i * THREADS + tid;
            if (j >= win1 && j < winN) L.sk[j] = keys[a + j];
        }
        __syncthreads();
        uint32_t hi = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t j = i * THREADS + tid;
            if (j >= win1 && j < winN && key_bin(L.sk[j], pshift) != key_bin(L.sk[j - 1], pshift)) hi = min(hi, j);
        }
        if (hi != 0xFFFFFFFFu) atomicMin(&L.hi, hi);
        __syncthreads();
    }
    SORT_STAMP(1);
    const uint32_t js = L.lo;
    if (js == 0xFFFFFFFFu || js >= S) return;   // no bin starts in [a, a + S): an earlier workgroup's chunk covers this stretch
    uint32_t je = L.hi;
    if (je == 0xFFFFFFFFu && a + winN >= n) je = winN;   // the array ends inside the window
    if (je == 0xFFFFFFFFu) {
        // the chunk's last bin runs past the window: its end is the first key of a later bin (the keys are ordered by bin)
        const uint32_t bin = key_bin(L.sk[winN - 1], pshift);
        uint32_t lo = a + winN, hi = n;   // first position with a later bin lies in [lo, hi]
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (key_bin(keys[mid], pshift) > bin) hi = mid; else lo = mid + 1u;
        }
        const uint32_t gs = a + js, m = lo - gs;
        __syncthreads();
        local_slow<CAP, THREADS>(L, keys, vals, altk, altv, gs, m);
        if (tid == 0 && overflow) __hip_atomic_fetch_add(overflow, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const uint32_t size = je - js, gs = a + js;
    // wave w works on slots [w per, (w + 1) per) of the chunk, 64 per round
    const uint32_t rounds = (size + NW * 64u - 1u) / (NW * 64u), per = rounds * 64u;
    uint64_t k[ITEMS];
    uint32_t v[ITEMS], r[ITEMS];
    uint64_t dif = 0;
    const uint64_t k0 = L.sk[js];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const uint32_t q = w * per + i * 64 + l;
        const bool valid = (uint32_t)i < rounds && q < size;
        k[i] = valid ? L.sk[js + q] : k0;
        v[i] = valid ? vals[gs + q] : 0u;
        dif |= k[i] ^ k0;
    }
    {
        uint32_t dl = (uint32_t)dif, dh = (uint32_t)(dif >> 32);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { dl |= __shfl_xor(dl, d, 64); dh |= __shfl_xor(dh, d, 64); }
        if (l == 0 && (dl | dh)) { atomicOr(&L.diff[0], dl); atomicOr(&L.diff[1], dh); }
    }
    __syncthreads();
    const uint64_t diff = ((uint64_t)L.diff[1] << 32) | L.diff[0];
    SORT_STAMP(2);
    if (diff == 0ull) return;   // one key value: the chunk is in order as it lies
    int stamp = 3;
    (void)stamp;
    for (int p = 0; p < 8; p++) {
        const int shift = 8 * p;
        if (((diff >> shift) & 255ull) == 0ull) continue;   // (the same for the whole workgroup)
        for (int j = l; j < 256; j += 64) L.wcount[w][j] = 0;   // the wave's own row: nobody else touches it before the barrier
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            if ((uint32_t)i < rounds) {
                const uint32_t q = w * per + i * 64 + l;
                r[i] = wave_rank((uint32_t)(k[i] >> shift) & 255u, q < size, &L.wcount[w][0]);
            }
        }
        __syncthreads();
        SORT_STAMP(stamp); stamp++;
        if (tid < 256) {   // thread d: exclusive prefix of digit d over the waves, then over the digits of its group of 64
            uint32_t run = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const uint32_t t = L.wcount[q][tid];
                L.wcount[q][tid] = (uint16_t)run;
                run += t;
            }
            const uint32_t inc = wave_scan_incl(run);
            L.tpart[tid] = inc - run;
            if (l == 63) L.wtot[w] = inc;
        }
        __syncthreads();
        SORT_STAMP(stamp); stamp++;
        {
            const uint32_t t0 = L.wtot[0], t1 = t0 + L.wtot[1], t2 = t1 + L.wtot[2];
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                if ((uint32_t)i < rounds) {
                    const uint32_t q = w * per + i * 64 + l;
                    if (q < size) {
                        const uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
                        const uint32_t g = d >> 6;
                        const uint32_t pos = (g == 0u ? 0u : g == 1u ? t0 : g == 2u ? t1 : t2) + L.tpart[d] + L.wcount[w][d] + r[i];
                        L.sk[pos] = k[i];
                        L.sv[pos] = v[i];
                    }
                }
            }
        }
        __syncthreads();
        SORT_STAMP(stamp); stamp++;
        if ((diff >> shift) >> 8 == 0ull) break;   // the last pass: written out below, straight from LDS
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            if ((uint32_t)i < rounds) {
                const uint32_t q = w * per + i * 64 + l;
                if (q < size) { k[i] = L.sk[q]; v[i] = L.sv[q]; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const uint32_t q = i * THREADS + tid;
        if (q < size) { keys[gs + q] = L.sk[q]; vals[gs + q] = L.sv[q]; }
    }
#if PSM_EXP_SORTLOG
    if (g_sortlog && tid == 0) { g_sortlog[(size_t)blockIdx.x * 32 + 30] = size; g_sortlog[(size_t)blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memtime(); }
#endif
}

#if PSM_EXP_SORTLOG
extern "C" int psm_sortlog_set(unsigned long long* d_log) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sortlog), &d_log, sizeof(d_log)) == hipSuccess ? 0 : -1;
}
#endif

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

// the algorithm a sort of this context runs: the one asked for (psm_sort_set_algorithm), except that a context whose keys have
// overflowed a hybrid chunk (radix_local's slow path raised the pinned word) sorts with the eight-pass kernels from then on
int sort_effective_algorithm(psm_ctx* c) {
    if (c->sort_algorithm == 2 && !c->sort_demoted && c->sort_overflow && *(volatile uint32_t*)c->sort_overflow != 0u) c->sort_demoted = true;
    return (c->sort_algorithm == 2 && c->sort_demoted) ? 0 : c->sort_algorithm;
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
    if (c->sort_algorithm == 2 && !c->sort_overflow) {   // the hybrid sort's pinned overflow word
        PSM_HIP(c, hipHostMalloc((void**)&c->sort_overflow, sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
        *c->sort_overflow = 0u;
    }
    return sort_buffers(c, n_max, sort_words(c, n_max));
}

// passes [first_shift, first_shift + 8 * count) of the LSD sort, eight bits each, lowest first; the result lies in the caller's
// buffers after an even number of passes, in the context's scratch buffers after an odd one
template <int ITEMS, int THREADS>
static int sort_passes(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n, int first_shift = 0, int count = 8) {
    constexpr uint32_t TILE = THREADS * ITEMS;
    uint32_t numTiles = (uint32_t)((n_max + TILE - 1) / TILE);
    size_t E = (size_t)256 * numTiles + 256;  // per-tile counts + 256 digit totals
    { int rc = sort_buffers(c, n_max, E); if (rc != PSM_OK) return rc; }
    uint64_t* kin = d_keys; uint32_t* vin = d_vals;
    uint64_t* kout = c->sort_keys_tmp; uint32_t* vout = c->sort_vals_tmp;
    for (int pass = 0; pass < count; pass++) {  // Radix.hpp:57: 64-bit keys, 8 passes
        int shift = first_shift + pass * 8;
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

// The hybrid sort (see radix_local): key_bits = the key bits that can be set at all (64; 63 for Morton codes) -- the two
// global passes take the sixteen bits below that.
static int sort_hybrid(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n, int key_bits) {
    if (!c->sort_overflow) {
        PSM_HIP(c, hipHostMalloc((void**)&c->sort_overflow, sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
        *c->sort_overflow = 0u;
    }
    if (n_max <= 4096u) {   // one chunk: no global pass (pshift 64: every key in bin 0)
        { int rc = sort_buffers(c, n_max, 256); if (rc != PSM_OK) return rc; }
        radix_local<4096, 1024><<<1, 1024, 0, c->stream>>>(d_keys, d_vals, c->sort_keys_tmp, c->sort_vals_tmp, (uint32_t)n_max, d_n, 4096u, 64, nullptr);
        PSM_HIP(c, hipGetLastError());
        return PSM_OK;
    }
    const int pshift = key_bits - 16;
    const bool small = n_max <= (1u << 19);
    int rc = small ? sort_passes<4, 256>(c, d_keys, d_vals, n_max, d_n, pshift, 2)
                   : sort_passes<4, 1024>(c, d_keys, d_vals, n_max, d_n, pshift, 2);   // (512- and 256-thread tiles: the same frame time with frames in flight, within the noise)
    if (rc != PSM_OK) return rc;
    // S: the stretch of bin starts a workgroup takes; a chunk fits LDS (CAP keys) while no bin is longer than CAP - S. Small sorts
    // take short stretches (1 024 of 4 096): as many workgroups as the chip has CUs matter more there than keys per workgroup.
    // Large ones 2 048 of 4 096 with 512-thread workgroups. Alone on the chip longer chunks under wider workgroups are faster -- a pass
    // costs a workgroup ~5 000 cycles of barriers and LDS round trips whatever the chunk holds (tools/sort_log.py): C5 0.43 ms this way,
    // 0.41 with 3 072 of 5 120 x 1 024 threads, profiles/r05_sort_sweep.txt -- but a sort shares the chip with the other frames'
    // one-wave traversal workgroups, and a 16-wave workgroup waits for a CU to empty: C5's frame with 4 frames in flight 11.40 ms this
    // way against 11.79 (profiles/r05_wg_shapes_in_flight.txt; the sort's launch in that trace: 2.6 ms instead of 0.25 alone)
    const uint32_t S = small ? c->sort_hybrid_s_small : c->sort_hybrid_s_large;
    const uint32_t cap = small ? c->sort_hybrid_cap_small : c->sort_hybrid_cap_large, threads = small ? c->sort_hybrid_threads : c->sort_hybrid_threads_large;
    const uint32_t grid = (uint32_t)((n_max + S - 1) / S);
#define PSM_LOCAL(CAP_, TH_) \
    if (cap == CAP_ && threads == TH_) { \
        radix_local<CAP_, TH_><<<grid, TH_, 0, c->stream>>>(d_keys, d_vals, c->sort_keys_tmp, c->sort_vals_tmp, (uint32_t)n_max, d_n, S, pshift, c->sort_overflow); \
        PSM_HIP(c, hipGetLastError()); \
        return PSM_OK; \
    }
    PSM_LOCAL(4096, 1024) PSM_LOCAL(4096, 512) PSM_LOCAL(5120, 1024) PSM_LOCAL(5120, 512) PSM_LOCAL(6144, 512)
#undef PSM_LOCAL
    return set_err(c, PSM_ERR_INVALID, "hybrid sort: no radix_local of this shape");
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

int launch_sort(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n, int key_bits) {
    if (n_max == 0) return PSM_OK;
    if (n_max > 0xFFFFFFF0ull) return set_err(c, PSM_ERR_CAPACITY, "sort: n exceeds 32-bit indexing");
    TimedScope ts(c, CAT_SORT);
    if (sort_uses_passes(c, n_max)) {  // (the one-sweep status words hold 30-bit counts)
        c->sort_error_word = nullptr;  // no look-back, nothing to time out (and the buffer it pointed into is reused)
        if (sort_effective_algorithm(c) == 2) return sort_hybrid(c, d_keys, d_vals, n_max, d_n, key_bits);
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
