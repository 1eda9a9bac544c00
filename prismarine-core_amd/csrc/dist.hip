// dist.hip -- tile-sharded frames across the GPUs of a node: the collectives behind the C ABI (psm_dist_*).
//
// The reference has no multi-GPU path. A frame shards by screen tile: 8-row bands dealt round-robin
// (psm_rt_set_tile_interleaved), the hierarchy is rebuilt redundantly on every GPU, and the path has exactly ONE
// data-path exchange per frame: the gather of per-texel radiance (16 B per texel) to rank 0, which runs the sampler.
// RCCL over xGMI, one process per GPU; every peer sends on its own direct link, rank 0 ingests 7 links in parallel.
// The only other exchange is a few ints per batch of frames for the reference's `fewer than 32 rays -> stop` rule on
// the frame's global ray count (psm_lanes_run_sharded), also provided here so a C++ host needs nothing but this ABI.
//
// Streams: all collectives of a communicator are issued on ITS stream in call order (every rank makes the same calls
// in the same order); a gather waits for the Pipeline's stream (event), packs, gathers, unpacks on rank 0, and the
// Pipeline's stream waits for that (event), so folding the frame afterwards needs no host synchronisation.
#include <rccl/rccl.h>

#include <climits>
#include <cstring>
#include <new>
#include <vector>

#include "psm_internal.h"

struct psm_dist {
    psm_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int rank = 0, world = 1;
    int tile_rank = 0, tile_world = 1;  // the tile geometry gathers use: (rank, world) unless psm_dist_emulate_tile changed it
    float* d_send = nullptr;   // per_floats
    float* d_recv = nullptr;   // rank 0: world * per_floats
    size_t per_floats = 0;
    int32_t* d_i32 = nullptr;  // all-gather staging: send | recv
    size_t i32_cap = 0;
};

namespace psm {
static int nccl_err(psm_ctx* c, ncclResult_t r, const char* what) {
    if (c) c->err = std::string(what) + ": " + ncclGetErrorString(r);
    return PSM_ERR_HIP;
}
#define PSM_NCCL(ctx, call)                                   \
    do {                                                      \
        ncclResult_t r__ = (call);                            \
        if (r__ != ncclSuccess) return psm::nccl_err((ctx), r__, #call); \
    } while (0)

}  // namespace psm

using namespace psm;

extern "C" {

int psm_dist_unique_id(uint8_t id[128]) {
    if (!id) return PSM_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PSM_ERR_HIP;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return PSM_OK;
}

int psm_dist_destroy(psm_dist* d) {
    if (!d) return PSM_ERR_INVALID;
    (void)hipSetDevice(d->ctx->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    if (d->d_send) (void)hipFree(d->d_send);
    if (d->d_recv) (void)hipFree(d->d_recv);
    if (d->d_i32) (void)hipFree(d->d_i32);
    if (d->ev_in) (void)hipEventDestroy(d->ev_in);
    if (d->ev_out) (void)hipEventDestroy(d->ev_out);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    return PSM_OK;
}

int psm_dist_init(psm_ctx* ctx, int rank, int world, const uint8_t id[128], psm_dist** out) {
    if (!ctx || !out || !id || world < 1 || rank < 0 || rank >= world) return PSM_ERR_INVALID;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    psm_dist* d = new (std::nothrow) psm_dist();
    if (!d) return PSM_ERR_INVALID;
    d->ctx = ctx; d->rank = rank; d->world = world;
    d->tile_rank = rank; d->tile_world = world;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_out, hipEventDisableTiming) != hipSuccess) {
        psm_dist_destroy(d);
        return set_err(ctx, PSM_ERR_HIP, "psm_dist_init: stream / events");
    }
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
        d->comm = nullptr;
        psm_dist_destroy(d);
        return nccl_err(ctx, r, "ncclCommInitRank");
    }
    *out = d;
    return PSM_OK;
}

int psm_dist_rank(const psm_dist* d) { return d ? d->rank : -1; }
int psm_dist_world(const psm_dist* d) { return d ? d->world : -1; }

// One-GPU rehearsal of another rank's share (bench.py --emulate-tile R/W --force-dist): the communicator keeps its real
// ranks, but gathers pack the tile (tile_rank, tile_world) and skip the unpack / let the caller skip the fold, i.e. they
// cost what a WORKER rank of a tile_world-GPU run pays per frame. The image is then not a complete frame.
int psm_dist_emulate_tile(psm_dist* d, int tile_rank, int tile_world) {
    if (!d || tile_world < 1 || tile_rank < 0 || tile_rank >= tile_world) return PSM_ERR_INVALID;
    d->tile_rank = tile_rank; d->tile_world = tile_world;
    return PSM_OK;
}

int psm_dist_gather_tiles(psm_dist* d, psm_rt* rt) {
    if (!d || !rt || !rt->t_sum) return PSM_ERR_INVALID;
    psm_ctx* c = rt->ctx;
    if (c->device != d->ctx->device) return set_err(c, PSM_ERR_INVALID, "psm_dist_gather_tiles: Pipeline and communicator live on different devices");
    const bool emulated = d->tile_world != d->world || d->tile_rank != d->rank;
    if (rt->tile_mode != 1 || (int)rt->tile_world != d->tile_world || (int)rt->tile_rank != d->tile_rank)
        return set_err(c, PSM_ERR_STATE, "psm_dist_gather_tiles: the Pipeline's tile is not psm_rt_set_tile_interleaved(rank, world) of this communicator");
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)interleaved_texels(0, (uint32_t)d->tile_world, rt->w, rt->h) * 4;  // rank 0 owns the most bands
    if (per != d->per_floats) {
        PSM_HIP(c, hipStreamSynchronize(d->stream));
        if (d->d_send) (void)hipFree(d->d_send);
        if (d->d_recv) (void)hipFree(d->d_recv);
        d->d_send = d->d_recv = nullptr;
        d->per_floats = 0;
        PSM_HIP(c, hipMalloc((void**)&d->d_send, per * sizeof(float)));
        PSM_HIP(c, hipMemsetAsync(d->d_send, 0, per * sizeof(float), d->stream));
        if (d->rank == 0) PSM_HIP(c, hipMalloc((void**)&d->d_recv, per * sizeof(float) * (size_t)d->world));
        d->per_floats = per;
    }
    // the communicator's stream takes over from the Pipeline's stream ...
    PSM_HIP(c, hipEventRecord(d->ev_in, c->stream));
    PSM_HIP(c, hipStreamWaitEvent(d->stream, d->ev_in, 0));
    hipStream_t keep = c->stream;
    c->stream = d->stream;  // launch_rt_pack launches on the context's stream
    int rc = launch_rt_pack(rt, d->d_send, 0, 1u, (uint32_t)d->tile_rank, (uint32_t)d->tile_world);
    if (rc == PSM_OK) {
        ncclResult_t r = ncclGather(d->d_send, d->d_recv, per, ncclFloat, 0, d->comm, d->stream);
        if (r != ncclSuccess) rc = nccl_err(c, r, "ncclGather");
    }
    if (rc == PSM_OK && d->rank == 0 && !emulated && d->world > 1)   // every other rank's tile into the image, one launch
        rc = launch_rt_unpack_all(rt, d->d_recv, (uint32_t)d->world, 0u, per);
    c->stream = keep;
    if (rc != PSM_OK) return rc;
    // ... and hands back: whatever the Pipeline's stream does next (sample(), the next camera()) sees the gathered image
    PSM_HIP(c, hipEventRecord(d->ev_out, d->stream));
    PSM_HIP(c, hipStreamWaitEvent(c->stream, d->ev_out, 0));
    return PSM_OK;
}

int psm_dist_allgather_i32(psm_dist* d, const int32_t* send, int32_t* recv, uint32_t n) {
    if (!d || !send || !recv || n == 0) return PSM_ERR_INVALID;
    psm_ctx* c = d->ctx;
    (void)hipSetDevice(c->device);
    const size_t need = (size_t)n * (size_t)(d->world + 1);
    if (need > d->i32_cap) {
        PSM_HIP(c, hipStreamSynchronize(d->stream));
        if (d->d_i32) (void)hipFree(d->d_i32);
        d->d_i32 = nullptr; d->i32_cap = 0;
        PSM_HIP(c, hipMalloc((void**)&d->d_i32, need * sizeof(int32_t)));
        d->i32_cap = need;
    }
    PSM_HIP(c, hipMemcpyAsync(d->d_i32, send, n * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    PSM_NCCL(c, ncclAllGather(d->d_i32, d->d_i32 + n, n, ncclInt32, d->comm, d->stream));
    PSM_HIP(c, hipMemcpyAsync(recv, d->d_i32 + n, (size_t)n * d->world * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
    PSM_HIP(c, hipStreamSynchronize(d->stream));
    return PSM_OK;
}

// The global `fewer than 32 rays -> stop` rule (Pipeline.inl:459-461) from every rank's (rounds done, local rays
// waiting) per lane. all: [world][2][lanes] (rounds, then counts). A rank behind the furthest one catches up (any round
// below the furthest rank's had >= 32 rays there alone); with every rank at the same round the global count decides.
int psm_dist_decide(uint32_t world, uint32_t lanes, const int32_t* all, uint32_t depth, int32_t* over, uint32_t* force_until) {
    if (!all || !over || !force_until || world == 0) return PSM_ERR_INVALID;
    for (uint32_t s = 0; s < lanes; s++) {
        int32_t pmax = INT32_MIN, pmin = INT32_MAX;
        int64_t total = 0;
        for (uint32_t r = 0; r < world; r++) {
            const int32_t rd = all[((size_t)r * 2 + 0) * lanes + s];
            pmax = rd > pmax ? rd : pmax;
            pmin = rd < pmin ? rd : pmin;
            total += all[((size_t)r * 2 + 1) * lanes + s];
        }
        if (pmin < pmax) { over[s] = 0; force_until[s] = (uint32_t)pmax; }
        else if ((uint32_t)pmax >= depth || total < 32) { over[s] = 1; force_until[s] = (uint32_t)pmax; }
        else { over[s] = 0; force_until[s] = (uint32_t)pmax + 1u; }
    }
    return PSM_OK;
}

// `lanes` tile-sharded frames in flight on this rank, start to finish (every rank calls it with the same arguments but
// its own objects): the lanes run free until their LOCAL counts park them (psm_lanes_run_sharded), one small
// all-gather per decision applies the stop rule to each frame's GLOBAL count, and every frame ends with the path's one
// data-path collective, the tile gather to rank 0, which folds it into `fold_into` in frame order.
int psm_dist_render_batch(psm_dist* d, psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                          const float proj_inv[16], const uint32_t* frame_seeds, uint32_t depth, int rebuild, const double* opt,
                          psm_rt* fold_into, uint32_t* rounds_out) {
    if (!d || !rts || !bvhs || !frame_seeds || lanes == 0 || lanes > 64) return PSM_ERR_INVALID;
    if (d->rank == 0 && !fold_into) return PSM_ERR_INVALID;
    std::vector<uint32_t> state(frame_seeds, frame_seeds + lanes), rounds(lanes, 0u), force(lanes, 0u);
    std::vector<int32_t> counts(lanes, 0), over(lanes, 0), verdict(lanes, 0);
    std::vector<int32_t> mine(2 * (size_t)lanes), all(2 * (size_t)lanes * (size_t)d->world);
    int rc = psm_lanes_run_sharded(rts, bvhs, lanes, cam_inv, proj_inv, state.data(), rounds.data(), force.data(), depth, 1, rebuild, opt,
                                   counts.data());
    while (rc == PSM_OK) {
        for (uint32_t s = 0; s < lanes; s++) { mine[s] = (int32_t)rounds[s]; mine[lanes + s] = counts[s]; }
        rc = psm_dist_allgather_i32(d, mine.data(), all.data(), 2 * lanes);
        if (rc != PSM_OK) break;
        rc = psm_dist_decide((uint32_t)d->world, lanes, all.data(), depth, verdict.data(), force.data());
        if (rc != PSM_OK) break;
        bool done = true;
        for (uint32_t s = 0; s < lanes; s++) {
            over[s] = over[s] | verdict[s];
            if (over[s]) force[s] = rounds[s];  // a finished frame's lane stays parked
            done = done && over[s];
        }
        if (done) break;
        rc = psm_lanes_run_sharded(rts, bvhs, lanes, cam_inv, proj_inv, state.data(), rounds.data(), force.data(), depth, 0, rebuild, opt,
                                   counts.data());
    }
    for (uint32_t s = 0; s < lanes && rc == PSM_OK; s++) {  // frame order
        rc = psm_dist_gather_tiles(d, rts[s]);
        if (rc == PSM_OK && d->rank == 0) rc = psm_rt_sample_from(fold_into, rts[s]);
    }
    if (rounds_out) for (uint32_t s = 0; s < lanes; s++) rounds_out[s] = rounds[s];
    return rc;
}

int psm_dist_barrier(psm_dist* d) {
    if (!d) return PSM_ERR_INVALID;
    int32_t one = 1;
    std::vector<int32_t> all((size_t)d->world);
    return psm_dist_allgather_i32(d, &one, all.data(), 1);
}

}  // extern "C"
