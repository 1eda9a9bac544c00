// dist.hip -- tile-sharded frames across the GPUs of a node: the collectives behind the C ABI (psm_dist_*).
//
// The reference has no multi-GPU path. A frame shards by screen tile: 8-row bands dealt to the ranks
// (psm_rt_set_tile_interleaved), the hierarchy is rebuilt redundantly on every GPU, and the path has exactly ONE
// data-path exchange per frame: the gather of per-texel radiance (16 B per texel) to rank 0, which runs the sampler.
// RCCL over xGMI, one process per GPU; every peer sends on its own direct link, rank 0 ingests 7 links in parallel.
// The only other exchange is a few ints per batch of frames for the reference's `fewer than 32 rays -> stop` rule on
// the frame's global ray count (psm_lanes_run_sharded), also provided here so a C++ host needs nothing but this ABI.
//
// Transport seam: both exchanges go through a table of two functions (psm_dist_transport). RCCL is what
// psm_dist_init / psm_dist_connect install; transport_shm.hip is a host-staged one for several processes that share
// one GPU (RCCL refuses two ranks on a device), which is how the sharded scheduler is run against real peers on the
// one-GPU test boxes (tests/test_gpu_dist.py).
//
// Streams: all collectives of a communicator are issued on ITS stream in call order (every rank makes the same calls
// in the same order); a gather waits for the Pipeline's stream (event), packs, gathers, unpacks on rank 0, and the
// Pipeline's stream waits for that (event), so folding the frame afterwards needs no host synchronisation.
//
// Failure: a rank whose own work fails keeps making the collective calls of the sequence (poisoned exchange values,
// empty tiles) until the next exchange tells everybody, so one rank's error is an error return on every rank, not a
// node-wide wait inside ncclGather (psm_dist_decide, psm_dist_render_batch, psm_dist_render_frames in lanes.hip).
#include <rccl/rccl.h>

#include <climits>
#include <cstring>
#include <new>
#include <vector>

#include "psm_internal.h"

namespace psm {

// ---- the RCCL transport ---------------------------------------------------------------------------------------------
struct RcclTransport {
    ncclComm_t comm = nullptr;
    std::string err;
};
static int rccl_fail(RcclTransport* t, ncclResult_t r, const char* what) {
    t->err = std::string(what) + ": " + ncclGetErrorString(r);
    return PSM_ERR_HIP;
}
static int rccl_gather_f32(void* u, const float* d_send, float* d_recv, size_t count, int root, void* stream) {
    RcclTransport* t = (RcclTransport*)u;
    ncclResult_t r = ncclGather(d_send, d_recv, count, ncclFloat, root, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? PSM_OK : rccl_fail(t, r, "ncclGather");
}
static int rccl_allgather_i32(void* u, const int32_t* d_send, int32_t* d_recv, size_t n, void* stream) {
    RcclTransport* t = (RcclTransport*)u;
    ncclResult_t r = ncclAllGather(d_send, d_recv, n, ncclInt32, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? PSM_OK : rccl_fail(t, r, "ncclAllGather");
}
static void rccl_destroy(void* u) {
    RcclTransport* t = (RcclTransport*)u;
    if (t->comm) (void)ncclCommDestroy(t->comm);
    delete t;
}
static const char* rccl_last_error(void* u) { return ((RcclTransport*)u)->err.c_str(); }

static int transport_err(psm_dist* d, int rc, const char* what) {
    const char* detail = (d->connected && d->tr.last_error) ? d->tr.last_error(d->tr.user) : nullptr;
    d->ctx->err = std::string(what) + " (transport " + (d->tr.name ? d->tr.name : "?") + ")" + (detail && *detail ? std::string(": ") + detail : std::string());
    return rc == PSM_ERR_PEER ? PSM_ERR_PEER : PSM_ERR_HIP;
}

// tile buffers of the gather for images of w x h texels
int dist_reserve(psm_dist* d, uint32_t w, uint32_t h) {
    psm_ctx* c = d->ctx;
    const size_t per = (size_t)max_owned_texels(d->bands, w, h) * 4;  // every rank sends as much as the largest tile
    if (per == d->per_floats) return PSM_OK;
    PSM_HIP(c, hipStreamSynchronize(d->stream));
    if (d->d_send) (void)hipFree(d->d_send);
    if (d->d_recv) (void)hipFree(d->d_recv);
    d->d_send = d->d_recv = nullptr;
    d->per_floats = 0;
    if (per == 0) return PSM_OK;
    PSM_HIP(c, hipMalloc((void**)&d->d_send, per * sizeof(float)));
    PSM_HIP(c, hipMemsetAsync(d->d_send, 0, per * sizeof(float), d->stream));
    if (d->rank == 0) PSM_HIP(c, hipMalloc((void**)&d->d_recv, per * sizeof(float) * (size_t)d->world));
    d->per_floats = per;
    return PSM_OK;
}

// this rank's place in a gather whose tile it cannot deliver (its frame failed): the collective call is made all the
// same, with whatever the send buffer holds, so that the other ranks' gather completes
int dist_gather_placeholder(psm_dist* d) {
    if (!d || !d->connected) return PSM_ERR_INVALID;
    if (!d->d_send || d->per_floats == 0) return set_err(d->ctx, PSM_ERR_STATE, "psm_dist: no tile buffer to keep the gather sequence with");
    int rc = d->tr.gather_f32(d->tr.user, d->d_send, d->d_recv, d->per_floats, 0, (void*)d->stream);
    return rc == PSM_OK ? PSM_OK : transport_err(d, rc, "gather (placeholder)");
}

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
    if (d->connected && d->tr.destroy) d->tr.destroy(d->tr.user);
    if (d->d_send) (void)hipFree(d->d_send);
    if (d->d_recv) (void)hipFree(d->d_recv);
    if (d->d_i32) (void)hipFree(d->d_i32);
    if (d->ev_in) (void)hipEventDestroy(d->ev_in);
    if (d->ev_out) (void)hipEventDestroy(d->ev_out);
    if (d->ev_x) (void)hipEventDestroy(d->ev_x);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    return PSM_OK;
}

int psm_dist_prepare(psm_ctx* ctx, int rank, int world, psm_dist** out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return PSM_ERR_INVALID;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    psm_dist* d = new (std::nothrow) psm_dist();
    if (!d) return PSM_ERR_INVALID;
    d->ctx = ctx; d->rank = rank; d->world = world;
    d->tile_rank = rank; d->tile_world = world;
    (void)band_map_make((uint32_t)world, nullptr, &d->bands);
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_out, hipEventDisableTiming) != hipSuccess) {
        psm_dist_destroy(d);
        return set_err(ctx, PSM_ERR_HIP, "psm_dist_prepare: stream / events");
    }
    *out = d;
    return PSM_OK;
}

int psm_dist_connect_transport(psm_dist* d, const psm_dist_transport* t) {
    if (!d || !t || !t->gather_f32 || !t->allgather_i32) return PSM_ERR_INVALID;
    if (d->connected) return set_err(d->ctx, PSM_ERR_STATE, "psm_dist_connect: the communicator already has a transport");
    d->tr = *t;
    d->connected = true;
    d->comm_ranks = d->world;   // (a caller's own transport: taken at its word; RCCL and the host-staged one overwrite it with what they count)
    return PSM_OK;
}

int psm_dist_connect(psm_dist* d, const uint8_t id[128]) {
    if (!d || !id) return PSM_ERR_INVALID;
    if (d->connected) return set_err(d->ctx, PSM_ERR_STATE, "psm_dist_connect: the communicator already has a transport");
    (void)hipSetDevice(d->ctx->device);
    RcclTransport* t = new (std::nothrow) RcclTransport();
    if (!t) return PSM_ERR_INVALID;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&t->comm, d->world, u, d->rank);
    if (r != ncclSuccess) {
        d->ctx->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        delete t;
        return PSM_ERR_HIP;
    }
    // what RCCL itself says about the communicator: its size and this process's rank in it (a launcher that handed two
    // processes the same rank, or a world the id was not made for, must not pass for an 8-GPU run)
    int cnt = -1, ur = -1;
    if (ncclCommCount(t->comm, &cnt) != ncclSuccess || ncclCommUserRank(t->comm, &ur) != ncclSuccess || cnt != d->world || ur != d->rank) {
        d->ctx->err = "psm_dist_connect: the RCCL communicator reports " + std::to_string(cnt) + " ranks, this one as " + std::to_string(ur) +
                      " (asked for rank " + std::to_string(d->rank) + " of " + std::to_string(d->world) + ")";
        rccl_destroy(t);
        return PSM_ERR_STATE;
    }
    psm_dist_transport tr = {t, rccl_gather_f32, rccl_allgather_i32, rccl_destroy, rccl_last_error, "rccl"};
    int rc = psm_dist_connect_transport(d, &tr);
    if (rc == PSM_OK) d->comm_ranks = cnt;
    return rc;
}

int psm_dist_init(psm_ctx* ctx, int rank, int world, const uint8_t id[128], psm_dist** out) {
    if (!id || !out) return PSM_ERR_INVALID;
    int rc = psm_dist_prepare(ctx, rank, world, out);
    if (rc != PSM_OK) return rc;
    rc = psm_dist_connect(*out, id);
    if (rc != PSM_OK) { psm_dist_destroy(*out); *out = nullptr; }
    return rc;
}

int psm_dist_rank(const psm_dist* d) { return d ? d->rank : -1; }
int psm_dist_world(const psm_dist* d) { return d ? d->world : -1; }
int psm_dist_comm_ranks(const psm_dist* d) { return d ? d->comm_ranks : -1; }
const char* psm_dist_transport_name(const psm_dist* d) { return (d && d->connected) ? d->tr.name : nullptr; }

// One-GPU rehearsal of another rank's share (bench.py --emulate-tile R/W --force-dist): the communicator keeps its real
// ranks, but gathers pack the tile (tile_rank, tile_world) and skip the unpack / let the caller skip the fold, i.e. they
// cost what a WORKER rank of a tile_world-GPU run pays per frame. The image is then not a complete frame.
int psm_dist_emulate_tile(psm_dist* d, int tile_rank, int tile_world) {
    if (!d || tile_world < 1 || tile_rank < 0 || tile_rank >= tile_world) return PSM_ERR_INVALID;
    d->tile_rank = tile_rank; d->tile_world = tile_world;
    if (!band_map_make((uint32_t)tile_world, nullptr, &d->bands)) return PSM_ERR_INVALID;
    return PSM_OK;
}

// The dealing of the bands the gathers use (every rank passes the same weights, and gives its Pipelines the same ones
// with psm_rt_set_tile_weighted): weights[tile_world] bands per period for each rank, NULL = one each (round-robin).
int psm_dist_set_band_weights(psm_dist* d, const uint32_t* weights) {
    if (!d) return PSM_ERR_INVALID;
    BandMap m;
    if (!band_map_make((uint32_t)d->tile_world, weights, &m))
        return set_err(d->ctx, PSM_ERR_INVALID, "psm_dist_set_band_weights: the weights must add up to 1..64");
    d->bands = m;
    return PSM_OK;
}

}  // extern "C"

namespace psm {
// psm_dist_gather_tiles; *collective = INT_MIN while the transport's gather has not been called, its result afterwards
// (a caller that must keep the collective sequence makes the call itself in the first case: dist_gather_placeholder)
int dist_gather_tiles(psm_dist* d, psm_rt* rt, int* collective, bool* defer_wait = nullptr) {
    *collective = INT_MIN;
    if (!d || !rt || !rt->t_sum) return PSM_ERR_INVALID;
    psm_ctx* c = rt->ctx;
    if (!d->connected) return set_err(c, PSM_ERR_STATE, "psm_dist_gather_tiles: the communicator has no transport (psm_dist_connect*)");
    if (c->device != d->ctx->device) return set_err(c, PSM_ERR_INVALID, "psm_dist_gather_tiles: Pipeline and communicator live on different devices");
    const bool emulated = d->tile_world != d->world || d->tile_rank != d->rank;
    if (rt->tile_mode != 1 || (int)rt->tile_world != d->tile_world || (int)rt->tile_rank != d->tile_rank || !band_map_equal(rt->bands, d->bands))
        return set_err(c, PSM_ERR_STATE, "psm_dist_gather_tiles: the Pipeline's tile is not this communicator's (psm_rt_set_tile_interleaved / _weighted with its rank, world and band weights)");
    (void)hipSetDevice(c->device);
    int rc = dist_reserve(d, rt->w, rt->h);
    if (rc != PSM_OK) { if (c != d->ctx) c->err = d->ctx->err; return rc; }
    const size_t per = d->per_floats;
    if (per == 0) { *collective = PSM_OK; return PSM_OK; }  // an image without texels: no rank has anything to send
    // the communicator's stream takes over from the Pipeline's stream ...
    PSM_HIP(c, hipEventRecord(d->ev_in, c->stream));
    PSM_HIP(c, hipStreamWaitEvent(d->stream, d->ev_in, 0));
    rc = launch_rt_pack(rt, d->stream, d->d_send, 0, &d->bands, (uint32_t)d->tile_rank, (uint32_t)d->tile_world);
    // the collective call is made even when the pack could not be launched: the peers' gather must complete
    *collective = d->tr.gather_f32(d->tr.user, d->d_send, d->d_recv, per, 0, (void*)d->stream);
    if (*collective != PSM_OK) { int e = transport_err(d, *collective, "gather"); if (c != d->ctx) c->err = d->ctx->err; return e; }
    if (rc == PSM_OK && d->rank == 0 && !emulated && d->world > 1)   // every other rank's tile into the image, one launch
        rc = launch_rt_unpack_all(rt, d->stream, d->d_recv, d->bands, 0u, per);
    if (rc != PSM_OK) return rc;
    // ... and hands back: whatever the Pipeline's stream does next (sample(), the next camera()) sees the gathered image
    PSM_HIP(c, hipEventRecord(d->ev_out, d->stream));
    if (defer_wait) *defer_wait = true;   // the caller enqueues the wait (a later record of ev_out covers this one: one stream)
    else PSM_HIP(c, hipStreamWaitEvent(c->stream, d->ev_out, 0));
    return PSM_OK;
}

// one frame's gather (+ fold on rank 0) inside a sharded batch; `local` is this rank's own status so far. Returns
// false when the transport itself failed (the sequence cannot be kept: the caller returns `local` at once).
bool dist_frame_gather(psm_dist* d, psm_rt* rt, psm_rt* fold_into, int& local, uint32_t* defer) {
    int coll = INT_MIN;
    if (local == PSM_OK) {
        bool wait_gather = false, wait_fold = false;
        local = dist_gather_tiles(d, rt, &coll, defer ? &wait_gather : nullptr);
        if (local == PSM_OK && d->rank == 0) {
            // (the Pipeline's stream has not waited for the gather: the accumulating Pipeline's stream does, before it samples)
            if (wait_gather && hipStreamWaitEvent(fold_into->ctx->stream, d->ev_out, 0) != hipSuccess)
                local = set_err(rt->ctx, PSM_ERR_HIP, "hipStreamWaitEvent (fold after gather)", hipGetLastError());
            if (local == PSM_OK) local = defer ? rt_fold(fold_into, rt, &wait_fold) : psm_rt_sample_from(fold_into, rt);
        }
        if (defer) *defer |= (wait_gather ? LANE_WAIT_GATHER : 0u) | (wait_fold ? LANE_WAIT_FOLD : 0u);
    }
    if (coll == INT_MIN) {  // this rank did not get as far as the collective call: make it, the others are in it
        if (dist_gather_placeholder(d) != PSM_OK) return false;
        coll = PSM_OK;
    }
    return coll == PSM_OK;
}

// the waits a deferred gather / fold left to the caller, on the Pipeline's stream
int lane_flush_waits(psm_dist* d, psm_rt* rt, uint32_t* pending) {
    psm_ctx* c = rt->ctx;
    if (*pending & LANE_WAIT_GATHER) PSM_HIP(c, hipStreamWaitEvent(c->stream, d->ev_out, 0));
    if (*pending & LANE_WAIT_FOLD) PSM_HIP(c, hipStreamWaitEvent(c->stream, rt->ev_fold, 0));
    *pending = 0;
    return PSM_OK;
}
}  // namespace psm

extern "C" {

int psm_dist_gather_tiles(psm_dist* d, psm_rt* rt) {
    int coll;
    return dist_gather_tiles(d, rt, &coll);
}

int psm_dist_allgather_i32(psm_dist* d, const int32_t* send, int32_t* recv, uint32_t n) {
    if (!d || !send || !recv || n == 0) return PSM_ERR_INVALID;
    psm_ctx* c = d->ctx;
    if (!d->connected) return set_err(c, PSM_ERR_STATE, "psm_dist_allgather_i32: the communicator has no transport (psm_dist_connect*)");
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
    int trc = d->tr.allgather_i32(d->tr.user, d->d_i32, d->d_i32 + n, n, (void*)d->stream);
    if (trc != PSM_OK) return transport_err(d, trc, "allgather");
    PSM_HIP(c, hipMemcpyAsync(recv, d->d_i32 + n, (size_t)n * d->world * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
    if (d->idle_fn) {   // the peers may be a while: the caller's other lanes are served meanwhile (psm_dist_render_frames)
        if (!d->ev_x) PSM_HIP(c, hipEventCreateWithFlags(&d->ev_x, hipEventDisableTiming));
        PSM_HIP(c, hipEventRecord(d->ev_x, d->stream));
        for (;;) {
            const hipError_t q = hipEventQuery(d->ev_x);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return set_err(c, PSM_ERR_HIP, "psm_dist_allgather_i32: hipEventQuery", q);
            d->idle_fn(d->idle_user);
        }
    }
    PSM_HIP(c, hipStreamSynchronize(d->stream));
    return PSM_OK;
}

// The global `fewer than 32 rays -> stop` rule (Pipeline.inl:459-461) from every rank's (rounds done, local rays
// waiting) per lane. all: [world][2][lanes] (rounds, then counts). A rank behind the furthest one catches up (any round
// below the furthest rank's had >= 32 rays there alone); with every rank at the same round the global count decides.
// A negative round is a rank's report that it failed: everybody gets PSM_ERR_PEER from the same exchange.
int psm_dist_decide(uint32_t world, uint32_t lanes, const int32_t* all, uint32_t depth, int32_t* over, uint32_t* force_until) {
    if (!all || !over || !force_until || world == 0) return PSM_ERR_INVALID;
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t s = 0; s < lanes; s++)
            if (all[((size_t)r * 2 + 0) * lanes + s] < 0) return PSM_ERR_PEER;
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

// every rank's own status (PSM_OK or its error) -> the call's return code, the same kind on every rank: the last
// exchange of psm_dist_render_batch / psm_dist_render_frames
int psm_dist_agree(psm_dist* d, int local_rc) {
    if (!d) return PSM_ERR_INVALID;
    const std::string keep = d->ctx->err;
    int32_t mine = local_rc == PSM_OK ? 0 : -1;
    std::vector<int32_t> all((size_t)d->world, 0);
    int rc = psm_dist_allgather_i32(d, &mine, all.data(), 1);
    if (rc != PSM_OK) return local_rc != PSM_OK ? local_rc : rc;
    if (local_rc != PSM_OK) { d->ctx->err = keep; return local_rc; }
    for (int r = 0; r < d->world; r++)
        if (all[(size_t)r] < 0) {
            d->ctx->err = "psm_dist: rank " + std::to_string(r) + " reported a failure";
            return PSM_ERR_PEER;
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
    for (uint32_t s = 0; s < lanes; s++)
        if (!rts[s] || !bvhs[s]) return PSM_ERR_INVALID;
    std::vector<uint32_t> state(frame_seeds, frame_seeds + lanes), rounds(lanes, 0u), force(lanes, 0u);
    std::vector<int32_t> counts(lanes, 0), over(lanes, 0), verdict(lanes, 0);
    std::vector<int32_t> mine(2 * (size_t)lanes), all(2 * (size_t)lanes * (size_t)d->world);
    // `local` is this rank's own failure: it keeps taking part in the exchanges (rounds = -1) until everybody knows
    int local = dist_reserve(d, rts[0]->w, rts[0]->h);
    if (local == PSM_OK)
        local = psm_lanes_run_sharded(rts, bvhs, lanes, cam_inv, proj_inv, state.data(), rounds.data(), force.data(), depth, 1, rebuild, opt,
                                      counts.data());
    int rc = PSM_OK;  // the call's fate as all ranks see it
    for (;;) {
        for (uint32_t s = 0; s < lanes; s++) { mine[s] = local == PSM_OK ? (int32_t)rounds[s] : -1; mine[lanes + s] = counts[s]; }
        const std::string keep = d->ctx->err;
        rc = psm_dist_allgather_i32(d, mine.data(), all.data(), 2 * lanes);
        if (rc != PSM_OK) return local != PSM_OK ? local : rc;  // the transport itself failed: nothing left to keep in step
        rc = psm_dist_decide((uint32_t)d->world, lanes, all.data(), depth, verdict.data(), force.data());
        if (rc != PSM_OK) { if (local != PSM_OK) d->ctx->err = keep; break; }
        bool done = true;
        for (uint32_t s = 0; s < lanes; s++) {
            over[s] = over[s] | verdict[s];
            if (over[s]) force[s] = rounds[s];  // a finished frame's lane stays parked
            done = done && over[s];
        }
        if (done) break;
        local = psm_lanes_run_sharded(rts, bvhs, lanes, cam_inv, proj_inv, state.data(), rounds.data(), force.data(), depth, 0, rebuild, opt,
                                      counts.data());
    }
    if (rounds_out) for (uint32_t s = 0; s < lanes; s++) rounds_out[s] = rounds[s];
    if (rc != PSM_OK) return local != PSM_OK ? local : rc;  // every rank leaves here together: no gathers
    for (uint32_t s = 0; s < lanes; s++)  // frame order; a rank that fails now still takes part in the remaining gathers
        if (!dist_frame_gather(d, rts[s], fold_into, local)) return local;
    return psm_dist_agree(d, local);
}

int psm_dist_barrier(psm_dist* d) {
    if (!d) return PSM_ERR_INVALID;
    int32_t one = 1;
    std::vector<int32_t> all((size_t)d->world);
    return psm_dist_allgather_i32(d, &one, all.data(), 1);
}

}  // extern "C"
