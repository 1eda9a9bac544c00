// lanes.hip -- several frames in flight on one GPU.
//
// A bounce round ends with the slowest ray of its traversal launch (a wave running alone takes ~1 us per
// BVH step), so a single frame leaves most of the chip idle for a good part of every
// round. The frames of one accumulation (the 4 spp of BASELINE's metric) are independent until the sampler
// folds them together, so this scheduler keeps `lanes` of them in flight, each on its own context / HIP
// stream with its own hierarchy and ray buffers: while one frame's traversal drains, the other frames'
// kernels fill the machine. Every frame is exactly GltfViewer::process() (Viewer.cpp:296-312): build,
// camera, <= depth x (getRayCount >= 32 ? intersection, shade), sample -- with its own rand() stream, and
// with sample() issued in frame order on the accumulating Pipeline (psm_rt_sample_from), which makes the
// image identical to rendering the same frames one after another.
//
// Host side only: no kernel lives here. The scheduler is event driven: after each shade it queues an
// asynchronous read-back of the lane's next ray count and polls the lanes' events, so no lane waits for
// another one's round; a lane that finishes a frame takes the next one as soon as its frame is folded.
#include "psm_internal.h"

#include <algorithm>

#include <chrono>
#include <cstring>
#include <thread>

namespace psm {

static inline uint32_t lcg_next(uint32_t& state) {  // the CRT rand() stand-in of DESIGN.md 2.1
    state = state * 214013u + 2531011u;
    return (state >> 16) & 0x7fffu;
}

static int lane_resources(psm_rt* r) {
    psm_ctx* c = r->ctx;
    // (coherent: rt_scan_blocks writes the next ray count into it from the device, the host reads it after the event)
    if (!r->h_cnt) PSM_HIP(c, hipHostMalloc((void**)&r->h_cnt, sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
    if (!r->ev_cnt) PSM_HIP(c, hipEventCreateWithFlags(&r->ev_cnt, hipEventDisableTiming));
    if (!r->ev_fold) PSM_HIP(c, hipEventCreateWithFlags(&r->ev_fold, hipEventDisableTiming));
    return PSM_OK;
}

enum LaneState { IDLE, RUNNING, FINISHED };

struct Lane {
    psm_rt* rt = nullptr;
    psm_bvh* bvh = nullptr;
    LaneState state = IDLE;
    int frame = -1;
    uint32_t rand = 0;
    uint32_t round = 0;
    uint64_t rays = 0;
    bool fold_wait = false;   // the lane's stream has yet to wait for the fold of its last frame (see fold)
    int next = -1;            // the frame the lane claimed when its last one ended, -1: none
    bool marked = false;      // ... and its rebuild is queued behind a mark of the old frame's end (ev_fold)
};

// sample() of `r` fed with src's frame; stream-ordered against both contexts. With `defer` the caller takes over the last
// step -- src's stream waiting for the sample before anything touches src's texels again -- and may put work that does not
// touch them (the next frame's rebuild) in front of the wait: hipStreamWaitEvent(src stream, src->ev_fold).
// `recorded`: src->ev_fold already marks the end of src's frame on src's stream (the caller recorded it before it queued
// further work there that the sample need not wait for).
static int fold(psm_rt* r, psm_rt* src, bool* defer = nullptr, bool recorded = false) {
    psm_ctx* c = r->ctx;
    if (r == src || c->stream == src->ctx->stream) return launch_rt_sample(r, src);
    int rc = lane_resources(src);
    if (rc != PSM_OK) return rc;
    if (!recorded) PSM_HIP(c, hipEventRecord(src->ev_fold, src->ctx->stream));
    PSM_HIP(c, hipStreamWaitEvent(c->stream, src->ev_fold, 0));
    rc = launch_rt_sample(r, src);
    if (rc != PSM_OK) return rc;
    PSM_HIP(c, hipEventRecord(src->ev_fold, c->stream));
    if (defer) *defer = true;
    else PSM_HIP(c, hipStreamWaitEvent(src->ctx->stream, src->ev_fold, 0));  // src's next camera() waits for the fold
    return PSM_OK;
}

int rt_fold(psm_rt* r, psm_rt* src, bool* defer) { return fold(r, src, defer); }

// The one step every scheduler of this file is made of. A bounce round of a lane's frame, queued on the lane's stream:
// intersection + shade (whose segment scan writes the next ray count straight into the lane's pinned slot) + the event that
// marks the count's arrival ...
static int lane_queue_round(psm_rt* r, psm_bvh* b, uint32_t time) {
    int e = psm_rt_traverse(r, b);
    if (e != PSM_OK) return e;
    e = psm_rt_shade(r, b, time);
    if (e != PSM_OK) return e;
    PSM_HIP(r->ctx, hipEventRecord(r->ev_cnt, r->ctx->stream));
    return PSM_OK;
}
// ... and one non-blocking look at it: 1 = the round has ended and r->ray_count is what reloadQueuedRays would have learnt
// (Pipeline.inl:325-359), 0 = still in flight, < 0 = error.
static int lane_poll_round(psm_rt* r) {
    hipError_t q = hipEventQuery(r->ev_cnt);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return set_err(r->ctx, PSM_ERR_HIP, "hipEventQuery", q);
    r->ray_count = *r->h_cnt;
    r->count_valid = true;
    return 1;
}

}  // namespace psm

using namespace psm;

extern "C" int psm_rt_sample_from(psm_rt* r, psm_rt* src) {
    if (!r || !src || !r->presampled || !src->t_sum) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    if (r->w != src->w || r->h != src->h) return set_err(r->ctx, PSM_ERR_INVALID, "psm_rt_sample_from: ray grids differ");
    return fold(r, src);
}

extern "C" int psm_lanes_render(psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                                const float proj_inv[16], const uint32_t* frame_seeds, uint32_t frames, uint32_t depth,
                                int rebuild, const double* opt, psm_rt* fold_into, psm_lane_result* results) {
    if (!rts || !bvhs || !cam_inv || !proj_inv || lanes == 0 || lanes > 64 || (frames && !frame_seeds)) return PSM_ERR_INVALID;
    for (uint32_t s = 0; s < lanes; s++) {
        if (!rts[s] || !bvhs[s]) return PSM_ERR_INVALID;
        for (uint32_t q = 0; q < s; q++)
            if (rts[q] == rts[s] || rts[q]->ctx->stream == rts[s]->ctx->stream)
                return set_err(rts[s]->ctx, PSM_ERR_INVALID, "psm_lanes_render: every lane needs its own context (stream) and ray buffers");
        if (rts[s]->ctx != bvhs[s]->ctx)
            return set_err(rts[s]->ctx, PSM_ERR_INVALID, "psm_lanes_render: a lane's hierarchy must live on the lane's context");
        if (fold_into && (fold_into->w != rts[s]->w || fold_into->h != rts[s]->h))
            return set_err(rts[s]->ctx, PSM_ERR_INVALID, "psm_lanes_render: fold_into and the lanes differ in ray-grid size");
    }
    if (!fold_into && frames > lanes)
        return set_err(rts[0]->ctx, PSM_ERR_INVALID, "psm_lanes_render: more frames than lanes need fold_into (a lane's texel sums are overwritten by its next frame)");
    (void)hipSetDevice(rts[0]->ctx->device);
    std::vector<Lane> L(lanes);
    for (uint32_t s = 0; s < lanes; s++) {
        L[s].rt = rts[s];
        L[s].bvh = bvhs[s];
        int e = lane_resources(rts[s]);
        if (e != PSM_OK) return e;
    }
    for (uint32_t s = 0; s < lanes; s++) rts[s]->in_flight = lanes;
    int rc = PSM_OK;
    auto queue_round = [&](Lane& ln) -> int {  // intersection + shade + asynchronous read-back of the next count
        psm_rt* r = ln.rt;
        ln.rays += r->ray_count;
        int e = lane_queue_round(r, ln.bvh, lcg_next(ln.rand));
        if (e != PSM_OK) return e;
        ln.round++;
        return PSM_OK;
    };
    uint32_t next_frame = 0;
    // A frame has ended: its lane waits for its turn to fold (frame order). The lane's NEXT frame is claimed here and its
    // rebuild queued at once -- it touches nothing the fold reads -- so a lane that ends out of turn does not sit idle.
    auto finish = [&](Lane& ln) -> int {
        ln.state = FINISHED;
        if (results) { results[ln.frame].rounds = ln.round; results[ln.frame].rays = ln.rays; }
        ln.next = -1;
        ln.marked = false;
        if (fold_into && next_frame < frames) {
            ln.next = (int)next_frame++;
            if (rebuild && fold_into->ctx->stream != ln.rt->ctx->stream) {
                // the end of the frame, marked before the rebuild goes onto the stream: the sample waits for the mark only
                PSM_HIP(ln.rt->ctx, hipEventRecord(ln.rt->ev_fold, ln.rt->ctx->stream));
                ln.marked = true;
                return psm_bvh_build(ln.bvh, opt);
            }
        }
        return PSM_OK;
    };
    // frame f on lane ln; `built`: its rebuild has been queued already (finish)
    auto start = [&](Lane& ln, uint32_t f, bool built) -> int {
        ln.frame = (int)f;
        ln.rand = frame_seeds[f];
        ln.round = 0;
        ln.rays = 0;
        if (rebuild && !(built && ln.marked)) {   // in front of the wait for the last frame's fold: the rebuild does not touch the texels sample() reads
            int e = psm_bvh_build(ln.bvh, opt);
            if (e != PSM_OK) return e;
        }
        if (ln.fold_wait) {
            PSM_HIP(ln.rt->ctx, hipStreamWaitEvent(ln.rt->ctx->stream, ln.rt->ev_fold, 0));
            ln.fold_wait = false;
        }
        int e = psm_rt_camera(ln.rt, cam_inv, proj_inv, lcg_next(ln.rand));
        if (e != PSM_OK) return e;
        if (depth == 0 || ln.rt->ray_count < 32) return finish(ln);  // Pipeline.inl:459-461
        ln.state = RUNNING;
        return queue_round(ln);
    };
    // PSM_LANES_PROFILE=1: how much of the wall time the scheduler thread spends issuing work
    const bool profile = getenv("PSM_LANES_PROFILE") != nullptr;  // debug print only, read per call
    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    double issue_s = 0.0, gpu_wait_s = 0.0, react_s = 0.0;  // profile: issue -> count seen, count seen -> next issue
    double turn_wait_s = 0.0;                                 // profile: frame ended -> its turn to fold
    uint32_t waits = 0;
    std::vector<clk::time_point> t_issued(lanes), t_seen(lanes);
    uint32_t next_fold = 0, idle_spins = 0;
    while (rc == PSM_OK && next_fold < frames) {
        bool progressed = false;
        for (uint32_t s = 0; s < lanes && rc == PSM_OK; s++) {
            Lane& ln = L[s];
            if (ln.state == IDLE && next_frame < frames) {
                const clk::time_point t0 = clk::now();
                rc = start(ln, next_frame++, false);
                t_issued[s] = clk::now();
                issue_s += std::chrono::duration<double>(t_issued[s] - t0).count();
                progressed = true;
            } else if (ln.state == RUNNING) {
                const int q = lane_poll_round(ln.rt);
                if (q == 0) continue;
                if (q < 0) { rc = q; break; }
                progressed = true;
                t_seen[s] = clk::now();
                gpu_wait_s += std::chrono::duration<double>(t_seen[s] - t_issued[s]).count();
                waits++;
                if (ln.round >= depth || ln.rt->ray_count < 32) { rc = finish(ln); t_issued[s] = clk::now(); }
                else {
                    const clk::time_point t0 = clk::now();
                    rc = queue_round(ln);
                    t_issued[s] = clk::now();
                    react_s += std::chrono::duration<double>(t0 - t_seen[s]).count();
                    issue_s += std::chrono::duration<double>(t_issued[s] - t0).count();
                }
            }
        }
        // sample() in frame order
        for (bool again = true; again && rc == PSM_OK;) {
            again = false;
            for (uint32_t s = 0; s < lanes; s++) {
                Lane& ln = L[s];
                if (ln.state == FINISHED && (uint32_t)ln.frame == next_fold) {
                    turn_wait_s += std::chrono::duration<double>(clk::now() - t_issued[s]).count();
                    if (fold_into) rc = fold(fold_into, ln.rt, &ln.fold_wait, ln.marked);
                    ln.state = fold_into ? IDLE : FINISHED;
                    ln.frame = fold_into ? -1 : -2;  // without fold_into the lane keeps its frame (frames <= lanes)
                    next_fold++;
                    again = progressed = true;
                    if (rc == PSM_OK && fold_into && ln.next >= 0) {   // the frame it claimed when it ended (rebuild queued there)
                        const clk::time_point t0 = clk::now();
                        rc = start(ln, (uint32_t)ln.next, true);
                        t_issued[s] = clk::now();
                        issue_s += std::chrono::duration<double>(t_issued[s] - t0).count();
                    }
                    break;
                }
            }
        }
        if (!progressed) {
            if (++idle_spins > 256) std::this_thread::yield();
        } else {
            idle_spins = 0;
        }
    }
    for (uint32_t s = 0; s < lanes; s++) {
        (void)hipStreamSynchronize(L[s].rt->ctx->stream);
        rts[s]->in_flight = 1;
    }
    if (fold_into) (void)hipStreamSynchronize(fold_into->ctx->stream);
    if (profile) {
        const double wall = std::chrono::duration<double>(clk::now() - t_begin).count();
        fprintf(stderr, "psm_lanes_render: %u frames on %u lanes, wall %.3f ms, issuing %.3f ms (%.0f %%); per round: issued -> count seen "
                        "%.1f us; per frame: ended -> its turn to fold %.1f us\n", frames, lanes, wall * 1e3, issue_s * 1e3, 100.0 * issue_s / wall,
                waits ? gpu_wait_s * 1e6 / waits : 0.0, frames ? turn_wait_s * 1e6 / frames : 0.0);
        (void)react_s;
    }
    return rc;
}

// ---- tile-sharded frames: free-running lanes that park instead of stopping ------------------------------
//
// With the frame sharded over several GPUs the `fewer than 32 rays -> stop` rule (Pipeline.inl:459-461) looks
// at the frame's GLOBAL count. A rank whose local count is >= 32 knows the global one is too and needs nobody;
// only a rank with fewer than 32 local rays has to ask. So every lane runs free, exactly as above, until its
// local count drops below 32 (or `depth` is reached) and then PARKS, keeping its queue. The host exchanges
// (round, count) of all lanes once everybody is parked -- normally once per batch, because the tiles of a frame
// run dry in the same round -- and, where the global count says the frame goes on, calls again with
// force_until[lane] = the round the lane has to reach regardless of its local count (it traces its few rays,
// or none, and draws its rand() every round so the ranks stay in step).
namespace psm {

// Free-running lanes of tile-sharded frames (see above): shared by psm_lanes_run_sharded and the pipelined batches of
// psm_dist_render_frames (dist.hip). A lane is IDLE (nothing queued), RUNNING (a round's kernels and the read-back of
// its ray count are in flight) or FINISHED (= parked: fewer than 32 local rays and no round forced, or depth reached).
struct ShardedLanes {
    psm_rt* const* rts;
    psm_bvh* const* bvhs;
    uint32_t lanes, depth;
    uint32_t* rand_state;
    uint32_t* rounds;
    std::vector<uint32_t> force_until;
    std::vector<LaneState> st;
    std::vector<uint64_t> traced;  // rays handed to traverse per lane since its frame began
    std::vector<uint32_t> pending; // waits a deferred gather / fold left to this lane's stream (LANE_WAIT_*)
    psm_dist* dist = nullptr;      // whose gather they wait for
    int rc = PSM_OK;

    ShardedLanes(psm_rt* const* r, psm_bvh* const* b, uint32_t n, uint32_t d, uint32_t* rs, uint32_t* rd)
        : rts(r), bvhs(b), lanes(n), depth(d), rand_state(rs), rounds(rd), force_until(n, 0u), st(n, FINISHED), traced(n, 0ull), pending(n, 0u) {}

    int step(uint32_t s) {  // park, or queue one more round
        psm_rt* r = rts[s];
        if (rounds[s] >= depth || (r->ray_count < 32 && rounds[s] >= force_until[s])) { st[s] = FINISHED; return PSM_OK; }
        uint32_t t = lcg_next(rand_state[s]);  // drawn every round, ray or no ray
        rounds[s]++;
        if (r->ray_count == 0) return PSM_OK;  // nothing to trace: the lane is re-examined at once
        traced[s] += r->ray_count;
        int e = lane_queue_round(r, bvhs[s], t);
        if (e != PSM_OK) return e;
        st[s] = RUNNING;
        return PSM_OK;
    }
    // begin a new frame on lane s: build (if rebuild) + camera, then rounds until it parks or a round is in flight
    void start(uint32_t s, uint32_t seed, const float* cam_inv, const float* proj_inv, int rebuild, const double* opt) {
        if (rc != PSM_OK) return;
        if (rebuild) rc = psm_bvh_build(bvhs[s], opt);   // in front of the waits: the rebuild touches nothing a gather or a fold reads
        if (rc == PSM_OK && pending[s]) rc = lane_flush_waits(dist, rts[s], &pending[s]);
        begin(s, seed, cam_inv, proj_inv);
    }
    // the same without the build (the hierarchy has been rebuilt by somebody else and this lane's stream waits for it)
    void begin(uint32_t s, uint32_t seed, const float* cam_inv, const float* proj_inv) {
        if (rc != PSM_OK) return;
        rand_state[s] = seed;
        rounds[s] = 0;
        force_until[s] = 0;
        traced[s] = 0;
        rc = psm_rt_camera(rts[s], cam_inv, proj_inv, lcg_next(rand_state[s]));
        st[s] = IDLE;
        while (rc == PSM_OK && st[s] == IDLE) rc = step(s);
    }
    // one non-blocking look at lane s: true when its round in flight has ended (the lane has then been stepped on)
    bool poll(uint32_t s) {
        if (rc != PSM_OK || st[s] != RUNNING) return false;
        const int q = lane_poll_round(rts[s]);
        if (q <= 0) { if (q < 0) rc = q; return false; }
        st[s] = IDLE;
        while (rc == PSM_OK && st[s] == IDLE) rc = step(s);
        return true;
    }
    // continue a parked lane at least up to round `until`
    void resume(uint32_t s, uint32_t until) {
        if (rc != PSM_OK) return;
        force_until[s] = until;
        st[s] = IDLE;
        while (rc == PSM_OK && st[s] == IDLE) rc = step(s);
    }
    // one non-blocking pass over every lane that has a round in flight (whoever holds the host thread for a while calls it)
    void poll_all() {
        for (uint32_t s = 0; s < lanes && rc == PSM_OK; s++)
            if (st[s] == RUNNING) (void)poll(s);
    }
    // drive EVERY lane that has a round in flight until lanes [g0, g1) are all parked (lanes outside keep running)
    int drive(uint32_t g0, uint32_t g1) {
        uint32_t idle_spins = 0;
        for (;;) {
            if (rc != PSM_OK) return rc;
            bool waiting = false, progressed = false;
            for (uint32_t s = 0; s < lanes && rc == PSM_OK; s++) {
                if (st[s] != RUNNING) continue;
                if (s >= g0 && s < g1) waiting = true;
                if (poll(s)) progressed = true;
            }
            if (!waiting) {
                bool again = false;
                for (uint32_t s = g0; s < g1; s++) again = again || st[s] == RUNNING;
                if (!again) return rc;
                continue;
            }
            if (!progressed) {
                if (++idle_spins > 256) std::this_thread::yield();
            } else {
                idle_spins = 0;
            }
        }
    }
};

}  // namespace psm

extern "C" int psm_lanes_run_sharded(psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                                     const float proj_inv[16], uint32_t* rand_state, uint32_t* rounds,
                                     const uint32_t* force_until, uint32_t depth, int start, int rebuild, const double* opt,
                                     int32_t* counts_out) {
    if (!rts || !bvhs || !rand_state || !rounds || !force_until || !counts_out || lanes == 0 || lanes > 64) return PSM_ERR_INVALID;
    if (start && (!cam_inv || !proj_inv)) return PSM_ERR_INVALID;
    for (uint32_t s = 0; s < lanes; s++) {
        if (!rts[s] || !bvhs[s]) return PSM_ERR_INVALID;
        int e = lane_resources(rts[s]);
        if (e != PSM_OK) return e;
    }
    (void)hipSetDevice(rts[0]->ctx->device);
    for (uint32_t s = 0; s < lanes; s++) rts[s]->in_flight = lanes;
    ShardedLanes L(rts, bvhs, lanes, depth, rand_state, rounds);
    for (uint32_t s = 0; s < lanes; s++) {
        if (start) L.start(s, rand_state[s], cam_inv, proj_inv, rebuild, opt);
        else L.resume(s, force_until[s]);
    }
    int rc = L.drive(0, lanes);
    for (uint32_t s = 0; s < lanes; s++) {
        (void)hipStreamSynchronize(rts[s]->ctx->stream);
        counts_out[s] = (int32_t)rts[s]->ray_count;
        rts[s]->in_flight = 1;
    }
    return rc;
}

// ---- pipelined batches of tile-sharded frames -----------------------------------------------------------------------
//
// psm_dist_render_batch (dist.hip) runs a batch of `lanes` frames start to finish: when the batch ends the chip drains
// -- every lane waits for the slowest one, the exchange and the gathers -- before the next batch starts all lanes at
// once. Here the lanes form TWO groups that alternate: while group g sits in its exchange / gather / fold, the other
// group's frames keep the chip busy, and group g starts its next frames as soon as its tiles are on their way. Every
// rank makes the same psm_dist_* calls in the same order (group 0's batch, group 1's, group 0's ...), so one
// communicator serves both groups. Frames fold in frame order: batch b = frames [b*h, (b+1)*h), h = lanes / 2.
// The host thread is the only one that queues a lane's next round, so whatever else it does for a batch -- the exchange and its
// wait for the peers, a frame's gather and fold, a lane's rebuild and camera -- it looks after every lane with a round in
// flight in between (ShardedLanes::poll_all): a sixth of a tile's wall time used to pass with nobody looking.
extern "C" int psm_dist_render_frames(psm_dist* d, psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                                      const float proj_inv[16], const uint32_t* frame_seeds, uint32_t frames, uint32_t depth,
                                      int rebuild, const double* opt, psm_rt* fold_into, uint32_t* rounds_out /* [frames] or NULL */) {
    if (!d || !rts || !bvhs || !cam_inv || !proj_inv || lanes == 0 || lanes > 64 || (frames && !frame_seeds)) return PSM_ERR_INVALID;
    const int rank = psm_dist_rank(d), world = psm_dist_world(d);
    if (rank == 0 && !fold_into) return PSM_ERR_INVALID;
    for (uint32_t s = 0; s < lanes; s++)
        if (!rts[s] || !bvhs[s]) return PSM_ERR_INVALID;
    if (frames == 0) return PSM_OK;
    (void)hipSetDevice(rts[0]->ctx->device);
    int pre = PSM_OK;  // from here on a failure is `local`: the other ranks are on their way into the exchanges
    for (uint32_t s = 0; s < lanes && pre == PSM_OK; s++) pre = lane_resources(rts[s]);
    for (uint32_t s = 0; s < lanes; s++) rts[s]->in_flight = lanes;
    const uint32_t groups = lanes >= 2 ? 2u : 1u;
    const uint32_t h0 = groups == 2 ? lanes / 2 : lanes;         // group 0: lanes [0, h0), group 1: [h0, lanes)
    const uint32_t gbeg[2] = {0u, h0}, gend[2] = {h0, lanes};
    std::vector<uint32_t> state(lanes, 0u), rounds(lanes, 0u);
    ShardedLanes L(rts, bvhs, lanes, depth, state.data(), rounds.data());
    L.dist = d;
    // batches: consecutive runs of frames, alternating between the groups
    struct Batch { uint32_t f0, n, g; };
    std::vector<Batch> batches;
    for (uint32_t f = 0, b = 0; f < frames; b++) {
        const uint32_t g = b % groups, cap = gend[g] - gbeg[g];
        const uint32_t n = std::min(cap, frames - f);
        batches.push_back(Batch{f, n, g});
        f += n;
    }
    auto start_batch = [&](const Batch& B) {
        for (uint32_t k = 0; k < B.n; k++) {
            L.start(gbeg[B.g] + k, frame_seeds[B.f0 + k], cam_inv, proj_inv, rebuild, opt);
            L.poll_all();
        }
    };
    // `local` is this rank's own failure. A failed rank keeps the collective sequence (rounds = -1 at the exchanges,
    // placeholder tiles in gathers already decided) until an exchange has told everybody; `rc` is the call's fate as all
    // ranks see it. Only a failing transport call leaves at once: there is no sequence left to keep.
    int local = pre != PSM_OK ? pre : dist_reserve(d, rts[0]->w, rts[0]->h), rc = PSM_OK;
    bool transport_dead = false;
    if (local == PSM_OK) {
        for (size_t b = 0; b < batches.size() && b < groups; b++) start_batch(batches[b]);
        local = L.rc;
    }
    std::vector<int32_t> mine, all, verdict;
    std::vector<uint32_t> force;
    d->idle_user = &L;
    d->idle_fn = [](void* u) { ((ShardedLanes*)u)->poll_all(); };
    for (size_t b = 0; b < batches.size() && rc == PSM_OK && !transport_dead; b++) {
        const Batch& B = batches[b];
        const uint32_t g0 = gbeg[B.g], n = B.n;
        mine.assign(2 * (size_t)n, 0); all.assign(2 * (size_t)n * (size_t)world, 0); verdict.assign(n, 0); force.assign(n, 0u);
        std::vector<int32_t> over(n, 0);
        if (local == PSM_OK) local = L.drive(g0, g0 + n);
        for (;;) {
            for (uint32_t k = 0; k < n && local == PSM_OK; k++) {
                // (a parked lane's last count has been read behind the event that follows its last kernel: nothing of its frame is
                // in flight, and no host synchronisation is needed -- 30 us apiece with a dozen streams about)
                psm_rt* r = rts[g0 + k];
                mine[k] = (int32_t)rounds[g0 + k];
                mine[n + k] = (int32_t)r->ray_count;
            }
            if (local != PSM_OK) for (uint32_t k = 0; k < n; k++) mine[k] = -1;
            const std::string keep = d->ctx->err;
            rc = psm_dist_allgather_i32(d, mine.data(), all.data(), 2 * n);
            if (rc != PSM_OK) { transport_dead = true; break; }
            rc = psm_dist_decide((uint32_t)world, n, all.data(), depth, verdict.data(), force.data());
            if (rc != PSM_OK) { if (local != PSM_OK) d->ctx->err = keep; break; }  // everybody leaves at this exchange
            bool done = true;
            for (uint32_t k = 0; k < n; k++) { over[k] |= verdict[k]; done = done && over[k]; }
            if (done) break;
            for (uint32_t k = 0; k < n; k++)
                if (!over[k]) L.resume(g0 + k, force[k]);
            local = L.rc;
            if (local == PSM_OK) local = L.drive(g0, g0 + n);
        }
        if (rc != PSM_OK) break;
        for (uint32_t k = 0; k < n; k++) {  // frame order; a rank that fails here still takes part in the gathers that are due
            if (!dist_frame_gather(d, rts[g0 + k], fold_into, local, &L.pending[g0 + k])) { transport_dead = true; break; }
            if (rounds_out) rounds_out[B.f0 + k] = rounds[g0 + k];
            L.poll_all();
        }
        if (!transport_dead && local == PSM_OK && b + groups < batches.size()) {  // this group's next frames (their camera() waits for the gather: stream order)
            start_batch(batches[b + groups]);
            local = L.rc;
        }
    }
    d->idle_fn = nullptr;
    d->idle_user = nullptr;
    if (!transport_dead && rc == PSM_OK) rc = psm_dist_agree(d, local);  // a failure after the last decision reaches everybody here
    if (local != PSM_OK) rc = local;
    for (uint32_t s = 0; s < lanes; s++) {
        if (L.pending[s]) (void)lane_flush_waits(d, rts[s], &L.pending[s]);   // the last frames' gathers: nobody rebuilt in front of them
        (void)hipStreamSynchronize(rts[s]->ctx->stream);
        rts[s]->in_flight = 1;
    }
    if (fold_into) (void)hipStreamSynchronize(fold_into->ctx->stream);
    return rc;
}


