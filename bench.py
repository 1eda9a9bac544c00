#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native path-tracing core.

Metric (BASELINE.json): Mrays/s and ms/frame on the Sponza-class scene, 1920x1080, 4 spp,
at 1/2/4/8 MI355X.  One STEP = one frame = one pass of the hot path over one batch of input:
full BVH rebuild (bounds, Morton, radix sort, emit) + camera + the bounce loop
(traverse -> surface/shade -> queue hand-off) + sample, exactly the call order of
GltfViewer::process() (reference Source/Examples/Viewer.cpp:296-312).  4 spp = 4 steps.

  python bench.py --gpus N --steps K --warmup W
  N>1: either under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (python -m torch.distributed.run
  --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), or plainly as above: bench.py then starts the N ranks
  itself as child processes (before it has touched a GPU) and rank 0 prints the line.

Multi-GPU: the frame is sharded by screen rows; the BVH is rebuilt redundantly on every GPU; the
per-texel radiance of each tile is gathered to rank 0 (RCCL over xGMI) which runs the sampler.
Total work is fixed by the config, so scaling is "strong".

Rank 0 prints ONE JSON line.  `value` = rays traced by all ranks / wall time of the K timed frames
(inputs resident in HBM).  `roofline` prices the traversal kernel THE TIMED REGION LAUNCHES (with frames in
flight: the hand-over kernel rt_traverse<false,false,true>): HIP events around every traversal launch on every
lane's stream in a repeat of the timed call, algorithmic bytes (R*44 + V*64 + T*36, from per-round
counters) over that time, and first the physical figures of the committed PMC passes (profiles/traffic_r04.json:
HBM-side traffic, VALU issue and lane utilisation); the single-launch kernel of a frame running alone is priced
beside it.  `cpu_baseline` times the CPU oracle's traversal (oracle/, scalar C + OpenMP, every hardware thread)
on a bounded sample of the same rays.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed frames (with 4 in flight a run of 8 is half ramp-up and drain)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="sponza_like", choices=["sponza_like", "cornell", "stress"])
    ap.add_argument("--textured", action="store_true",
                    help="not the headline workload: add texcoords + the procedural texture table (SURVEY f2)")
    ap.add_argument("--depth", type=int, default=16)  # Application.hpp:237
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-build-graph", action="store_true", help="rebuild with plain launches instead of the captured hipGraph (A/B)")
    ap.add_argument("--cpu-sample-rays", type=int, default=7_000_000, help="cap of the CPU baseline's sample (C3: all 6.3 M rays of frame 0, ~25 CPU-seconds per run)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline; 0 (default) = every hardware thread of the box")
    ap.add_argument("--band-weights", default="default",
                    help="N > 1: bands per period for each rank, comma separated; 'default' = dist.default_band_weights (the gathering "
                         "rank owns fewer bands: it also unpacks, fills and samples the whole image); 'none' = round-robin")
    ap.add_argument("--no-obj-roundtrip", action="store_true",
                    help="feed the generator's triangle arrays straight to loadTriangles instead of through scenes.write_obj / read_obj")
    ap.add_argument("--emulate-tile", default=None, help="R/W: render only the tile of rank R of W on one GPU, no communication (Amdahl study)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL code path even on 1 GPU (rehearsal)")
    ap.add_argument("--no-rebuild", action="store_true", help="study only, not the headline workload: build the BVH once")
    ap.add_argument("--lanes", type=int, default=0,
                    help="frames in flight per GPU (psm_lanes_render): each on its own HIP stream, folded into the "
                         "accumulating image in frame order; 1 = one frame after another; 0 (default) = 4 on one GPU, "
                         "12 per GPU on several, with 16 hardware queues (a tile's launches are small and latency-bound: more frames in "
                         "flight fill the chip; measured with --force-dist --emulate-tile R/W, profiles/r04_tile_emulation.txt)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous check only: every rank builds its communicator (gloo, no GPU), proves the group works "
                         "with one all-reduce, prints one line and exits")
    ap.add_argument("--diag-clock", action="store_true",
                    help="diagnosis only: a second pass of the timed region with the counting kernels on every lane; reports "
                         "the shader clock the traversal waves ran at (s_memtime / s_memrealtime) and their wave-steps to stderr")
    ap.add_argument("--traverse", default="auto", choices=["auto", "whole", "phased", "adaptive"],
                    help="tuning study: traversal kernel schedule (psm_rt_set_traverse_mode); results never depend on it")
    ap.add_argument("--trav-caps", default="96", help="phased: wave-step caps, comma separated")
    ap.add_argument("--trav-adaptive", default="", help="adaptive: min_live,min_steps,final_rays,max_launches,min_rays")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the timed call of K steps is taken; the line reports the "
                                                            "median call, and all of them under \"timing\"")
    ap.add_argument("--solo", type=int, default=-1, help="tuning study: rays a traversal wave takes into the solo gear at most "
                                                          "(psm_rt_set_traverse_solo, 0..4; -1 = the library's default)")
    return ap.parse_args()


class Renderer:
    def __init__(self, psm, scenes, scene, args, dist):
        self.psm, self.dist, self.args = psm, dist, args
        self.scene = scene
        self.pdist = importlib.import_module("prismarine-core_amd.dist")
        stream = None
        self.native = None
        # the C ABI's own RCCL communicator carries the collectives unless PSM_DIST_NATIVE=0 (or the one-GPU rehearsal of
        # another rank's tile, whose tile is not this rank's)
        self.hoststaged = os.environ.get("PSM_DIST_TRANSPORT", "rccl") == "hoststaged"   # rehearsal on a shared GPU, asked for explicitly
        use_native = dist.active and (dist.backend == "nccl" or self.hoststaged) and os.environ.get("PSM_DIST_NATIVE", "1") != "0"
        if dist.active and dist.backend == "nccl" and not use_native:
            # torch.distributed collectives: run the kernels on torch's stream, so RCCL calls and kernels are ordered
            # without host syncs. (Default: the C ABI's own communicator, psm_dist_*, ordered by events.)
            stream = dist.torch.cuda.current_stream().cuda_stream
            dist.same_stream = True
        w, h = args.width, args.height
        # frames in flight: 4 on one GPU; 12 for a tile of a sharded frame (a tile's launches are all tail: profiles/r04_tile_emulation.txt,
        # 8 -> 12 lanes with 16 hardware queues: a worker's 1/8 tile 0.611 -> 0.588 ms, C5's 3.82 -> 3.64)
        self.lanes = args.lanes if args.lanes > 0 else ((8 if self.hoststaged else 12) if (dist.world > 1 or args.emulate_tile) else 4)
        self.active_lanes = self.lanes   # frames in flight the sharded timed region uses: main() may settle for fewer (lane calibration)
        self.lane_streams = None
        streams = None
        if stream is not None:  # lane 0 on torch's current stream, the others on torch side streams
            torch = dist.torch
            self.lane_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(self.lanes - 1)]
            streams = [st.cuda_stream for st in self.lane_streams]
        self.batch = psm.FrameBatch(self.lanes, w, h, device=dist.device_index, seed=1000, streams=streams)
        self.ctx, self.th, self.rt = self.batch.lanes[0].ctx, self.batch.lanes[0].th, self.batch.lanes[0].rays
        for b in [self.batch]:
            b.allocate(scene["tris"].shape[0])
            if args.no_build_graph:
                for ln in b.lanes:
                    ln.th.setBuildGraph(False)
            b.loadTriangles(scene["tris"], scene["normals"], scene["mats"], scene.get("texcoords"))
        self.ms = psm.MaterialSet()
        for m in scene["materials"]:
            self.ms.addSubmat(m)
        if scene.get("textures"):
            ts = psm.TextureSet()
            for slot in sorted(scene["textures"]):
                assert ts.loadTexture(scene["textures"][slot]) == slot
            self.ms.setTextureSet(ts)
        self.batch.applyMaterials(self.ms)
        if args.solo >= 0:
            self.batch.each(lambda r: r.setTraverseSolo(args.solo))
        if args.traverse != "auto" or args.trav_adaptive:
            def tune(r):
                if args.trav_adaptive:
                    r.setTraverseAdaptive(*[int(v) for v in args.trav_adaptive.split(",")])
                if args.traverse == "phased":
                    r.setTraversePhases([int(v) for v in args.trav_caps.split(",")])
                r.setTraverseMode(args.traverse)
            self.batch.each(tune)
        def weights_for(world):
            if args.band_weights == "none" or world < 2:
                return None
            if args.band_weights == "default":
                return self.pdist.default_band_weights(world)
            v = [int(x) for x in args.band_weights.split(",")]
            assert len(v) == world, "--band-weights needs one weight per rank"
            return v
        self.weights = None
        if dist.active and not (args.emulate_tile and dist.world == 1):
            self.weights = weights_for(dist.world)
            self.batch.each(lambda r: r.setTileInterleaved(dist.rank, dist.world, self.weights))  # 8-row bands dealt to the ranks
            dist.initial_total = w * h
        elif args.emulate_tile:
            r_, w_ = (int(v) for v in args.emulate_tile.split("/"))
            self.weights = weights_for(w_)
            self.batch.each(lambda r: r.setTileInterleaved(r_, w_, self.weights))
            dist.initial_total = self.rt.tile_texels()
        self.cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
        if use_native:
            tile_world = w_ if (args.emulate_tile and dist.world == 1) else dist.world   # (an emulated tile is dealt for W ranks)
            self.native = dist.attach_native(  # collective: every rank creates its communicator here (RCCL unless a rehearsal asked otherwise)
                self.ctx, self.pdist.largest_tile_texels(tile_world, w, h, self.weights) * 16)
            if args.emulate_tile and dist.world == 1:
                self.native.emulate_tile(r_, w_)
            self.native.set_band_weights(self.weights)
        if dist.active and self.native is None:
            torch = dist.torch
            gdev = torch.device("cuda", dist.device_index)
            self.per = self.pdist.largest_tile_texels(dist.world, w, h, self.weights) * 4  # every rank sends the largest tile's size
            self.tile_dev = torch.zeros(self.per, dtype=torch.float32, device=gdev)
            self.lane_tiles = [self.tile_dev] + [torch.zeros(self.per, dtype=torch.float32, device=gdev) for _ in range(self.lanes - 1)]
            self.gather_done = [None] * self.lanes
            self.all_dev = torch.zeros(dist.world * self.per, dtype=torch.float32, device=gdev) if dist.rank == 0 else None

    def frames_in_flight(self, k):
        """k x process() with `lanes` frames in flight (native scheduler), folded in frame order."""
        if self.args.no_rebuild and not getattr(self, "_built_once", False):
            for ln in self.batch.lanes:
                ln.th.markDirty()
                ln.th.build()
            self._built_once = True
        return self.batch.render(k, self.scene["eye"], self.scene["view"], depth=self.args.depth,
                                 rebuild=not self.args.no_rebuild)

    def frames_in_flight_sharded(self, k):
        """k x process() of a tile-sharded frame with `lanes` frames in flight on every rank. RCCL: the lanes run
        free until their local ray counts park them and one small all-gather per batch applies the global stop
        rule (dist.run_batch_sharded); gloo rehearsal: lock-step rounds with a host exchange
        (dist.run_rounds_lanes). One gather per frame, rank 0 folds the frames in order. Returns the rays this rank
        traced."""
        dist, batch, ms = self.dist, self.batch, self.ms
        torch = dist.torch
        w, h = self.args.width, self.args.height
        traced = 0
        if self.native is not None and os.environ.get("PSM_DIST_PIPELINE", "1") != "0":
            # all k frames in the C ABI: rounds, exchanges, ONE gather per frame, fold -- no drain between batches
            before = [ln.ctx.stats().rays_traced for ln in batch.lanes]
            batch.render_frames_sharded(self.native, batch.frame_seeds(k), self.cam[0], self.cam[1], depth=self.args.depth, lanes=self.active_lanes)
            return sum(ln.ctx.stats().rays_traced - b for ln, b in zip(batch.lanes, before))
        for f0 in range(0, k, self.lanes):
            seeds = batch.frame_seeds(min(self.lanes, k - f0))
            lanes = batch.lanes[: len(seeds)]
            before = [ln.ctx.stats().rays_traced for ln in lanes]
            if self.native is not None:  # the whole batch in the C ABI: rounds, exchanges, one gather per frame, fold
                batch.render_batch_sharded(self.native, seeds, self.cam[0], self.cam[1], depth=self.args.depth)
                traced += sum(ln.ctx.stats().rays_traced - b for ln, b in zip(lanes, before))
                continue
            if dist.backend == "nccl":
                self.pdist.run_batch_sharded(dist, batch, self.cam[0], self.cam[1], seeds, self.args.depth)
            else:  # host-exchange rehearsal path (gloo)
                for ln, sd in zip(lanes, seeds):
                    ln.rays.setSeed(sd)
                    ms.loadToVGA()
                    ln.th.markDirty()
                    ln.th.build()
                    ln.rays.camera_matrices(self.cam[0], self.cam[1])
                self.pdist.run_rounds_lanes(dist, [(ln.rays, ln.th, ms) for ln in lanes], self.args.depth,
                                            initial_totals=[dist.initial_total] * len(lanes))
            traced += sum(ln.ctx.stats().rays_traced - b for ln, b in zip(lanes, before))
            for s, ln in enumerate(lanes):  # frame order
                if self.native is not None:  # ONE collective per frame, in the C ABI (psm_dist_gather_tiles)
                    self.native.gather_tiles(ln.rays)
                    if dist.rank == 0:
                        batch.fold_one(ln)
                    continue
                if dist.backend == "nccl" and self.gather_done[s] is not None:
                    self.lane_streams[s].wait_event(self.gather_done[s])  # the previous gather has read this tile buffer
                ln.rays.pack_texels_dev(self.lane_tiles[s].data_ptr())
                if dist.backend == "nccl":
                    main = torch.cuda.current_stream()
                    main.wait_stream(self.lane_streams[s])
                    dist.gather_to_root(self.lane_tiles[s], self.all_dev)
                    self.gather_done[s] = main.record_event()
                    if dist.rank == 0:
                        self.lane_streams[s].wait_event(self.gather_done[s])  # the unpack below reads all_dev
                else:  # host-staged rehearsal path (gloo)
                    ln.ctx.sync()
                    got = dist.gather_to_root(self.lane_tiles[s].cpu())
                    if dist.rank == 0:
                        self.all_dev.copy_(got)
                        torch.cuda.synchronize()
                if dist.rank == 0:
                    ln.rays.unpack_tiles_dev(dist.world, 0, self.all_dev.data_ptr(), self.per)  # the other ranks' tiles, one launch
                    if dist.backend == "nccl":  # all_dev is reused by the next lane's gather
                        torch.cuda.current_stream().wait_stream(self.lane_streams[s])
                    batch.fold_one(ln)
        return traced

    def frame(self, record=None, round_log=None):
        """GltfViewer::process(), Viewer.cpp:296-312 (display excluded): one frame on lane 0, with the rand()
        stream the FrameBatch policy gives it (one draw of the accumulating stream seeds the frame's own).
        round_log: a list that receives (rays, V, T) of every bounce round (counting pass: V and T per round are what
        each traversal launch schedule is priced with)."""
        rt, th, ms, dist = self.rt, self.th, self.ms, self.dist
        if not hasattr(self, "_master_state"):
            self._master_state = 1000
        self._master_state = (self._master_state * 214013 + 2531011) & 0xFFFFFFFF
        rt.setSeed((self._master_state >> 16) & 0x7FFF)
        ms.loadToVGA()
        th.markDirty()
        th.build()
        rt.camera_matrices(self.cam[0], self.cam[1])
        snaps = []

        def on_round(local):
            if record is not None and local > 0:
                record.append(rt.download_rays())
            if round_log is not None:
                st_ = self.ctx.stats()
                snaps.append((local, st_.node_visits, st_.tri_tests))
        self.pdist.run_rounds(dist, rt, th, ms, self.args.depth, on_round if (record is not None or round_log is not None) else None)
        if round_log is not None:
            st_ = self.ctx.stats()
            snaps.append((0, st_.node_visits, st_.tri_tests))
            for a, b in zip(snaps, snaps[1:]):
                round_log.append((a[0], b[1] - a[1], b[2] - a[2]))
        self._gather()
        if dist.rank == 0:
            rt.sample()

    def _gather(self):
        """ONE collective per frame: per-texel radiance of every tile -> rank 0 (RCCL gather over xGMI)."""
        dist = self.dist
        if not dist.active:
            return
        if self.native is not None:
            self.native.gather_tiles(self.rt)
            return
        torch = dist.torch
        self.rt.pack_texels_dev(self.tile_dev.data_ptr())
        if dist.backend == "nccl":
            dist.gather_to_root(self.tile_dev, self.all_dev)  # same stream as the kernels: no host sync
        else:  # host-staged rehearsal path (gloo)
            self.ctx.sync()
            got = dist.gather_to_root(self.tile_dev.cpu())
            if dist.rank == 0:
                self.all_dev.copy_(got)
                torch.cuda.synchronize()
        if dist.rank == 0:
            self.rt.unpack_tiles_dev(dist.world, 0, self.all_dev.data_ptr(), self.per)  # the other ranks' tiles, one launch


def sharded_self_check(psm, scenes, dist, R):
    """A run on several ranks proves itself before it is timed: a small frame set -- S-sponza-like with 20 011 triangles, 160 x 90
    (12 bands, the last one two rows: padded tiles), 3 frames on 2 lanes -- through the SAME communicator, dealing and entry
    point as the timed region (psm_dist_render_frames), and on rank 0 the same frames unsharded (psm_lanes_render); the two
    images must agree -- deposit counts exactly, radiance to float-atomic order (1e-5) -- and every rank must have run the
    unsharded rounds per frame. A wrong dealing, a rank that idles, a gather that delivers another frame's tile cannot then pass
    for a plausible Mrays/s: every rank leaves with exit code 4. Returns what the JSON line reports.
    (PSM_BENCH_SABOTAGE_RANK=r: that rank looks at the scene from somewhere else -- the test of this check.)"""
    w, h, frames, lanes, seed = 160, 90, 3, 2, 4242
    sc = scenes.sponza_like(n_tris=20011)
    ms = psm.MaterialSet()
    for m in sc["materials"]:
        ms.addSubmat(m)

    def batch(tiled):
        b = psm.FrameBatch(lanes, w, h, device=dist.device_index, seed=seed)
        b.allocate(sc["tris"].shape[0])
        b.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
        b.applyMaterials(ms)
        if tiled:
            b.each(lambda r: r.setTileInterleaved(dist.rank, dist.world, R.weights))
        return b

    eye = np.asarray(sc["eye"], np.float32).copy()
    if os.environ.get("PSM_BENCH_SABOTAGE_RANK") == str(dist.rank):
        eye[0] += 0.75
    cam = scenes.camera_matrices(eye, sc["view"], w, h)
    ok, why, img, rounds = 1, "", None, None
    b = batch(True)
    try:
        rounds = b.render_frames_sharded(R.native, b.frame_seeds(frames), cam[0], cam[1], depth=16)
        if dist.rank == 0:
            img = b.snapHdr()
    except psm.PsmError as e:
        ok, why = 0, "rank %d: %s" % (dist.rank, e)
    b.close()
    worst = None
    want_r = [0] * frames
    if dist.rank == 0 and ok:
        u = batch(False)
        want_r = [int(r) for r, _ in u.render(frames, sc["eye"], sc["view"], depth=16)]
        want = u.snapHdr()
        u.close()
        worst = float((np.abs(img[..., :3] - want[..., :3]) / np.maximum(np.abs(want[..., :3]), 0.1)).max())
        if not np.array_equal(img[..., 3], want[..., 3]):
            ok, why = 0, "deposit counts of the sharded image differ from the unsharded one in %d texels" % int((img[..., 3] != want[..., 3]).sum())
        elif not np.allclose(img[..., :3], want[..., :3], rtol=1e-5, atol=1e-6):
            ok, why = 0, "radiance of the sharded image differs from the unsharded one (largest difference relative to max(|value|, 0.1): %.3g)" % worst
        elif not want[..., :3].max() > 0.05:
            ok, why = 0, "the check frames are black"
    want_r = [int(v) for v in dist.sum_ints(want_r)]          # rank 0's unsharded rounds per frame, to everybody
    if ok and rounds is not None and sum(want_r) > 0 and [int(v) for v in rounds] != want_r:   # (all zeros: rank 0 had failed before it got there)
        ok, why = 0, "rank %d ran %s rounds per frame, the unsharded frames %s" % (dist.rank, [int(v) for v in rounds], want_r)
    if dist.min_int(ok) == 0:
        if not ok:
            print("bench: the sharded self-check FAILED: %s" % why, file=sys.stderr, flush=True)
        dist.close()
        sys.exit(4)
    return {"result": "ok", "what": "S-sponza-like 20011 tris, %dx%d, %d frames on %d lanes through psm_dist_render_frames on %d ranks "
                                    "against psm_lanes_render on rank 0: deposit counts equal, radiance within 1e-5, rounds per frame %s on every rank"
                                    % (w, h, frames, lanes, dist.world, want_r),
            "largest_difference": worst}


def cpu_baseline(scene, ray_sets, args):
    """CPU oracle traversal (scalar C restatement, OpenMP over rays) on a bounded sample of the GPU's ray set, on every
    hardware thread of the box; best of 3 after a warm-up."""
    from oracle import oracle as O
    O.build()
    rays = np.concatenate(ray_sets)
    n = rays.shape[0]
    if n > args.cpu_sample_rays:
        stride = n // args.cpu_sample_rays + 1
        rays = rays[::stride]
    t0 = time.time()
    ob = O.build_scene(scene["tris"])
    build_s = time.time() - t0
    t0 = time.time()
    O.radix_sort(ob["keys_unsorted"], ob["idx"])
    sort_s = time.time() - t0
    hw = os.cpu_count() or 1                      # std::thread::hardware_concurrency() of the box
    try:
        usable = len(os.sched_getaffinity(0))     # what this process may run on (a container's CPU share)
    except (AttributeError, OSError):
        usable = hw
    quota = None                                  # the container's CPU quota, which affinity does not show (cgroup v2 / v1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            v = parse(open(path).read())
            if v:
                quota = max(1, int(np.ceil(v)))
                break
        except (OSError, ValueError, IndexError):
            pass
    origins = np.ascontiguousarray(rays["origin"])
    directs = np.ascontiguousarray(rays["direct"])
    # every hardware thread the process may use -- unless fewer are faster: a container with a CPU share smaller than the
    # box (gpurun: 16 of 256) runs 256 threads slower than 16. Calibrated on a slice of the sample, halving from the top.
    cal = slice(0, rays.shape[0], max(1, rays.shape[0] // 1_000_000))   # ~1 M rays spread over all bounce rounds
    corig, cdir = np.ascontiguousarray(origins[cal]), np.ascontiguousarray(directs[cal])
    O.traverse(ob["nodes"], scene["tris"], ob["M"], corig, cdir, min(usable, 16), want_hits=False)  # warm-up
    tried = {}
    cands = [args.cpu_threads] if args.cpu_threads > 0 else sorted({usable, quota or usable} | {max(1, usable >> k) for k in range(1, 6)}, reverse=True)
    for tcount in cands:
        dt = None
        for _ in range(2):   # best of 2: one timing of a short run is noise
            t0 = time.time()
            O.traverse(ob["nodes"], scene["tris"], ob["M"], corig, cdir, tcount, want_hits=False)
            dt = min(dt, time.time() - t0) if dt is not None else time.time() - t0
        tried[tcount] = corig.shape[0] / dt / 1e6
    threads = max(tried, key=tried.get)
    best = None
    for _ in range(3):
        t0 = time.time()
        O.traverse(ob["nodes"], scene["tris"], ob["M"], origins, directs, threads, want_hits=False)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    return {"value": rays.shape[0] / best / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "hardware_concurrency": hw, "usable_threads": usable, "cpu_quota": quota, "threads_tried_mrays_s": {str(k): v for k, v in sorted(tried.items())},
            "build_ms": build_s * 1e3, "sort_mkeys_s": ob["count"] / sort_s / 1e6, "build_sort_cores": 1,
            "sample": "%d of the %d rays of frame 0 (all bounce rounds, every %d-th ray), oracle psmo_traverse_batch on %d threads "
                      "(box: hardware_concurrency %d, affinity %d, CPU quota of this container %s; thread count = the fastest of %s on a "
                      "%d-ray sub-sample, best of 2 each: a CPU share smaller than the box runs all hardware threads slower than fewer), best of 3 after a "
                      "warm-up; oracle BVH build %.2f s on 1 core" % (
                          rays.shape[0], n, max(1, n // max(rays.shape[0], 1)), threads, hw, usable, quota,
                          ", ".join("%d: %.1f Mrays/s" % (k, v) for k, v in sorted(tried.items())), corig.shape[0], build_s)}


L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md: L2 aggregate, streaming
MALL_GATHER_GBS = 8600.0     # MI355X_MICROARCH.md: uniformly random rows of a 38 MB table (Infinity Cache), chip-wide
VALU_QUAD_CYCLES = 4.0       # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md cycle constants)
SIMDS = 1024


KERNEL_SOURCES = ("prismarine-core_amd/csrc/trace.hip", "prismarine-core_amd/csrc/shade.hip", "prismarine-core_amd/csrc/psm_common.h",
                  "prismarine-core_amd/csrc/psm_math.h", "prismarine-core_amd/csrc/psm_internal.h")
TRAFFIC_FILES = ("traffic_r05.json", "traffic_r04.json", "traffic_r03.json")


def traffic_provenance(root=None):
    """The newest committed counter file and whether its counters still describe the kernels this run executes: the file
    carries the sha256 of the traversal / shading sources its rocprofv3 passes ran with (profiles/collect.sh,
    make_traffic.py); they are compared with the sources beside this bench.py. Returns (path, provenance, stale, why)."""
    import hashlib
    root = root or ROOT
    for name in TRAFFIC_FILES:
        path = os.path.join(root, "profiles", name)
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        prov = d.get("provenance") or {}
        want = prov.get("sources_sha256") or {}
        if not want:
            return path, prov, True, "profiles/%s carries no source hashes (collected before round 5): its counters cannot be tied to the kernels of this run" % name
        for rel in KERNEL_SOURCES:
            try:
                have = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()
            except OSError:
                return path, prov, True, "%s is not beside bench.py" % rel
            if want.get(rel) != have:
                return path, prov, True, "%s has changed since the counters of profiles/%s were collected (commit %s)" % (rel, name, prov.get("commit"))
        return path, prov, False, None
    return None, {}, True, "no committed counter file"


def pmc_entry(scene, width, height, kernel, root=None):
    """Per-launch PMC figures of `kernel` for this workload from the committed rocprofv3 passes (profiles/collect.sh ->
    profiles/make_traffic.py): bench.py cannot run the profiler on itself. None when the newest counter file is stale
    (traffic_provenance): a replayed counter that describes another kernel is worse than none."""
    path, prov, stale, _ = traffic_provenance(root)
    if path is None or stale:
        return None, None
    for e in json.load(open(path)).get("entries", []):
        if (e.get("scene"), e.get("width"), e.get("height")) == (scene, width, height) and e.get("kernel", "").endswith(kernel):
            return e, "profiles/" + os.path.basename(path)
    return None, None


def price(kernel, launches, total_ms, R, V, T, rounds, steps, scene, width, height, how, use_pmc=True, root=None):
    """One traversal kernel priced over `launches` launches that took `total_ms` (HIP events on the launching streams) and
    traced R rays with V node visits and T triangle tests in `rounds` intersections."""
    launches = max(int(launches), 1)
    sec = total_ms * 1e-3
    alg = R * 44 + V * 64 + T * 36
    own = R * 44 + V * 32 + T * 48           # what the kernel's own records amount to
    out = {"kernel": kernel, "launches": launches, "rounds": int(rounds), "launches_per_round": launches / max(rounds, 1),
           "avg_launch_ms": total_ms / launches, "ms_per_step": total_ms / steps, "measured": how,
           "achieved": alg / sec / 1e9 if sec > 0 else 0.0, "algorithmic_bytes_per_launch": alg / launches,
           "own_record_bytes_per_launch": own / launches, "own_record_gbs": own / sec / 1e9 if sec > 0 else 0.0,
           "node_visits_per_s": V / sec if sec > 0 else None, "R": int(R), "V": int(V), "T": int(T)}
    # the committed counter passes are of whole frames on one GPU: a tile's launches are smaller, their traffic is not these
    e, src = pmc_entry(scene, width, height, kernel, root) if use_pmc else (None, None)
    if e is not None and sec > 0:
        traffic = e.get("traffic_bytes_per_launch")
        out["traffic"] = traffic
        out["traffic_source"] = src + ": " + e.get("source", "")
        out["traffic_fetch_raw"] = e.get("fetch_bytes_per_launch_raw")
        if traffic:
            # the counters' launches and this run's launches are the same kernel on the same workload (same launches per
            # frame): per-launch traffic over per-launch time is the HBM-side rate of ONE launch while it runs
            out["hbm_gbs_of_one_launch"] = traffic * launches / sec / 1e9
            out["hbm_frac"] = out["hbm_gbs_of_one_launch"] / HBM_PEAK_GBS   # replaced below where launches overlap
            out["traffic_over_own_record_bytes"] = traffic * launches / own if own else None
        if e.get("l2_hit") is not None:
            out["l2_hit"] = e["l2_hit"]
        sq = e.get("sq_per_launch") or {}
        if sq.get("SQ_ACTIVE_INST_VALU") and e.get("alone_avg_us"):
            # share of the launch (running alone, as the profiler serialises it) during which a SIMD's VALU executes
            out["valu_issue_utilisation"] = sq["SQ_ACTIVE_INST_VALU"] * VALU_QUAD_CYCLES / SIMDS / (e["alone_avg_us"] * 1e-6 * 2.4e9)
            out["valu_insts_per_launch"] = sq.get("SQ_INSTS_VALU")
        if e.get("valu_lane_utilisation") is not None:
            out["valu_lane_utilisation"] = e["valu_lane_utilisation"]
    return out


def roofline(timed, whole, bytes_per_step, ms_per_step, copy_gbs, root=None):
    """What bounds the dominant kernel of the TIMED region: `timed` = price() of the traversal kernel the timed schedule
    launches (HIP events on every lane's stream while the frames are in flight), `whole` = the single-launch kernel of a
    frame running alone, for comparison. The counters' figures lead; `achieved` / `frac` keep SURVEY 8(d)'s algorithmic
    bytes over the launch time for continuity -- that figure prices a node visit at 64 B where the kernel fetches one 32-B
    record, most of them from cache, so it is NOT an HBM utilisation (hbm_frac is)."""
    t = timed
    traffic, own = t.get("traffic"), t["own_record_bytes_per_launch"]
    if t.get("hbm_frac") is not None and traffic >= 0.5 * own:
        bound = "hbm" if t["hbm_frac"] >= 0.4 else "hbm latency (HBM-resident records, divergent 32-byte gathers)"
    elif t.get("hbm_frac") is not None:
        bound = "cache/VALU: records served by L2 + Infinity Cache, VALU issue under divergence"
    else:
        bound = "unknown: no committed PMC pass for this kernel and workload"
    tpath, tprov, tstale, twhy = traffic_provenance(root)
    # what binds the timed region, first, so that nobody takes `frac` below for an HBM figure: the chip's VALU issue slots, of which
    # the step's traversal + shading launches use `frac`, at `lane_utilisation` live lanes per issued vector instruction
    binding = ({"roof": "valu_issue", "frac": t.get("valu_issue_frac_of_step"), "lane_utilisation": t.get("valu_lane_utilisation"),
                "hbm_frac": t.get("hbm_frac")} if not tstale and t.get("valu_issue_frac_of_step") is not None else
               {"roof": "valu_issue (DESIGN.md 5.2; the counters that say so are not quoted: traffic_stale)" if tstale else "unknown: no PMC pass for this workload",
                "frac": None, "lane_utilisation": None, "hbm_frac": None})
    out = {"binding": binding, "traffic_commit": tprov.get("commit"), "traffic_stale": bool(tstale), "traffic_stale_why": twhy,
           "traffic_file": ("profiles/" + os.path.basename(tpath)) if tpath else None,
           "bound": bound, "hbm_frac": t.get("hbm_frac"), "valu_issue_frac_of_step": t.get("valu_issue_frac_of_step"),
           "valu_issue_utilisation": t.get("valu_issue_utilisation"),
           "timed_schedule_valu_lane_utilisation": t.get("valu_lane_utilisation"),
           "achieved": t["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": t["achieved"] / HBM_PEAK_GBS,
           "traffic": traffic}
    step_gbs = bytes_per_step / (ms_per_step * 1e-3) / 1e9
    if step_gbs > HBM_PEAK_GBS or out["frac"] > 1.0:
        out["frac_note"] = ("SURVEY 8(d)'s algorithmic bytes per step over ms_per_step = %.0f GB/s, above the %.0f GB/s peak: the figure "
                            "prices a node visit at 64 B where the kernel fetches one 32-B record, most of them from L2 / Infinity "
                            "Cache. `frac` is kept for continuity and is not an HBM utilisation; hbm_frac (PMC counters) is" % (
                                step_gbs, HBM_PEAK_GBS))
    out.update({k: v for k, v in t.items() if k not in out})
    out["own_record_frac_of_l2_peak"] = t["own_record_gbs"] / L2_PEAK_GBS
    out["own_record_frac_of_infinity_cache_gather"] = t["own_record_gbs"] / MALL_GATHER_GBS
    out["measured_copy_gbs"] = copy_gbs
    out["rank"] = 0
    out["single_launch_kernel_alone"] = whole
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this one, which
    has not touched (and never touches) a GPU, with the rendezvous environment torch.distributed.run would have
    set; the children's stdout is this process's, so rank 0's JSON line is the only output. Returns the worst exit
    code; if a rank fails the others are ended (by the PIDs started here)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, live = 0, list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    return worst


def dry_run(world):
    """Every rank reaches its communicator (gloo, no GPU) and the group proves itself with the collectives the real line's
    multi-rank fields come from (all-reduce, all-gather of per-rank figures); rank 0 prints those fields as the real line names them."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    comm = pdist.Comm(world, backend="gloo")
    total = comm.sum_int(1)
    rows = comm.gather_floats([float(comm.rank), 0.0, 0.0, 0.0, -1.0])
    comm.barrier()
    print("bench.py dry run: rank %d of %d reached its communicator; all-reduce over the group = %d" % (comm.rank, world, total),
          flush=True)
    if comm.rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "value": None,
                          "ranks": {"world": world, "transport": "gloo (dry run: no GPU, no RCCL)", "comm_ranks": total,
                                    "rays_traced_per_rank": [0] * len(rows), "ranks_heard_from": [int(r[0]) for r in rows]},
                          "rccl_ranks": None, "sharded_check": "not run: dry run", "sharded_check_detail": None}), flush=True)
    comm.close()
    return 0 if total == world and [int(r[0]) for r in rows] == list(range(world)) else 1


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # nothing below runs in the parent: it never initialises a GPU
    if args.dry_run:
        sys.exit(dry_run(int(os.environ.get("WORLD_SIZE", str(args.gpus))) if args.gpus > 1 else 1))
    os.environ["NCCL_DEBUG"] = os.environ.get("PSM_NCCL_DEBUG", "WARN")  # keep RCCL's version banner off stdout
    # one hardware queue per frame in flight + the accumulating stream + the communicator's (the runtime's default of 4 makes
    # streams share a queue): 8 for the 4 lanes of one GPU, 16 for the 12 lanes of a tile; read by the HIP runtime when it initialises
    # (a rehearsal of several ranks on ONE shared GPU keeps 8: beyond ~20 queues per GPU the runtime's scheduling collapses -- 24 queues
    # under 16 lanes 2.4 times slower, two ranks with 16 each twelve times, profiles/r04_tile_emulation.txt)
    shared_gpu = os.environ.get("PSM_DIST_TRANSPORT", "rccl") == "hoststaged"
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if ((args.gpus > 1 or args.force_dist or args.emulate_tile) and not shared_gpu) else "8")
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus))) if args.gpus > 1 else 1
    psm = importlib.import_module("prismarine-core_amd")
    pdist = importlib.import_module("prismarine-core_amd.dist")
    dist = pdist.Comm(world, force=args.force_dist)
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    scene = {"sponza_like": scenes.sponza_like, "cornell": scenes.cornell, "stress": scenes.stress}[args.scene]()
    obj_note = "generator arrays fed to loadTriangles directly"
    if not args.no_obj_roundtrip and args.scene != "stress":
        # north_star: synthetic OBJ scenes. The generator's scene goes through the OBJ writer and reader once (outside every
        # timed region): what is rendered is what the OBJ file holds (v / vn / f with %.9g floats: the arrays round-trip exactly)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, args.scene + ".obj")
            scenes.write_obj(path, scene)
            back = scenes.read_obj(path)
            obj_bytes = os.path.getsize(path)
        assert back["tris"].shape == scene["tris"].shape and np.array_equal(back["tris"], scene["tris"]), "OBJ round trip changed the triangles"
        assert np.array_equal(back["mats"], scene["mats"]), "OBJ round trip changed the material ids"
        assert len(back["materials"]) == len(scene["materials"]) and np.abs(back["normals"] - scene["normals"]).max() < 1e-6
        # normals: the reader normalises what it reads (loader.comp:119-128), one ulp off the generator's here and there;
        # materials: Kd / Ks / Ke carry 6 decimals in the .mtl, the generator's are kept
        scene = dict(scene, tris=back["tris"], normals=back["normals"], mats=back["mats"])
        obj_note = "scene written by scenes.write_obj and read back by scenes.read_obj (%d bytes of OBJ)" % obj_bytes
    elif args.scene == "stress":
        obj_note += " (a 10 M-triangle OBJ is 1.5 GB of text: the stress scene skips the file)"
    if args.textured:
        scene = scenes.textured(scene)
    try:
        R = Renderer(psm, scenes, scene, args, dist)
    except pdist.NativeUnavailable as e:
        # every rank gets this together (Comm.attach_native agrees on it). No silent fall-back: a scaling run that is green
        # on torch.distributed collectives would say nothing about the path SURVEY 8(e) describes. PSM_DIST_NATIVE=0 asks
        # for the torch.distributed collectives explicitly.
        if dist.rank == 0:
            print("bench: the C ABI's RCCL communicator is unavailable (%s); set PSM_DIST_NATIVE=0 to run on torch.distributed "
                  "collectives instead" % e, file=sys.stderr)
        dist.close()
        sys.exit(3)
    ctx = R.ctx
    check = None
    if dist.active and dist.world > 1 and R.native is not None:
        check = sharded_self_check(psm, scenes, dist, R)   # exits non-zero (every rank) unless sharded == unsharded

    def run_steps(k):
        if dist.active:
            return R.frames_in_flight_sharded(k)
        return sum(r for _, r in R.frames_in_flight(k))

    def reseed():
        R.batch.setSeed(1000)
        R._master_state = 1000
        R.rt.clearSampler()
        R.batch.clearSampler()

    # setup: one frame on every lane (first use of a lane allocates its continuation queues and loads code),
    # then the untimed warm-up steps on the timed path
    run_steps(R.lanes)
    run_steps(max(args.warmup, 0))
    R.batch.sync()

    # Frames in flight per rank, settled by measurement: 12 lanes on 16 hardware queues won every one-GPU tile
    # emulation (profiles/r04_tile_emulation.txt), but no multi-GPU run has confirmed it -- RCCL's own streams share the queues there.
    # A sharded run with the default lane count therefore times a short untimed stretch with all its lanes and with two thirds of
    # them (12 -> 8), every rank the same stretches, and keeps the faster (the maximum over the ranks decides, as in the timed region).
    lane_cal = None
    if dist.active and world > 1 and args.lanes == 0 and R.native is not None and os.environ.get("PSM_DIST_PIPELINE", "1") != "0" and R.lanes >= 4:
        alt = max(2, (R.lanes * 2 // 3) & ~1)
        kc = 2 * R.lanes

        def stretch(n):
            R.active_lanes = n
            dist.barrier()
            R.batch.sync()
            c0 = time.perf_counter()
            run_steps(kc)
            R.batch.sync()
            dist.barrier()
            return dist.max_float(time.perf_counter() - c0) / kc * 1e3
        full, fewer = stretch(R.lanes), stretch(alt)
        R.active_lanes = R.lanes if full <= fewer * 1.01 else alt
        lane_cal = {"what": "untimed stretches of %d steps before the timed region, ms per step, maximum over the ranks" % kc,
                    "lanes_%d" % R.lanes: full, "lanes_%d" % alt: fewer, "chosen": R.active_lanes}

    # counting pass: the frames of the timed region (same rand() streams), one after another on lane 0,
    # counters on: V, T, R are deterministic per seed; logged per bounce round
    reseed()
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    ray_sets = [] if (dist.rank == 0 and not args.no_cpu_baseline) else None
    round_log = []
    for i in range(args.steps):
        R.frame(record=ray_sets if i == 0 else None, round_log=round_log)
    cnt = ctx.stats()
    V, T, Rr = cnt.node_visits, cnt.tri_tests, cnt.rays_traced
    assert sum(r[1] for r in round_log) == V and sum(r[2] for r in round_log) == T and sum(r[0] for r in round_log) == Rr

    # kernel pass, serial: the same frames again, one after another on lane 0, HIP events on every launch (on the launching
    # stream): per-stage times, and the single-launch traversal kernel running alone
    reseed()
    R.batch.sync()
    ctx.stats_enable(True, False)
    ctx.stats_reset()
    for i in range(args.steps):
        R.frame()
    ctx.sync()
    kst = ctx.stats()
    assert kst.rays_traced == Rr, (kst.rays_traced, Rr)
    ctx.stats_enable(False, False)

    # kernel pass, as timed: the call of the timed region with HIP events around every traversal launch on every lane's own
    # stream (psm_stats_enable(2, 0): traversal launches only, so the rebuild keeps its captured graph and the schedule is
    # the timed one). Prices the kernel the timed region actually launches.
    reseed()
    tctx = list(R.batch.contexts())        # every context that traces in the timed region (all parts of all frames in flight)
    for c_ in tctx:
        c_.stats_enable(2, False)
        c_.stats_reset()
        c_.stats_reference(tctx[0])        # one time axis for all their launches
    dist.barrier()
    R.batch.sync()
    e0 = time.perf_counter()
    traced_ev = run_steps(args.steps)
    R.batch.sync()
    ev_elapsed = time.perf_counter() - e0
    assert traced_ev == Rr, (traced_ev, Rr)
    lane_stats = [c_.stats() for c_ in tctx]
    spans = sorted(iv for c_ in tctx for iv in c_.traverse_intervals())
    busy_ms, edge = 0.0, -1e30        # union of the traversal launches' intervals: time with at least one of them on the chip
    for a, b in spans:
        if b > edge:
            busy_ms += b - max(a, edge)
            edge = b
    for c_ in tctx:
        c_.stats_enable(False, False)
    ho_launches = sum(s_.handover_launches for s_ in lane_stats)
    ho_ms = sum(s_.handover_ms for s_ in lane_stats)
    wh_launches = sum(s_.traverse_launches for s_ in lane_stats) - ho_launches
    wh_ms = sum(s_.traverse_ms for s_ in lane_stats) - ho_ms

    # timed region: exactly K steps between barrier + synchronize on both sides, the maximum over the ranks -- taken
    # `repeats` times over (the same frames: reseeded), `value` and `ms_per_step` from the MEDIAN call, every call's figure
    # in the line's "timing" object (median and minimum over repeats; one wall-clock sample of a 45 ms call
    # moves by 1-2 % with whatever else the box does)
    samples = []
    for rep in range(max(args.repeats, 1)):
        reseed()
        ctx.stats_reset()
        dist.barrier()
        R.batch.sync()
        t0 = time.perf_counter()
        traced = run_steps(args.steps)
        R.batch.sync()
        dist.barrier()
        samples.append(dist.max_float(time.perf_counter() - t0))
        assert traced == Rr, (traced, Rr)
    elapsed = sorted(samples)[(len(samples) - 1) // 2]   # the median call (the lower one of an even count)
    st = kst
    total_rays = dist.sum_int(int(Rr))

    if args.diag_clock and not dist.active:
        reseed()
        for ln in R.batch.lanes:
            ln.ctx.stats_enable(False, True)
            ln.ctx.stats_reset()
        R.batch.sync()
        d0 = time.perf_counter()
        run_steps(args.steps)
        R.batch.sync()
        d_el = time.perf_counter() - d0
        tick = real = wsteps = waves = 0
        for ln in R.batch.lanes:
            s_ = ln.ctx.stats()
            tick, real, wsteps, waves = tick + s_.wave_clock_ticks, real + s_.wave_real_ticks, wsteps + s_.wave_steps, waves + s_.waves
            ln.ctx.stats_enable(False, False)
        sys.stderr.write("diag-clock: %d frames in %.3f ms (counting kernels); traversal waves %d, wave-steps %d (%.2f G/s), "
                         "mean wave life %.1f us, shader clock while they ran %.3f GHz, resident traversal waves on average %.0f\n" % (
                             args.steps, d_el * 1e3, waves, wsteps, wsteps / d_el / 1e9, real / max(waves, 1) / 100.0,
                             tick / max(real, 1) * 0.1, real / 100e6 / d_el))
        reseed()

    per_rank = None
    if dist.active:
        # what every rank did, to rank 0's line: a rank that idles or a tile that costs twice the others' shows here
        gather_ms = -1.0
        if R.native is not None:   # one frame's gather, timed on its own: 5 x (pack, gather, unpack on rank 0), drained by a barrier on the communicator
            dist.barrier()
            g0 = time.perf_counter()
            for _ in range(5):
                R._gather()
            R.native.barrier()
            ctx.sync()
            gather_ms = (time.perf_counter() - g0) / 5 * 1e3
        per_rank = dist.gather_floats([float(Rr), kst.build_ms / args.steps, kst.traverse_ms / args.steps, kst.shade_ms / args.steps, gather_ms])

    if dist.rank == 0:
        img = R.batch.snapHdr()
        copy_gbs = ctx.copy_bandwidth(1 << 30, 5)  # the box's achievable ceiling next to the vendor peak
        # which rounds of the timed region ran the hand-over kernel: plan_traverse's rule (csrc/trace.hip) -- frames in flight
        # (or a forced hand-over schedule) and at least phase_min_rays rays in the round
        min_rays = 1 << 19
        if args.trav_adaptive and len(args.trav_adaptive.split(",")) >= 5:
            min_rays = int(args.trav_adaptive.split(",")[4])
        hand = args.traverse in ("phased", "adaptive") or (args.traverse == "auto" and R.lanes > 1)
        ho_rounds = [r for r in round_log if hand and r[0] >= min_rays]
        wh_rounds = [r for r in round_log if not (hand and r[0] >= min_rays)]
        assert (ho_launches > 0) == (len(ho_rounds) > 0), (ho_launches, len(ho_rounds))
        ms_step = elapsed / args.steps * 1e3
        how = ("the call of the timed region repeated with HIP events around every traversal launch on every lane's stream "
               "(%d frame(s) in flight: a launch shares the chip with the other frames' kernels); that pass ran at %.3f ms per "
               "step against %.3f timed" % (R.lanes, ev_elapsed / args.steps * 1e3, ms_step))
        sums = lambda rs: (sum(r[0] for r in rs), sum(r[1] for r in rs), sum(r[2] for r in rs))
        whole_frame = not dist.active and not args.emulate_tile
        if ho_rounds:
            timed_k = price("rt_traverse<false, false, true>", ho_launches, ho_ms, *sums(ho_rounds), len(ho_rounds), args.steps,
                            args.scene, args.width, args.height, how, whole_frame)
            if wh_launches:
                timed_k["rounds_below_min_rays_run_single_launch"] = price(
                    "rt_traverse<false, false, false>", wh_launches, wh_ms, *sums(wh_rounds), len(wh_rounds), args.steps,
                    args.scene, args.width, args.height, how, whole_frame)
        else:
            timed_k = price("rt_traverse<false, false, false>", wh_launches, wh_ms, *sums(wh_rounds), len(wh_rounds), args.steps,
                            args.scene, args.width, args.height, how, whole_frame)
        whole_k = price("rt_traverse<false, false, false>", st.traverse_launches, st.traverse_ms, Rr, V, T, len(round_log), args.steps,
                        args.scene, args.width, args.height,
                        "serial kernel pass: the same frames one after another on one stream, every launch alone on the chip", whole_frame)
        # launches of different frames overlap: their summed durations exceed the wall time; the union of their intervals is
        # the time during which traversal is on the chip at all, and the ratio is how many run side by side on average
        timed_k["traversal_busy_ms_per_step"] = busy_ms / args.steps
        timed_k["launches_side_by_side"] = (ho_ms + wh_ms) / busy_ms if busy_ms > 0 else None
        timed_k["achieved_over_busy_time"] = (Rr * 44 + V * 64 + T * 36) / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else None
        if timed_k.get("traffic") and busy_ms > 0:
            # chip level: the HBM-side bytes of all traversal launches of a step over the time traversal is on the chip
            tot = timed_k["traffic"] * timed_k["launches"]
            small = timed_k.get("rounds_below_min_rays_run_single_launch")
            if small and small.get("traffic"):
                tot += small["traffic"] * small["launches"] * (small["algorithmic_bytes_per_launch"] / max(whole_k["algorithmic_bytes_per_launch"], 1.0))
            timed_k["hbm_gbs"] = tot / (busy_ms * 1e-3) / 1e9
            timed_k["hbm_frac"] = timed_k["hbm_gbs"] / HBM_PEAK_GBS
            timed_k["hbm_frac_note"] = ("HBM-side bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE per launch, committed passes) of all traversal "
                                        "launches of a step over traversal_busy_ms_per_step, the time at least one of them is on the chip")
        if whole_frame and timed_k.get("valu_insts_per_launch"):
            # How much of the chip's VALU issue capacity the timed step uses: wave-instructions of the traversal and shading
            # launches of a step (PMC: SQ_INSTS_VALU per launch of the committed passes; the small rounds' single launches scaled by
            # their rays) x 4 cycles per wave64 instruction per SIMD (the step's mix issues at 3.9-4.3, tools/ubench/valu_rate)
            # over 1024 SIMDs x 2.4 GHz x ms_per_step. This -- not HBM -- is the roof the path runs under.
            insts = timed_k["valu_insts_per_launch"] * timed_k["launches"]
            small = timed_k.get("rounds_below_min_rays_run_single_launch")
            if small and whole_k.get("valu_insts_per_launch") and whole_k["R"]:
                insts += whole_k["valu_insts_per_launch"] * whole_k["launches"] * (small["R"] / whole_k["R"])
            e_sh, _ = pmc_entry(args.scene, args.width, args.height, "rt_shade<false, false, false>")   # TEX, MULTI, BOTH: the texture-less, single-hierarchy, one-lobe instantiation the bench scenes run
            sh = (e_sh or {}).get("sq_per_launch", {}).get("SQ_INSTS_VALU")
            if sh:
                insts += sh * len(round_log)
            timed_k["valu_wave_insts_per_step"] = insts / args.steps
            timed_k["valu_issue_frac_of_step"] = insts / args.steps * VALU_QUAD_CYCLES / (SIMDS * 2.4e9 * ms_step * 1e-3)
            timed_k["valu_issue_frac_of_step_covers"] = "traversal + shading launches" if sh else "traversal launches"
        roof = roofline(timed_k, whole_k, (Rr * 44 + V * 64 + T * 36) / args.steps, ms_step, copy_gbs)  # rank 0's launches
        out = {
            "metric": "Mrays/sec + ms/frame, Sponza 1920x1080 4spp",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "S-%s %d tris, %dx%d, 1 spp per step (4 steps = 4 spp), full HLBVH rebuild "
                                   "per frame + camera + <=%d bounce rounds + sample; %d frame(s) in flight per GPU" % (
                                       args.scene.replace("_", "-"), scene["tris"].shape[0], args.width, args.height, args.depth, R.active_lanes if dist.active else R.lanes),
                       "scene": args.scene + ("+tex" if args.textured else ""), "width": args.width, "height": args.height,
                       "parallelism": "tile%d" % world, "frames_in_flight": R.active_lanes if dist.active else R.lanes, "input": obj_note,
                       "band_weights": R.weights,
                       "collectives": ("none" if not dist.active else
                                       "psm_dist_* (transport %s%s)" % (R.native.transport, ": a REHEARSAL, the ranks share a GPU and exchange through host memory -- not a scaling measurement" if R.hoststaged else " from libpsm_hip.so")
                                       if R.native is not None else "torch.distributed " + dist.backend)},
            "rays_per_frame": total_rays / args.steps,
            "traverse_mrays_s": (Rr / (st.traverse_ms * 1e-3) / 1e6) if st.traverse_ms > 0 else None,
            "stage_ms_per_frame_measured": ("serial kernel pass: the same frames one after another on one stream (with frames in "
                                            "flight the stages of different frames overlap, so these do not add up to ms_per_step)"),
            "stage_ms_per_frame": {"build": st.build_ms / args.steps, "bounds": st.bounds_ms / args.steps,
                                   "morton": st.morton_ms / args.steps, "sort": st.sort_ms / args.steps,
                                   "emit_refit": st.emit_ms / args.steps,
                                   "camera": st.camera_ms / args.steps, "traverse": st.traverse_ms / args.steps,
                                   "shade": st.shade_ms / args.steps, "sample": st.sample_ms / args.steps},
            "timing": {"repeats": len(samples), "what": "the timed call of `steps` steps taken this many times over the same frames; value and ms_per_step are the median call's",
                       "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_min": min(samples) / args.steps * 1e3,
                       "ms_per_step_max": max(samples) / args.steps * 1e3, "ms_per_step_all": [t_ / args.steps * 1e3 for t_ in samples]},
            "roofline": roof,
            "image_mean": float(img[..., :3].mean()),
        }
        if dist.active:
            mm = lambda col: {"min": min(r[col] for r in per_rank), "max": max(r[col] for r in per_rank)}
            out["ranks"] = {
                "world": world, "transport": R.native.transport if R.native is not None else "torch.distributed " + dist.backend,
                "comm_ranks": R.native.comm_ranks if R.native is not None else None,
                "frames_in_flight_per_rank": R.active_lanes, "lane_calibration": lane_cal, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "rays_traced_per_rank": [int(r[0]) for r in per_rank],
                "per_frame_ms_over_ranks": {"what": "serial kernel pass of every rank (its tile's frames one after another); gather: pack + gather + rank 0's unpack, 5 in a row",
                                            "build": mm(1), "traverse": mm(2), "shade": mm(3), "gather": mm(4) if R.native is not None else None}}
            # the communicator's own count (ncclCommCount) next to n_gpus: what proves that N ranks took part
            out["rccl_ranks"] = R.native.comm_ranks if (R.native is not None and R.native.transport == "rccl") else None
            out["sharded_check"] = check["result"] if check else ("not run: one rank" if world == 1 else "not run: no native communicator")
            out["sharded_check_detail"] = check
        if dist.active and R.hoststaged:
            # ranks that share one GPU and exchange through host memory: a rehearsal of the scheduler, never a scaling
            # measurement -- the line says so at the top level and carries no value a driver could ingest as one
            out["rehearsal"] = True
            out["rehearsal_value_mrays_s"] = out["value"]
            out["value"] = None
            out["vs_baseline"] = None
        if ray_sets is not None and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, ray_sets, args)
        print(json.dumps(out))
    dist.close()


if __name__ == "__main__":
    main()
