"""The native sharded scheduler against REAL peers on one GPU.

RCCL refuses two ranks on one device, so the ranks here are child processes that share GPU 0 and exchange through the
C ABI's host-staged transport (psm_dist_connect_hoststaged, csrc/transport_shm.hip) -- everything else is what a rank
of an 8-GPU run executes: psm_dist_render_frames / psm_dist_render_batch with their two alternating lane groups, lanes
that park on their LOCAL counts in different rounds on different ranks, psm_dist_decide on the all-gathered answers,
forced catch-up rounds, padded tiles of ranks with fewer bands, rt_unpack_all fed with real peers' tiles,
rt_camera_rest on the gathering rank, the fold in frame order. Only the ncclGather / ncclAllGather calls themselves are
replaced (by the same table of two functions RCCL sits behind).

Bar: rank 0's accumulated image equals the unsharded psm_lanes_render image of the same frames -- deposit counts
exactly, radiance to float-atomic order (1e-5) -- every rank returns, and the frames ran the same number of rounds.
What forces the exchange: Include/Prismarine/Pipeline.inl:459-461 (`getRayCount() < 32`); what makes tiles independent:
ShadersSDK/raytracing/sampler.comp:53-66, include/rayslib.glsl:148.
"""
import importlib
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEER = os.path.join(ROOT, "tests", "dist_peer.py")


def run_group(tmp_path, world, timeout=420, **cfg):
    """Start `world` ranks (fresh processes sharing GPU 0), wait for all of them; returns their reports."""
    shm = "/psm-test-%s" % uuid.uuid4().hex[:12]
    out = str(tmp_path)
    procs = []
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    try:
        for r in range(world):
            c = dict(cfg, rank=r, world=world, shm=shm, out=out)
            procs.append(subprocess.Popen([sys.executable, PEER, json.dumps(c)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()       # by the PIDs started here
                pytest.fail("a rank did not return within %d s: the sharded scheduler hangs" % timeout)
            logs.append(o.decode(errors="replace"))
        for r, p in enumerate(procs):
            assert p.returncode == 0, "rank %d exited with %s:\n%s" % (r, p.returncode, logs[r][-3000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm" + shm)
        except OSError:
            pass
    return [json.load(open(os.path.join(out, "rank%d.json" % r))) for r in range(world)]


def unsharded(psm, scenes, cfg):
    """The same frames through psm_lanes_render on this process's GPU contexts: image and rounds per frame."""
    peer = importlib.import_module("dist_peer")
    scene = peer.make_scene(scenes, cfg)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    b = psm.FrameBatch(cfg["lanes"], cfg["w"], cfg["h"], seed=cfg["seed"])
    b.allocate(scene["tris"].shape[0])
    b.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    b.applyMaterials(ms)
    res = b.render(cfg["frames"], scene["eye"], scene["view"], depth=cfg.get("depth", 16))
    img = b.snapHdr()
    b.close()
    return img, [r for r, _ in res], scene


def parks(psm, scenes, scene, cfg, rank, world):
    """(rounds, local count) at which rank's tile of frame 0 parks when it runs alone (psm_lanes_run_sharded)."""
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    b = psm.FrameBatch(1, cfg["w"], cfg["h"], seed=cfg["seed"])
    b.allocate(scene["tris"].shape[0])
    b.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    b.applyMaterials(ms)
    b.each(lambda r: r.setTileInterleaved(rank, world, cfg.get("weights")))
    cam = scenes.camera_matrices(scene["eye"], scene["view"], cfg["w"], cfg["h"])
    rounds, counts = b.run_sharded(b.frame_seeds(1), cam[0], cam[1], depth=cfg.get("depth", 16))
    b.close()
    return rounds[0], counts[0]


def check_equal(psm, scenes, tmp_path, world, lit=True, **cfg):
    want, want_rounds, scene = unsharded(psm, scenes, cfg)
    reps = run_group(tmp_path, world, **cfg)
    for r in reps:
        assert r["rc"] == 0, "rank %d: %s" % (r["rank"], r["error"])
        assert r["rounds"] == want_rounds, (r["rank"], r["rounds"], want_rounds)   # the GLOBAL stop rule: as unsharded
    got = np.load(os.path.join(str(tmp_path), "image.npy"))
    assert np.array_equal(got[..., 3], want[..., 3])                           # sample weights: exact
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)
    assert not lit or want[..., :3].max() > 0.05
    return scene, want_rounds


@pytest.mark.parametrize("mode", ["frames", "batch"])
def test_world2_sharded_frames_equal_unsharded(psm, scenes, tmp_path, mode):
    """Two ranks, Sponza-class scene, 160x90 (12 bands, the last one 2 rows: rank 1 sends a padded tile), 4 lanes in two
    groups, 7 frames (a short last batch)."""
    check_equal(psm, scenes, tmp_path, 2, scene="sponza_small", w=160, h=90, lanes=4, frames=7, seed=77, mode=mode)


def test_world3_tiles_that_die_rounds_before_the_frame(psm, scenes, tmp_path):
    """Three ranks, 192x48 = six bands, two each: the box sits in the two middle bands (ranks 2 and 0), the bands of
    rank 1 see sky only, so rank 1 runs dry after the first round while the others' rays bounce on -- it parks early,
    learns from the exchange that the frame goes on and catches up with empty forced rounds (drawing its rand() in
    step). 3 lanes = groups of 1 and 2."""
    cfg = dict(scene="cornell_far", w=192, h=48, lanes=3, frames=5, seed=5, mode="frames")
    scene, rounds = check_equal(psm, scenes, tmp_path, 3, **cfg)
    alone = [parks(psm, scenes, scene, cfg, r, 3) for r in range(3)]
    assert alone[1][0] < alone[0][0] and alone[1][0] < alone[2][0], alone    # the scenario the test is about
    assert max(a[0] for a in alone) <= rounds[0], (alone, rounds)


@pytest.mark.parametrize("mode", ["frames", "batch"])
def test_world3_unequal_band_counts(psm, scenes, tmp_path, mode):
    """Three ranks on 160x90: 12 bands deal 4 / 4 / 4 but the last band has 2 rows, and at 128x72 (9 bands) ranks own
    3 / 3 / 3; 100x52 gives 7 bands = 3 / 2 / 2 with a 4-row last band: unequal tiles, padded sends."""
    check_equal(psm, scenes, tmp_path, 3, scene="sponza_small", w=100, h=52, lanes=2, frames=4, seed=9, mode=mode)


@pytest.mark.parametrize("world,weights,w,h", [(3, [1, 2, 2], 160, 90), (4, "default", 96, 200)], ids=["3ranks-1:2:2", "4ranks-5:6:6:6"])
def test_weighted_band_dealing_equals_unsharded(psm, scenes, tmp_path, world, weights, w, h):
    """The gathering rank owns fewer bands than the workers (psm_rt_set_tile_weighted + psm_dist_set_band_weights: it also
    unpacks, fills and samples the whole image): tiles of different sizes, every rank sends the largest tile's size, the
    unpack finds each texel's owner through the dealing. 4 ranks: bench.py's default dealing (periods of 23 bands, 25
    bands here)."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    if weights == "default":
        weights = pdist.default_band_weights(world)
        assert weights == [5, 6, 6, 6]
    shares = [pdist.owned_texels(r, world, w, h, weights) for r in range(world)]
    assert sum(shares) == w * h and shares[0] < min(shares[1:])
    check_equal(psm, scenes, tmp_path, world, scene="sponza_small", w=w, h=h, lanes=4, frames=6, seed=21, mode="frames", weights=weights)


def test_c4_sponza_1080p_16_frames_tile_sharded_over_4_ranks(psm, scenes, tmp_path):
    """BASELINE C4 -- S-sponza-like (262 267 triangles), 1920x1080, 16 frames (16 spp), tile-sharded -- through
    psm_dist_render_frames with bench.py's default dealing (5 : 6 : 6 : 6 of every 23 bands), full rebuild per frame on every
    rank. World 4, not 8: a GPU box lets 6 processes use its card at once (this one holds the unsharded reference), and the
    ranks share one GPU through the host-staged transport -- the scheduler, the dealing, the padded tiles, rt_unpack_all and the
    fold at the metric's full size, not the xGMI gather. Rank 0's image equals psm_lanes_render's: deposit counts exactly,
    radiance to 1e-5; every rank reports the unsharded rounds per frame (Pipeline.inl:459-461 on the GLOBAL count)."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    weights = pdist.default_band_weights(4)
    assert weights == [5, 6, 6, 6]
    check_equal(psm, scenes, tmp_path, 4, scene="sponza", w=1920, h=1080, lanes=4, frames=16, seed=1000, mode="frames", weights=weights,
                timeout_ms=180000)


def test_c5_stress_2160p_tile_sharded_over_2_ranks(psm, scenes, tmp_path):
    """BASELINE C5's scene, resolution and sample count sharded: S-stress (9 999 616 triangles, rebuilt per frame on every rank),
    3840x2160, 8 frames (C5's 8 spp) on 2 lanes, world 2 with the default 11 : 12 dealing (the box lets six processes use its
    card and the ten-million-triangle scene is held three times over as it is; world 8 of the dealing and the protocol is the
    gloo test of tests/test_dist_cpu.py). Same bar as above."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    weights = pdist.default_band_weights(2)
    assert weights == [11, 12]
    check_equal(psm, scenes, tmp_path, 2, scene="stress", w=3840, h=2160, lanes=2, frames=8, seed=1000, mode="frames", weights=weights,
                timeout_ms=300000)


@pytest.mark.parametrize("seed", [3, 5, 11, 39])   # (soups without coplanar stacks: a full chain pool cuts chains by timing, test_gpu_fuzz.py)
def test_fuzzed_group_of_real_peers(psm, scenes, tmp_path, seed):
    """The fuzzer's scenes (tests/test_gpu_fuzz.py::fuzz_scene) rendered by two to four real processes under random conditions: lanes,
    frames, depth, batch or pipelined frames, band weights with ranks that own nothing. Rank 0's image and every rank's round counts
    as the unsharded render's."""
    rng = np.random.RandomState(45000 + seed)
    world = int(rng.randint(2, 5))
    weights = None if rng.rand() < 0.5 else [int(x) for x in rng.randint(0, 3, world)]
    if weights is not None and sum(weights) == 0:
        weights[0] = 1
    cfg = dict(scene="fuzz:%d" % seed, w=int(rng.randint(24, 97)), h=int(rng.randint(17, 65)), lanes=int(rng.randint(1, 5)), frames=int(rng.randint(1, 7)),
               seed=seed, mode=str(rng.choice(["frames", "batch"])), depth=int(rng.choice([2, 4, 16])))
    if weights is not None:
        cfg["weights"] = weights
    check_equal(psm, scenes, tmp_path, world, lit=False, **cfg)


def test_world3_rank_without_a_band(psm, scenes, tmp_path):
    """16 rows = 2 bands on 3 ranks: rank 2 owns nothing, traces nothing and still keeps every collective."""
    check_equal(psm, scenes, tmp_path, 3, scene="cornell_open", w=64, h=16, lanes=2, frames=3, seed=3, mode="frames")


@pytest.mark.parametrize("fail,mode", [("gather", "frames"), ("build", "frames"), ("gather", "batch"), ("build", "batch")])
def test_one_ranks_failure_is_every_ranks_error_return(psm, scenes, tmp_path, fail, mode):
    """A rank whose own work fails -- a rebuild without triangles (before the first exchange), a gather its Pipeline's
    tile does not allow (after the last decision) -- keeps the collective sequence, so that EVERY rank returns an error
    from the same call instead of waiting inside a collective: the failing rank its own, the others PSM_ERR_PEER."""
    reps = run_group(tmp_path, 2, timeout=120, scene="cornell_open", w=64, h=48, lanes=2, frames=4, seed=3, mode=mode,
                     fail=fail, fail_rank=1, timeout_ms=20000)
    assert reps[0]["rc"] == 1 and reps[1]["rc"] == 1, reps
    assert "(-6)" in reps[0]["error"], reps[0]            # PSM_ERR_PEER on the healthy rank
    assert "(-6)" not in reps[1]["error"], reps[1]        # its own error on the failing one


def test_missing_peer_times_out_instead_of_hanging(psm, ctx):
    """A rank whose peer never arrives gets PSM_ERR_PEER from the connect after the timeout."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    nd = pdist.NativeDist(ctx, 0, 2)
    name = "/psm-test-%s" % uuid.uuid4().hex[:12]
    try:
        with pytest.raises(psm.PsmError, match="attached"):
            nd.connect_hoststaged(name, 4096, timeout_ms=500)
    finally:
        nd.close()
        assert not os.path.exists("/dev/shm" + name)


def _bench_ranks(tmp_path, world, extra_env=None, timeout=420):
    """bench.py --gpus N the way a user starts it (it spawns its ranks itself), the ranks sharing GPU 0 over the host-staged
    transport: a rehearsal line (value null) -- here for its self-check and its multi-rank fields."""
    env = dict(os.environ, PSM_DIST_TRANSPORT="hoststaged", PSM_DIST_BACKEND="gloo", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--scene", "cornell", "--width", "320", "--height", "184",
                           "--steps", "4", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-obj-roundtrip"],
                          env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_proves_sharded_equals_unsharded_before_it_times(tmp_path):
    """bench.py on 2 ranks: before the warm-up it renders a small frame set through psm_dist_render_frames on the run's own
    communicator and dealing and the same frames unsharded on rank 0 (sharded_self_check); the line then says how many ranks
    the transport counted, what every rank traced, and sharded_check = ok."""
    out = _bench_ranks(tmp_path, 2)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["sharded_check"] == "ok" and line["sharded_check_detail"]["largest_difference"] < 1e-5
    rk = line["ranks"]
    assert rk["world"] == 2 and rk["comm_ranks"] == 2 and rk["transport"] == "host-staged" and line["rccl_ranks"] is None
    assert len(rk["rays_traced_per_rank"]) == 2 and min(rk["rays_traced_per_rank"]) > 0
    assert sum(rk["rays_traced_per_rank"]) == round(line["rays_per_frame"] * line["steps"])
    assert rk["per_frame_ms_over_ranks"]["traverse"]["min"] > 0 and rk["per_frame_ms_over_ranks"]["gather"]["max"] > 0
    cal = rk["lane_calibration"]      # the default lane count is settled by measurement: all lanes against two thirds of them
    assert cal["chosen"] == rk["frames_in_flight_per_rank"] and cal["chosen"] in (8, 4) and cal["lanes_8"] > 0 and cal["lanes_4"] > 0
    assert line["rehearsal"] is True and line["value"] is None      # ranks that share a GPU: never a scaling figure


def test_bench_self_check_fails_every_rank_when_one_renders_another_picture(tmp_path):
    """The same run with rank 1 looking at the check scene from somewhere else (PSM_BENCH_SABOTAGE_RANK): its bands of the
    gathered image are another picture's, the check sees it, nothing is timed, every rank exits with code 4."""
    out = _bench_ranks(tmp_path, 2, {"PSM_BENCH_SABOTAGE_RANK": "1"})
    assert out.returncode == 4, (out.returncode, out.stderr[-2000:])
    assert "sharded self-check FAILED" in out.stderr and not any(l.startswith("{") for l in out.stdout.splitlines())
