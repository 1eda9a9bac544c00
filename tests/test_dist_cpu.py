"""CPU test of the N>1 path: two gloo ranks shard a frame by rows, run the bounce loop in lock step
on the global ray count (prismarine-core_amd.dist.run_rounds + sharded_rounds), gather the per-texel
radiance to rank 0 (Comm.gather_to_root) and reproduce the unsharded frame.

The renderer behind the Pipeline interface is the CPU oracle here (no GPU in this container); the
code under test is the sharding / lock-step / gather plumbing the GPU bench uses unchanged.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePipeline:
    """The subset of psm.Pipeline that sharded_rounds() drives, rendered by the oracle."""

    def __init__(self, O, scenes, scene, w, h, y0, y1, seed):
        self.O, self.scene, self.w, self.h = O, scene, w, h
        self.built = O.build_scene(scene["tris"])
        self.mats = scenes.materials_array(scene["materials"])
        self.cfg = O.make_cfg(w, h, material_count=len(self.mats))
        self.lights = O.default_lights(1)
        self.state = seed
        cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
        t = self._rand()
        self.rays, self.coord, self.tsum, self.flag = O.camera(self.cfg, cam[0], cam[1], t, y0, y1)
        self.raycountCache = self.rays.shape[0]

    def _rand(self):
        v, self.state = self.O.rand_next(self.state)
        return v

    def applyMaterials(self, ms):
        pass

    def intersection(self, obj, force=False):
        if self.raycountCache <= 0:
            return 0
        b = self.built
        self.hits, self.counts, _ = self.O.traverse(b["nodes"], self.scene["tris"], b["M"], self.rays["origin"], self.rays["direct"], 2)
        return 1

    def shade(self, force=False):
        t = self._rand()
        if self.raycountCache <= 0:
            return
        sc = self.scene
        self.rays = self.O.shade(self.cfg, self.lights, self.mats, sc["mats"], sc["tris"], sc["normals"], t, self.rays,
                                 self.hits, self.counts, self.tsum, self.flag)
        self.raycountCache = self.rays.shape[0]

    def reclaim(self):
        pass


def _worker(rank, world, port, w, h, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PSM_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import torch
    pdist = importlib.import_module("prismarine-core_amd.dist")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    from oracle import oracle as O
    comm = pdist.Comm(world, backend="gloo")
    scene = scenes.cornell(open_top=True)
    y0, y1, per = pdist.tile_rows(comm.rank, world, h)
    rt = OraclePipeline(O, scenes, scene, w, h, y0, y1, seed=5)
    rounds = pdist.run_rounds(comm, rt, None, None)
    tile = torch.zeros(per * w * 4, dtype=torch.float32)
    tile[: (y1 - y0) * w * 4] = torch.from_numpy(rt.tsum[y0 * w:y1 * w].reshape(-1).copy())
    allt = comm.gather_to_root(tile)
    total_rounds = comm.sum_int(rounds)
    if comm.rank == 0:
        merged = np.zeros((w * h, 4), np.float32)
        for r in range(world):
            a, b, _ = pdist.tile_rows(r, world, h)
            merged[a * w:b * w] = allt[r * per * w * 4: r * per * w * 4 + (b - a) * w * 4].numpy().reshape(-1, 4)
        np.save(out_path, merged)
        assert total_rounds == world * rounds  # every rank ran the same number of rounds
    comm.close()


def _worker_lanes(rank, world, port, w, h, seeds, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PSM_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import torch
    pdist = importlib.import_module("prismarine-core_amd.dist")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    from oracle import oracle as O
    comm = pdist.Comm(world, backend="gloo")
    scene = scenes.cornell(open_top=True)
    y0, y1, per = pdist.tile_rows(comm.rank, world, h)
    lanes = [OraclePipeline(O, scenes, scene, w, h, y0, y1, seed=sd) for sd in seeds]
    rounds = pdist.run_rounds_lanes(comm, [(rt, None, None) for rt in lanes], depth=16)
    merged = []
    for rt in lanes:  # one gather per frame, in frame order
        tile = torch.zeros(per * w * 4, dtype=torch.float32)
        tile[: (y1 - y0) * w * 4] = torch.from_numpy(rt.tsum[y0 * w:y1 * w].reshape(-1).copy())
        allt = comm.gather_to_root(tile)
        if comm.rank == 0:
            m = np.zeros((w * h, 4), np.float32)
            for r in range(world):
                a, b, _ = pdist.tile_rows(r, world, h)
                m[a * w:b * w] = allt[r * per * w * 4: r * per * w * 4 + (b - a) * w * 4].numpy().reshape(-1, 4)
            merged.append(m)
    if comm.rank == 0:
        np.save(out_path, np.stack(merged))
        np.save(out_path + ".rounds.npy", np.asarray(rounds))
    comm.close()


@pytest.mark.timeout(300)
def test_two_rank_gloo_frames_in_flight_equal_unsharded(tmp_path, oracle, scenes):
    """dist.run_rounds_lanes: two frames in flight per rank, lock step on each frame's own global ray count."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h, seeds = 36, 30, (5, 977)
    out = str(tmp_path / "lanes.npy")
    mp.spawn(_worker_lanes, args=(2, port, w, h, seeds, out), nprocs=2, join=True)
    merged = np.load(out)
    rounds = np.load(out + ".rounds.npy")
    pdist = importlib.import_module("prismarine-core_amd.dist")
    for f, sd in enumerate(seeds):
        full = OraclePipeline(oracle, scenes, scenes.cornell(open_top=True), w, h, 0, h, seed=sd)
        r = pdist.run_rounds(pdist.Comm(1), full, None, None)
        assert r == rounds[f]
        np.testing.assert_allclose(merged[f][:, :3], full.tsum[:, :3], rtol=1e-5, atol=1e-6)
        assert np.array_equal(merged[f][:, 3], full.tsum[:, 3])
    assert not np.array_equal(merged[0], merged[1])


def _park_run(lanes, rounds, force, depth):
    """psm_lanes_run_sharded's stepping rule (lanes.hip) on oracle pipelines: run until the LOCAL count parks."""
    for s, rt in enumerate(lanes):
        while not (rounds[s] >= depth or (rt.raycountCache < 32 and rounds[s] >= force[s])):
            rt.intersection(None, force=True)
            rt.shade(force=True)
            rounds[s] += 1
    return [rt.raycountCache for rt in lanes]


def _worker_parked(rank, world, port, w, h, seeds, rows, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PSM_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as tdist
    pdist = importlib.import_module("prismarine-core_amd.dist")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    from oracle import oracle as O
    comm = pdist.Comm(world, backend="gloo")
    scene = scenes.cornell(open_top=True)
    y0, y1 = rows[rank]
    lanes = [OraclePipeline(O, scenes, scene, w, h, y0, y1, seed=sd) for sd in seeds]
    k, depth = len(seeds), 16
    rounds = [0] * k
    counts = _park_run(lanes, rounds, [0] * k, depth)
    over, exchanges = [False] * k, 0
    while True:
        mine = torch.tensor([rounds, counts], dtype=torch.int32)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        tdist.all_gather(allv, mine)
        exchanges += 1
        verdict = pdist.decide_sharded([a[0].tolist() for a in allv], [a[1].tolist() for a in allv], depth)
        force = []
        for s in range(k):
            over[s] = over[s] or verdict[s][0]
            force.append(rounds[s] if over[s] else verdict[s][1])
        if all(over):
            break
        counts = _park_run(lanes, rounds, force, depth)
    per = max(b - a for a, b in rows)
    merged = []
    for rt in lanes:
        tile = torch.zeros(per * w * 4, dtype=torch.float32)
        tile[: (y1 - y0) * w * 4] = torch.from_numpy(rt.tsum[y0 * w:y1 * w].reshape(-1).copy())
        allt = comm.gather_to_root(tile)
        if comm.rank == 0:
            m = np.zeros((w * h, 4), np.float32)
            for r in range(world):
                a, b = rows[r]
                m[a * w:b * w] = allt[r * per * w * 4: r * per * w * 4 + (b - a) * w * 4].numpy().reshape(-1, 4)
            merged.append(m)
    if comm.rank == 0:
        np.save(out_path, np.stack(merged))
        np.save(out_path + ".meta.npy", np.asarray(rounds + [exchanges]))
    comm.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("rows", [((0, 2), (2, 30)), ((0, 1), (1, 2), (2, 5), (5, 9), (9, 14), (14, 20), (20, 26), (26, 30))],
                         ids=["world2", "world8"])
def test_two_rank_parking_protocol_equals_unsharded(tmp_path, oracle, scenes, rows):
    """The free-running sharded protocol (lanes park on their LOCAL count, dist.decide_sharded applies the stop rule
    to the GLOBAL one): a rank whose tile dies early has to catch up and keep drawing rand() in step. With two ranks, and
    with the eight of the driver's largest run (gloo; strips of one to six rows: several ranks park rounds before the frame ends)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h, seeds = 36, 30, (5, 977)
    world = len(rows)   # (world 2: rank 0 owns two rows: its local count drops below 32 rounds before the frame ends)
    out = str(tmp_path / "parked.npy")
    mp.spawn(_worker_parked, args=(world, port, w, h, seeds, rows, out), nprocs=world, join=True)
    merged = np.load(out)
    meta = np.load(out + ".meta.npy")
    pdist = importlib.import_module("prismarine-core_amd.dist")
    for f, sd in enumerate(seeds):
        full = OraclePipeline(oracle, scenes, scenes.cornell(open_top=True), w, h, 0, h, seed=sd)
        r = pdist.run_rounds(pdist.Comm(1), full, None, None)
        assert r == meta[f]
        np.testing.assert_allclose(merged[f][:, :3], full.tsum[:, :3], rtol=1e-5, atol=1e-6)
        assert np.array_equal(merged[f][:, 3], full.tsum[:, 3])
    assert meta[-1] >= 2  # the early-parked rank needed at least one catch-up exchange


def test_decide_sharded_rules():
    pdist = importlib.import_module("prismarine-core_amd.dist")
    d = pdist.decide_sharded
    assert d([[4], [4]], [[3], [7]], 16) == [(True, 4)]            # same round, 10 rays in all: frame over
    assert d([[4], [4]], [[20], [20]], 16) == [(False, 5)]         # same round, 40 rays in all: one more round
    assert d([[2], [5]], [[0], [9]], 16) == [(False, 5)]           # rank 0 is behind: catch up to round 5
    assert d([[16], [16]], [[500], [900]], 16) == [(True, 16)]     # depth reached
    assert d([[3, 4], [3, 2]], [[1, 0], [2, 31]], 16) == [(True, 3), (False, 4)]


def test_tile_rows_cover_the_frame():
    pdist = importlib.import_module("prismarine-core_amd.dist")
    for world in (1, 2, 3, 4, 8):
        for h in (1, 7, 64, 1080):
            rows = [pdist.tile_rows(r, world, h) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert all(y1 - y0 <= per for y0, y1, per in rows)


@pytest.mark.timeout(300)
def test_two_rank_gloo_frame_equals_unsharded(tmp_path, oracle, scenes):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h = 40, 36
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    merged = np.load(out)
    # unsharded reference: same seed, one rank
    pdist = importlib.import_module("prismarine-core_amd.dist")
    full = OraclePipeline(oracle, scenes, scenes.cornell(open_top=True), w, h, 0, h, seed=5)
    pdist.run_rounds(pdist.Comm(1), full, None, None)
    assert full.tsum[:, :3].max() > 0.1
    # the oracle adds deposits in queue order, which differs between 1 and 2 tiles: compare to 1e-5
    np.testing.assert_allclose(merged[:, :3], full.tsum[:, :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(merged[:, 3], full.tsum[:, 3])


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (no RANK in the environment) must not sit in init_process_group
    waiting for peers nobody started: it spawns the two ranks itself. --dry-run stops each rank after its
    communicator works (gloo, no GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    import re
    # (the ranks share this process's stdout: two lines written at the same moment can arrive as one)
    seen = re.findall(r"bench\.py dry run: rank (\d+) of (\d+) reached its communicator; all-reduce over the group = (\d+)", out.stdout)
    assert sorted(seen) == [("0", "2", "2"), ("1", "2", "2")], out.stdout[-1000:]


def test_bench_under_the_drivers_launcher_command():
    """The driver's own N > 1 command -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W -- with --dry-run appended: every rank takes RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the launcher (no second spawn), reaches its communicator and sees the whole group (gloo, no GPU)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    for attempt in range(3):   # (the port is free when it is picked, not necessarily when the launcher binds it: another try then)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                             env=env, capture_output=True, text=True, timeout=300, cwd=root)
        if out.returncode == 0 or "address already in use" not in (out.stderr + out.stdout).lower():
            break
    assert out.returncode == 0, out.stderr[-2000:]
    import re
    # (the ranks share the launcher's stdout: two lines written at the same moment can arrive as one)
    seen = re.findall(r"bench\.py dry run: rank (\d+) of (\d+) reached its communicator; all-reduce over the group = (\d+)", out.stdout)
    assert sorted(seen) == [("0", "2", "2"), ("1", "2", "2")], out.stdout[-1000:]
    # rank 0's line carries the multi-rank fields of the real line (a scaling line must say how many ranks the
    # communicator saw, what each rank traced and that sharded == unsharded was checked): here from gloo's collectives
    import json
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["value"] is None
    assert line["ranks"]["comm_ranks"] == 2 and line["ranks"]["ranks_heard_from"] == [0, 1] and len(line["ranks"]["rays_traced_per_rank"]) == 2
    assert "rccl_ranks" in line and line["sharded_check"].startswith("not run")


def test_native_decide_equals_python_protocol():
    """psm_dist_decide (the C ABI's host-side restatement of the global `< 32 rays -> stop` rule used by
    psm_dist_render_batch) against dist.decide_sharded (the protocol the gloo tests above exercise) on random
    states: ranks at unequal rounds, everybody level with few / many rays, depth reached."""
    import ctypes as C
    psm = importlib.import_module("prismarine-core_amd")
    pdist = importlib.import_module("prismarine-core_amd.dist")
    L = psm.lib()
    rng = np.random.RandomState(3)
    for _ in range(500):
        world, lanes, depth = int(rng.randint(1, 9)), int(rng.randint(1, 7)), int(rng.randint(1, 6))
        base = rng.randint(0, depth + 1, lanes)
        rounds = [[int(max(0, b - rng.randint(0, 2) * rng.randint(0, 3))) for b in base] for _ in range(world)]
        counts = [[int(rng.choice([0, 3, 10, 31, 32, 500])) for _ in range(lanes)] for _ in range(world)]
        want = pdist.decide_sharded(rounds, counts, depth)
        flat = []
        for r in range(world):
            flat += rounds[r] + counts[r]
        arr = (C.c_int32 * len(flat))(*flat)
        over = (C.c_int32 * lanes)()
        force = (C.c_uint32 * lanes)()
        assert L.psm_dist_decide(C.c_uint32(world), C.c_uint32(lanes), arr, C.c_uint32(depth), over, force) == 0
        assert [(bool(over[s]), int(force[s])) for s in range(lanes)] == [(bool(o), int(f)) for o, f in want]


def test_native_decide_turns_a_poisoned_round_into_every_ranks_error():
    """A rank that failed locally keeps the collective sequence and reports rounds = -1 (psm_dist_render_frames /
    _batch); psm_dist_decide then returns PSM_ERR_PEER -- from the same exchange on every rank, since all ranks see the
    same gathered values -- whatever the other entries say, and leaves its outputs alone."""
    import ctypes as C
    psm = importlib.import_module("prismarine-core_amd")
    L = psm.lib()
    PSM_ERR_PEER = -6
    rng = np.random.RandomState(11)
    for _ in range(200):
        world, lanes, depth = int(rng.randint(1, 9)), int(rng.randint(1, 7)), 16
        flat = [int(v) for v in rng.randint(0, 5, world * 2 * lanes)]
        arr = (C.c_int32 * len(flat))(*flat)
        over = (C.c_int32 * lanes)(*([7] * lanes))
        force = (C.c_uint32 * lanes)(*([9] * lanes))
        assert L.psm_dist_decide(C.c_uint32(world), C.c_uint32(lanes), arr, C.c_uint32(depth), over, force) == 0
        bad_rank, bad_lane = int(rng.randint(0, world)), int(rng.randint(0, lanes))
        flat[(bad_rank * 2 + 0) * lanes + bad_lane] = -1
        arr = (C.c_int32 * len(flat))(*flat)
        over = (C.c_int32 * lanes)(*([7] * lanes))
        force = (C.c_uint32 * lanes)(*([9] * lanes))
        assert L.psm_dist_decide(C.c_uint32(world), C.c_uint32(lanes), arr, C.c_uint32(depth), over, force) == PSM_ERR_PEER
        assert list(over) == [7] * lanes and list(force) == [9] * lanes
        # a negative COUNT is not a poison value (counts are never negative; only the rounds row is looked at)
    assert psm.lib().psm_dist_decide(C.c_uint32(1), C.c_uint32(1), (C.c_int32 * 2)(0, 5), C.c_uint32(4), (C.c_int32 * 1)(), (C.c_uint32 * 1)()) == 0


def test_band_dealing_is_the_same_function_everywhere(oracle):
    """The dealing of the 8-row bands (which rank owns band g) in its three statements: dist.band_pattern (Python host),
    psmo_band_pattern (oracle) and BandMap (csrc/psm_internal.h -- compared on the GPU through the tile cameras,
    test_interleaved_tiles_camera_and_gather). Round-robin for equal weights; every rank's share of a period equals its
    weight; a rank's bands are spread evenly (no two gaps of a rank differ by more than the number of ranks)."""
    pdist = importlib.import_module("prismarine-core_amd.dist")
    rng = np.random.RandomState(5)
    for world in range(1, 9):
        assert pdist.band_pattern(world) == list(range(world)) == oracle.band_pattern(world)
    for _ in range(300):
        world = int(rng.randint(1, 9))
        wts = [int(v) for v in rng.randint(0, 8, world)]
        if sum(wts) == 0 or sum(wts) > 64:
            continue
        pat = pdist.band_pattern(world, wts)
        assert pat == oracle.band_pattern(world, wts)
        assert [pat.count(r) for r in range(world)] == wts
    for world in (2, 4, 8):
        wts = pdist.default_band_weights(world)
        pat = pdist.band_pattern(world, wts)
        assert len(pat) == 23 and wts[0] < wts[1]
        for r in range(world):
            pos = [p for p in range(2 * len(pat)) if (pat * 2)[p] == r]
            gaps = [b - a for a, b in zip(pos, pos[1:])]
            assert max(gaps) - min(gaps) <= world, (world, r, gaps)
        total = 1920 * 1080
        shares = [pdist.owned_texels(r, world, 1920, 1080, wts) / total for r in range(world)]
        assert abs(sum(shares) - 1.0) < 1e-12 and shares[0] < min(shares[1:])
