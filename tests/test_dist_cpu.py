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
