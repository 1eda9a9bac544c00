"""One rank of a tile-sharded render, started as a child process by tests/test_gpu_dist.py.

Several of these share ONE GPU: each builds its own contexts, hierarchy copies and ray buffers, joins the group through
the host-staged transport of the C ABI (psm_dist_connect_hoststaged: RCCL refuses two ranks on one device) and then
calls exactly what a rank of an 8-GPU run calls -- psm_dist_render_frames or psm_dist_render_batch. Rank 0 writes the
accumulated image; every rank writes what it returned.

  python tests/dist_peer.py <json config>
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_scene(scenes, cfg):
    name = cfg["scene"]
    if name == "cornell_far":   # the open box small in the middle of a wide image: the bands above and below see sky only
        sc = scenes.cornell(open_top=True)
        sc["eye"] = np.asarray((0.0, 0.0, 6.4), np.float32)
        sc["view"] = np.asarray((0.0, 0.0, 0.0), np.float32)
        return sc
    if name == "cornell_open":
        return scenes.cornell(open_top=True)
    if name == "sponza_small":
        return scenes.sponza_like(n_tris=20011)
    if name == "sponza":        # BASELINE C3 / C4's scene: S-sponza-like, 262 267 triangles
        return scenes.sponza_like()
    if name == "stress":        # BASELINE C5's scene: S-stress, 9 999 616 triangles
        return scenes.stress()
    if name.startswith("fuzz:"):   # tests/test_gpu_fuzz.py's scene of that seed, without its textures (the ranks load none)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sc, _, _, _ = importlib.import_module("test_gpu_fuzz").fuzz_scene(int(name[5:]), scenes)
        sc["materials"] = [{k: v for k, v in m.items() if not k.endswith("Part")} for m in sc["materials"]]
        return sc
    raise KeyError(name)


def main():
    cfg = json.loads(sys.argv[1])
    rank, world = cfg["rank"], cfg["world"]
    out = {"rank": rank, "rc": None, "error": None, "rounds": None}
    path = os.path.join(cfg["out"], "rank%d.json" % rank)
    psm = importlib.import_module("prismarine-core_amd")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    pdist = importlib.import_module("prismarine-core_amd.dist")
    scene = make_scene(scenes, cfg)
    w, h, lanes, frames = cfg["w"], cfg["h"], cfg["lanes"], cfg["frames"]
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    b = psm.FrameBatch(lanes, w, h, seed=cfg["seed"])
    fail = cfg.get("fail") if cfg.get("fail_rank") == rank else None
    b.allocate(scene["tris"].shape[0])
    b.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    if fail == "build":         # this rank's LAST lane holds no triangles: its rebuild fails while the peers' succeed
        b.lanes[-1].th.clearTribuffer()
    b.applyMaterials(ms)
    b.each(lambda r: r.setTileInterleaved(rank, world, cfg.get("weights")))
    if fail == "gather":        # this rank's last lane carries a tile that is not the communicator's: its gather is refused
        b.lanes[-1].rays.setTileInterleaved((rank + 1) % world, world, cfg.get("weights"))
    nd = pdist.NativeDist(b.lanes[0].ctx, rank, world)
    if cfg.get("weights"):
        nd.set_band_weights(cfg["weights"])
    try:
        nd.connect_hoststaged(cfg["shm"], pdist.largest_tile_texels(world, w, h, cfg.get("weights")) * 16, cfg.get("timeout_ms", 60000))
        assert nd.transport == "host-staged"
        seeds = b.frame_seeds(frames)
        cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
        if cfg["mode"] == "frames":
            rounds = b.render_frames_sharded(nd, seeds, cam[0], cam[1], depth=cfg.get("depth", 16))
        else:
            rounds = []
            for f0 in range(0, frames, lanes):
                rounds += b.render_batch_sharded(nd, seeds[f0:f0 + lanes], cam[0], cam[1], depth=cfg.get("depth", 16))
        out["rc"], out["rounds"] = 0, [int(v) for v in rounds]
        if rank == 0:
            np.save(os.path.join(cfg["out"], "image.npy"), b.snapHdr())
    except psm.PsmError as e:
        out["rc"], out["error"] = 1, str(e)
    json.dump(out, open(path, "w"))
    nd.close()
    b.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
