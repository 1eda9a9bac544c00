#!/usr/bin/env python3
"""A headless run of what the reference's example viewer does with a scene file -- GltfViewer::init then `frames` times
GltfViewer::process (Source/Examples/Viewer.cpp:15-315) -- on the package's mirror of the reference's classes: load the glTF (or OBJ)
scene, rebuild the hierarchy and trace one sample per pixel per frame, write the accumulated HDR image as a PFM (the viewer's own
snapshot is an EXR through FreeImage, Application.hpp:324-343). An example and a smoke test of the file-to-image path, not a viewer.

  python tools/render_gltf.py -m court.gltf -di tests/golden/gltf -s 1.0 -d 16 --size 640x360 --frames 64 -o court.pfm

-m / -s / -di / -d are the reference viewer's own options (Viewer.cpp:22-49). Needs an MI355X: there is no CPU path."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--model", required=True)
    ap.add_argument("-s", "--scale", type=float, default=1.0)
    ap.add_argument("-di", "--dir", default=".")
    ap.add_argument("-d", "--depth", type=int, default=16)
    ap.add_argument("--size", default="640x360")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--eye", default=None, help="x,y,z (default: in front of and above the scene's bounds)")
    ap.add_argument("--view", default=None, help="x,y,z (default: the centre of the scene's bounds)")
    ap.add_argument("--sky", default="0.5,0.7,1.0")
    ap.add_argument("-o", "--out", default="frame.pfm")
    args = ap.parse_args()
    psm = importlib.import_module("prismarine-core_amd")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    gltf = importlib.import_module("prismarine-core_amd.gltf")
    w, h = (int(v) for v in args.size.lower().split("x"))
    path = os.path.join(args.dir, args.model)
    ctx = psm.Context(0)
    intersector = psm.TriangleHierarchy(ctx)
    materials = psm.MaterialSet()
    if path.lower().endswith(".obj"):
        sc = scenes.read_obj(path)
        intersector.allocate(sc["tris"].shape[0])
        intersector.loadTriangles(sc["tris"] * np.float32(args.scale), sc["normals"], sc["mats"], sc.get("texcoords"))
        for m in sc["materials"]:
            materials.addSubmat(m)
        if sc.get("textures"):
            ts = psm.TextureSet()
            for slot in sorted(sc["textures"]):
                ts.loadTexture(sc["textures"][slot])
            materials.setTextureSet(ts)
    else:
        sc = gltf.read_gltf(path, mscale=args.scale)
        intersector.allocate(max(sc["triangle_count"], 1))
        gltf.load_into(sc, intersector, materials)
    n = intersector.triangleCount
    pos = intersector.download(psm.BVH_POSITIONS, np.float32, 9 * n).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    c, ext = 0.5 * (lo + hi), float((hi - lo).max())
    eye = np.asarray([float(v) for v in args.eye.split(",")], np.float32) if args.eye else (c + np.asarray((0.1 * ext, 0.35 * ext, 0.6 * ext))).astype(np.float32)
    view = np.asarray([float(v) for v in args.view.split(",")], np.float32) if args.view else c.astype(np.float32)
    rays = psm.Pipeline(ctx)
    rays.resizeBuffers(w, h)
    rays.resize(w, h)
    rays.setSky([float(v) for v in args.sky.split(",")])
    rays.clearSampler()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        intersector.markDirty()
        psm.render_frame(rays, intersector, materials, eye, view, depth=args.depth)   # GltfViewer::process, Viewer.cpp:296-312
    img = rays.snapHdr()
    ctx.sync()
    dt = time.perf_counter() - t0
    psm.write_pfm(args.out, img[..., :3])
    print("%s: %d triangles, %d materials, %dx%d, %d frames in %.1f ms (%.2f ms per frame), mean radiance %.4f -> %s"
          % (args.model, n, materials.getMaterialCount(), w, h, args.frames, dt * 1e3, dt * 1e3 / max(args.frames, 1), float(img[..., :3].mean()), args.out))


if __name__ == "__main__":
    main()
