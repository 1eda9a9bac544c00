#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no fixtures, so these freeze the oracle's own output on the
32-triangle Cornell scene: sorted Morton codes, canonical BFS node array, per-pixel hit triangle /
distance at 64x64, and accumulated radiance at a fixed seed.  Re-run only when the canonical
semantics in DESIGN.md change deliberately.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

scenes = importlib.import_module("prismarine-core_amd.scenes")
HERE = os.environ.get("PSM_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))   # (PSM_GOLDEN_OUT: oracle/asan.sh regenerates into a scratch directory)

sc = scenes.cornell()
b = O.build_scene(sc["tris"])
w = h = 64
cam = scenes.camera_matrices(sc["eye"], sc["view"], w, h)
rays, *_ = O.camera(O.make_cfg(w, h), cam[0], cam[1], 4242)
hits, counts, _ = O.traverse(b["nodes"], sc["tris"], b["M"], rays["origin"], rays["direct"])
np.savez_compressed(os.path.join(HERE, "cornell_32.npz"), tris=sc["tris"], M=b["M"], keys=b["keys"], idx=b["idx"],
                    nodes_pdata=b["nodes"]["pdata"], nodes_box=b["nodes"]["box"], hit_counts=counts,
                    hit_tri=hits[:, 0]["tri"], hit_t=hits[:, 0]["t"])

sc = scenes.cornell(open_top=True)
img, stats = O.render_frames(sc, 48, 48, frames=2, seed=31337)
np.savez_compressed(os.path.join(HERE, "cornell_open_radiance.npz"), image=img, rays=np.int64(stats["rays"]))
# round-1 additions: textures + normal maps + every texture part, three lights, the 2x supersampled ray grid, the
# per-frame rand() streams of frames in flight, and the 360-degree camera
sct = scenes.textured(scenes.cornell(open_top=True))
L = O.default_lights(3)
L[1]["lightVector"] = (-0.5, 0.8, 0.6, 30.0); L[1]["lightColor"] = (40.0, 10.0, 5.0, 3.0); L[1]["lightOffset"] = (0.2, 0.0, -0.3, 0.0)
L[2]["lightVector"] = (0.1, -1.0, 0.2, 5.0); L[2]["lightColor"] = (2.0, 8.0, 30.0, 1.5); L[2]["lightAmbient"] = (0.05, 0.02, 0.01, 0.0)
img2, st2 = O.render_frames(sct, 64, 48, frames=3, seed=2718, frame_streams=True, lights=L, display=(32, 24))
img3, st3 = O.render_frames(sc, 48, 24, frames=1, seed=99, enable360=True)
np.savez_compressed(os.path.join(HERE, "cornell_open_round1_features.npz"), textured=img2, textured_rays=np.int64(st2["rays"]),
                    pano=img3, pano_rays=np.int64(st3["rays"]))
print("wrote fixtures; rays", stats["rays"], "hit fraction", float((counts > 0).mean()))
