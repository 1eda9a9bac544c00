#!/usr/bin/env python3
"""CPU-side study (uses the oracle: lives under tests/): does the ORDER in which the traversal kernel maps queue
slots to lanes matter for wave-steps? The queue order is canonical (parents in order, each parent's diffuse /
reflection / shadow rays next to each other), so a wave of a bounce round mixes ray types whose paths differ in
length. Results are per ray, so the kernel may walk a segment's rays in any order. Simulated on the oracle's
per-ray node-visit counts: lane maps within groups of G consecutive rays (a shading workgroup's output segment):
  canonical      queue order
  by type        the group's rays sorted by ray type (stable)
  by type alt    the same, type order reversed in every other group (waves that straddle groups stay pure)
  type+octant    by type, then by direction octant
  by steps       sorted by the step count itself (unreachable bound for any grouping inside a group)
usage: python tests/studies/order_sim.py [W H]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

scenes = importlib.import_module("prismarine-core_amd.scenes")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from repack_sim import sim_cap, sim_live  # noqa: E402


def reorder(steps, key_fn, G):
    out = []
    for g, s in enumerate(range(0, len(steps), G)):
        idx = np.arange(s, min(s + G, len(steps)))
        k = key_fn(idx, g)
        out.append(idx[np.argsort(k, kind="stable")])
    return steps[np.concatenate(out)] if out else steps


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    sc = scenes.sponza_like() if os.environ.get("SCENE", "sponza") == "sponza" else scenes.stress(n_tris=int(os.environ.get("NTRIS", 2_000_000)))
    ob = O.build_scene(sc["tris"])
    rec = []
    O.render_frames(sc, W, H, frames=1, seed=1000, nthreads=8, built=ob, record=rec, frame_streams=True)
    prev_n = None
    prev_steps_by_texel = None
    for r in rec:
        rays = r["rays"]
        v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 8)
        steps = v.astype(np.int64)
        n = len(steps)
        typ = (rays["bitfield"].astype(np.int64) >> 1) & 3
        d = rays["direct"]
        octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
        # a shading workgroup turns 256 input rays into one segment: its mean size is 256 * n / (rays of the round before)
        G = 256 if prev_n is None else max(64, int(round(256.0 * n / prev_n)))
        prev_n = n
        # the step count of the ray(s) of the round before at the same texel: what a shading workgroup could know
        texel = rays["texel"].astype(np.int64) if "texel" in rays.dtype.names else rays["origin"][:, 3].view(np.int32).astype(np.int64)
        if prev_steps_by_texel is not None:
            parent = prev_steps_by_texel[texel]
            print("   correlation of a ray's steps with its texel's steps in the round before: %.3f" % float(np.corrcoef(parent, steps)[0, 1]))
        else:
            parent = np.zeros(n, dtype=np.int64)
        acc = np.zeros(W * H, dtype=np.float64)
        cnt = np.zeros(W * H, dtype=np.float64)
        np.add.at(acc, texel, steps)
        np.add.at(cnt, texel, 1.0)
        prev_steps_by_texel = acc / np.maximum(cnt, 1.0)
        ideal = steps.sum() / 64.0
        print("round %d: %d rays, segment ~%d rays; types spec/diffuse/shadow = %s; mean steps by type %s" % (
            r["round"], n, G, np.bincount(typ, minlength=3)[:3].tolist(),
            [round(float(steps[typ == t].mean()), 1) if (typ == t).any() else 0 for t in range(3)]))
        orders = [
            ("canonical", steps),
            ("by type", reorder(steps, lambda i, g: typ[i], G)),
            ("by type alt", reorder(steps, lambda i, g: typ[i] if g % 2 == 0 else -typ[i], G)),
            ("type+octant alt", reorder(steps, lambda i, g: (typ[i] * 8 + octant[i]) * (1 if g % 2 == 0 else -1), G)),
            ("by parent steps", reorder(steps, lambda i, g: parent[i] * (1 if g % 2 == 0 else -1), G)),
            ("parent, 4 G", reorder(steps, lambda i, g: parent[i] * (1 if g % 2 == 0 else -1), 4 * G)),
            ("by steps (bound)", reorder(steps, lambda i, g: steps[i] * (1 if g % 2 == 0 else -1), G)),
        ]
        for name, st in orders:
            whole = sim_cap(st, [])[0]
            live = sim_live(st, 12, 8, final_rays=65536 // 4, max_launch=3)[0]
            print("   %-18s whole: wave-steps %9d util %5.1f %%   live<12 x3: wave-steps %9d util %5.1f %%" % (
                name, whole, 100.0 * ideal / whole, live, 100.0 * ideal / live))


if __name__ == "__main__":
    main()
