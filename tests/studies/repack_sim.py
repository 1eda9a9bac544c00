#!/usr/bin/env python3
"""CPU-side study (uses the oracle: lives under tests/): how many wave-steps do different lane-repacking
policies of the traversal kernel need on the real per-ray step counts of a frame?

A wave64 steps as long as its slowest ray. Policies simulated on the oracle's per-ray node-visit counts,
rays in queue order:
  whole        one launch, 64 consecutive rays per wave, run to completion        (rt_traverse<..,false>)
  cap C        phased: stop every wave after C steps, survivors re-packed densely  (round 1's PSM_TRAV_PHASES)
  live<T       adaptive: a wave hands over when fewer than T of its lanes are live (after >= MIN steps)
  ideal        sum(steps) / 64
Reports wave-steps (the VALU cost driver), lane utilisation of the box step, hand-overs per ray and launches.
usage: python tests/studies/repack_sim.py [W H]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

scenes = importlib.import_module("prismarine-core_amd.scenes")


def waves(steps):
    pad = (-len(steps)) % 64
    return np.concatenate([steps, np.zeros(pad, np.int64)]).reshape(-1, 64)


def sim_cap(steps, caps):
    """fixed caps per launch; the last launch runs to completion"""
    total, saves, launches = 0, 0, 0
    cur = steps.copy()
    for c in list(caps) + [None]:
        if len(cur) == 0:
            break
        w = waves(cur)
        launches += 1
        if c is None:
            total += w.max(1).sum()
            break
        total += np.minimum(w.max(1), c).sum()
        cur = cur[cur > c] - c
        saves += len(cur)
    return total, saves, launches


def sim_live(steps, thr, min_steps, final_rays=4096, max_launch=32):
    """a wave hands its live rays over once fewer than `thr` are live (not before min_steps)"""
    total, saves, launches = 0, 0, 0
    cur = steps.copy()
    while len(cur):
        launches += 1
        w = waves(cur)
        if len(cur) <= final_rays or launches >= max_launch:
            total += w.max(1).sum()
            break
        srt = np.sort(w, axis=1)[:, ::-1]          # descending: srt[:, k] = (k+1)-th longest ray of the wave
        # live lanes after s steps = #(steps > s); fewer than thr live <=> s >= srt[:, thr-1]
        stop = np.maximum(srt[:, thr - 1], min_steps)
        stop = np.minimum(stop, srt[:, 0])          # never beyond the wave's own end
        total += stop.sum()
        rem = w - stop[:, None]
        cur = rem[rem > 0]
        saves += len(cur)
    return total, saves, launches


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    sc = scenes.sponza_like() if os.environ.get("SCENE", "sponza") == "sponza" else scenes.stress(n_tris=int(os.environ.get("NTRIS", 2_000_000)))
    ob = O.build_scene(sc["tris"])
    rec = []
    O.render_frames(sc, W, H, frames=1, seed=1000, nthreads=8, built=ob, record=rec, frame_streams=True)
    for r in rec:
        rays = r["rays"]
        v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 8)
        steps = v.astype(np.int64)
        n = len(steps)
        ideal = steps.sum() / 64.0
        print("round %d: %d rays, mean %.1f p50 %d p90 %d p99 %d max %d" % (
            r["round"], n, steps.mean(), np.percentile(steps, 50), np.percentile(steps, 90), np.percentile(steps, 99), steps.max()))
        rows = [("whole",) + sim_cap(steps, [])]
        for caps in ([96], [64, 64], [48, 48, 48], [32, 32, 32, 64], [32] * 8):
            rows.append(("cap %s" % ",".join(map(str, caps)),) + sim_cap(steps, caps))
        for thr in (48, 40, 32, 24, 16, 8):
            for ms in (8, 24):
                rows.append(("live<%d min%d" % (thr, ms),) + sim_live(steps, thr, ms))
        for name, tot, saves, launches in rows:
            print("   %-22s wave-steps %9d  util %5.1f %%  hand-overs/ray %.3f  launches %2d" % (
                name, tot, 100.0 * ideal / max(tot, 1), saves / n, launches))


if __name__ == "__main__":
    main()
