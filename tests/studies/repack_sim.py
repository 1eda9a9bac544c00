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




def sim_refill(steps, thr, resident_waves, drain="inplace"):
    """persistent waves with wave-level refill: a wave whose live lanes drop below `thr` finalises its finished
    rays and pulls fresh rays (queue order) into the free lanes; survivors keep their lanes. When the queue is
    exhausted a wave drains in place ("inplace") or hands its survivors over for packed resume launches
    ("handover": simulated with sim_live on the leftovers). Event-driven over wave-step time."""
    import heapq
    n = len(steps)
    cursor = 0
    total = 0
    leftovers = []
    heap = []  # (time of next event, wave id)
    lanes = {}
    for w in range(resident_waves):
        if cursor >= n:
            break
        take = steps[cursor:cursor + 64].copy()
        cursor += len(take)
        lanes[w] = take
        heapq.heappush(heap, (0, w))
    refills = 0
    while heap:
        t, w = heapq.heappop(heap)
        rem = lanes[w]
        rem = rem[rem > 0]
        if cursor < n:
            free = 64 - len(rem)
            take = steps[cursor:cursor + free]
            cursor += len(take)
            rem = np.concatenate([rem, take])
            refills += 1
        if len(rem) == 0:
            continue
        if cursor >= n:  # exhausted: drain or hand over
            if drain == "inplace" or len(rem) >= thr:
                # run until fewer than thr are live, then decide again (or to the end when draining in place)
                if drain == "inplace":
                    total += rem.max()
                    continue
                srt = np.sort(rem)[::-1]
                stop = srt[thr - 1] if len(srt) >= thr else 0
                total += stop
                left = rem - stop
                leftovers.append(left[left > 0])
                continue
            leftovers.append(rem)
            continue
        srt = np.sort(rem)[::-1]
        stop = srt[thr - 1] if len(srt) >= thr else srt[0]  # steps until fewer than thr lanes are live
        stop = max(int(stop), 1)
        total += stop
        lanes[w] = rem - stop
        heapq.heappush(heap, (t + stop, w))
    extra = (0, 0, 0)
    if leftovers:
        lo = np.concatenate(leftovers)
        if len(lo):
            extra = sim_live(lo, thr, 0, final_rays=1024)
    return total + extra[0], refills, 1 + extra[2], len(np.concatenate(leftovers)) if leftovers else 0


def refill_report():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    sc = scenes.sponza_like()
    ob = O.build_scene(sc["tris"])
    rec = []
    O.render_frames(sc, W, H, frames=1, seed=1000, nthreads=8, built=ob, record=rec, frame_streams=True)
    res_waves = 8192 * (W * H) // (1920 * 1080)
    for r in rec:
        rays = r["rays"]
        v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 8)
        steps = v.astype(np.int64)
        ideal = steps.sum() / 64.0
        print("round %d: %d rays; whole %d wave-steps (util %.1f %%)" % (r["round"], len(steps), sim_cap(steps, [])[0], 100 * ideal / sim_cap(steps, [])[0]))
        for thr in (8, 16, 24, 32, 40, 48):
            for drain in ("inplace", "handover"):
                tot, refills, launches, left = sim_refill(steps, thr, res_waves, drain)
                print("   refill<%d %-8s wave-steps %9d  util %5.1f %%  refills/wave-batch %.2f  launches %d  handed over %d" % (
                    thr, drain, tot, 100.0 * ideal / tot, refills / (len(steps) / 64.0), launches, left))


if __name__ == "__main__":
    if os.environ.get("REFILL"):
        refill_report()
    else:
        main()
