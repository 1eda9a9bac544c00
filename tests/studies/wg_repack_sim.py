"""CPU-side study (uses the oracle: lives under tests/), round 5: would repacking rays INSIDE a workgroup -- no extra launch, no global
hand-over -- buy lane utilisation? Simulation: park-and-last-wave-gathers. WG of NW waves; a wave that drops below T live rays (after MIN steps) parks them in the WG's pool
and exits; the LAST wave of the WG to get there keeps its rays, takes rays from the pool up to 64 and goes on; whenever it drops below T again and the
pool is not empty it refills from the pool; with the pool empty it runs to completion. Wave-steps, moves per ray, and the last wave's serial steps."""
import importlib, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
scenes = importlib.import_module("prismarine-core_amd.scenes")
W, H = 640, 360
sc = scenes.sponza_like()
ob = O.build_scene(sc["tris"])
rec = []
O.render_frames(sc, W, H, frames=1, seed=1000, nthreads=8, built=ob, record=rec, frame_streams=True)

def sim(steps, NW, T, MIN=8):
    total = 0; moves = 0; serial = []
    per = NW * 64
    for base in range(0, len(steps), per):
        chunk = steps[base:base + per]
        waves = [np.array([s for s in chunk[k * 64:(k + 1) * 64] if s > 0]) for k in range(NW)]
        # phase 1: each wave alone until fewer than T live (after MIN steps) -- or done
        pool = []; ends = []
        for w in waves:
            if len(w) == 0:
                ends.append(0); continue
            srt = np.sort(w)[::-1]
            stop = max(int(srt[T - 1]) if len(srt) >= T else 0, min(MIN, int(srt[0])))   # live < T  <=>  s >= srt[T-1]
            stop = min(stop, int(srt[0]))
            total += stop
            rest = w[w > stop] - stop
            ends.append(stop)
            pool.append(rest)
        # the wave that parks last keeps going: take the wave with the latest stop as the gatherer
        order = np.argsort(ends)
        rays = np.concatenate([p for p in pool]) if pool else np.array([], np.int64)
        moves += len(rays)
        ser = 0
        cur = rays[:64]; rest = rays[64:]
        while len(cur):
            srt = np.sort(cur)[::-1]
            if len(rest) == 0:
                total += int(srt[0]); ser += int(srt[0]); break
            stop = max(int(srt[T - 1]) if len(srt) >= T else 0, 1)
            stop = min(stop, int(srt[0]))
            total += stop; ser += stop
            cur = cur[cur > stop] - stop
            take = 64 - len(cur)
            cur = np.concatenate([cur, rest[:take]]); rest = rest[take:]
        serial.append(ser)
    return total, moves, np.mean(serial), np.max(serial)

for rnd, r_ in enumerate(rec[:4]):
    rays = r_["rays"]
    v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 8)
    steps = v.astype(np.int64)
    ideal = steps.sum() / 64.0
    w = np.concatenate([steps, np.zeros((-len(steps)) % 64, np.int64)]).reshape(-1, 64)
    print("round %d: %d rays, mean %.1f max %d; whole util %.1f %%" % (rnd, len(steps), steps.mean(), steps.max(), 100 * ideal / w.max(1).sum()))
    for NW in (2, 4, 8):
        for T in (16, 24, 32, 40):
            tot, mv, sm, sx = sim(steps, NW, T)
            print("   WG %d waves, park below %2d: util %5.1f %%  moves/ray %.2f  gatherer's serial steps mean %.0f max %d" % (NW, T, 100 * ideal / tot, mv / len(steps), sm, sx))
