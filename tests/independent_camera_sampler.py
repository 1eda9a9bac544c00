"""An independent second opinion on the camera pass and the sampler (test infrastructure).

Written from the reference's GLSL alone -- raytracing/camera.comp:22-101 (the perspective branch), include/rayslib.glsl
:129-156,205-222 (how the ray is stored), include/random.glsl:11-46, include/mathlib.glsl:27,73-81 (divW, mult4) and raytracing/sampler.comp:37-97 -- in plain numpy float32, NOT
from oracle/psm_oracle*.c. Canonical rules it shares with the oracle by construction: `time` is an
explicit seed, the two jitter draws are taken x first, a frame's texel value is the sum of its deposits (what
collectSamples adds up over the chain), a texel "has samples" when the camera pass has visited it.

camera_rays(cam_inv, proj_inv, w, h, time) -> origin (n,3), direct (n,3), coord (n,2), bitfield (n,)
sample(coord, tsum, flag, presampled, w, h, dw, dh, samples_lock) -> presampled'
"""
import numpy as np

F = np.float32
U32 = 0xFFFFFFFF


def hash32(x):  # random.glsl:11-22
    x = (x + (x << 10)) & U32
    x ^= x >> 6
    x = (x + (x << 3)) & U32
    x ^= x >> 11
    x = (x + (x << 15)) & U32
    return x


class Rng:  # random(), random.glsl:37-46: globalInvocationSMP = the texel index (camera.comp:27)
    def __init__(self, smp, time):
        self.smp, self.clocks, self.t5 = smp & U32, 0, (time << 5) & U32

    def next(self):
        hs = self.clocks
        self.clocks = hash32((self.clocks + 1) & U32)
        h = hash32(self.smp ^ hash32(hs) ^ hash32(self.t5))
        f = np.array([(h & 0x007FFFFF) | 0x3F800000], np.uint32).view(np.float32)[0]
        return F(f - np.floor(f))


def mult4(m, v):  # mult4(mat, vec) with the host's transposed upload: M v
    m = np.asarray(m, np.float32).reshape(4, 4)
    return (m @ np.asarray(v, np.float32)).astype(np.float32)


def div_w(v):
    return (v / v[3]).astype(np.float32)


def clamp(x, lo, hi):
    return F(min(max(F(x), F(lo)), F(hi)))


def camera_rays(cam_inv, proj_inv, w, h, time):
    n = w * h
    origin = np.zeros((n, 3), np.float32)
    direct = np.zeros((n, 3), np.float32)
    coord = np.zeros((n, 2), np.float32)
    bitfield = np.zeros(n, np.int32)
    res_inv = (F(1.0) / F(w), F(1.0) / F(h))          # sceneResInv, :30
    for idx in range(n):
        g = Rng(idx, time)
        x, y = idx % w, idx // w                        # :29
        rx = clamp(g.next(), 0.00001, 0.99999)          # :35, x then y
        ry = clamp(g.next(), 0.00001, 0.99999)
        cx = F(F(F(x) + rx) * res_inv[0])
        cy = F(F(F(y) + ry) * res_inv[1])
        coord[idx] = (cx, cy)
        ndc = (F(F(cx * F(2.0)) - F(1.0)), F(F(cy * F(2.0)) - F(1.0)))
        co = div_w(mult4(cam_inv, mult4(proj_inv, (ndc[0], ndc[1], F(0.999), F(1.0)))))   # :61
        og = div_w(mult4(cam_inv, mult4(proj_inv, (ndc[0], ndc[1], F(0.0), F(1.0)))))     # :62
        d = (co[:3] - og[:3]).astype(np.float32)
        d = (d / F(np.sqrt(F(np.dot(d, d))))).astype(np.float32)                          # :63
        origin[idx] = og[:3]
        direct[idx] = d
        # :84-92: active, type 0 (specular), DL 0, bounce 4, basis 1 (structs.glsl:73-78 bit layout); the ray is stored by
        # createRayIdx -> createRayStrict (rayslib.glsl:129-156,205-222), which takes one bounce off on the way in
        bitfield[idx] = 1 | (0 << 1) | (0 << 3) | ((4 - 1) << 8) | (1 << 12)
    return origin, direct, coord, bitfield


def sample(coord, tsum, flag, presampled, w, h, dw, dh, samples_lock):
    out = np.array(presampled, np.float32).reshape(dw * dh, 4).copy()
    ax, ay = F(w) / F(dw), F(h) / F(dh)                 # :41
    sclx, scly = int(np.ceil(ax)), int(np.ceil(ay))
    for it in range(dw * dh):
        px, py = it % dw, it // dw
        bx, by = int(F(px) * ax), int(F(py) * ay)       # :47
        cnt = 0
        newc = np.zeros(3, np.float32)
        for x in range(-1, sclx + 1):
            for y in range(-1, scly + 1):
                cx, cy = bx + x, by + y
                if 0 <= cx < w and 0 <= cy < h:
                    ts = cy * w + cx
                    if not flag[ts]:                    # :57 (a texel the camera pass has not visited)
                        continue
                    sx, sy = F(coord[ts][0] * F(dw)), F(coord[ts][1] * F(dh))
                    dx, dy = F(F(sx - F(px)) + F(0.00001)), F(F(sy - F(py)) + F(0.00001))
                    if F(0.0) <= dx < F(1.0) and F(0.0) <= dy < F(1.0):
                        cnt += 1
                        newc = (newc + np.asarray(tsum[ts][:3], np.float32)).astype(np.float32)
        if cnt > 0:
            newc = (newc / F(cnt)).astype(np.float32)   # :71
            xs = out[it]
            nxt = F(xs[3] + F(cnt))
            divisor = F(xs[3] / nxt)
            for k in range(3):                          # :86 fma(xsample, divisor, newc * (1 - divisor))
                xs[k] = F(np.float64(xs[k]) * np.float64(divisor) + np.float64(F(newc[k] * F(F(1.0) - divisor))))
            xs[3] = min(nxt, F(samples_lock - 1)) if samples_lock > 0 else nxt   # :88 (MOTION_BLUR as built: SAMPLES_LOCK-1)
    return out.reshape(np.asarray(presampled).shape)
