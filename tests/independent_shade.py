"""An independent second opinion on one shading round (test infrastructure).

Written from the reference's GLSL alone -- raytracing/rayshading.comp:48-278, raytracing/surface.comp:165-195 (the
texture-less path), raytracing/directTraverse.comp:116-217 (interpolateMeshData), include/shadinglib.glsl,
include/rayslib.glsl:59-203, include/random.glsl -- in plain Python with numpy float32 scalars and numpy's own
sin / cos / pow / sqrt, NOT from oracle/psm_oracle_shade.c and not through the pinned polynomials of psm_math.h. It
covers what a frame of the flat-material scenes needs: rays with at most one hit in their chain, no textures, any
number of lights, a constant sky colour. Canonical rules it shares with the oracle by construction:
the RNG stream id is the ray's path key, the output order per input ray is [current, diffuse, reflection, shadow],
radiance is a per-texel sum.

shade_round(...) -> (list of output ray tuples, {texel: [r, g, b, deposits]})
"""
import numpy as np

F = np.float32
PZERO = F(0.0005)
INF = F(10000.0)
GAP = F(PZERO * F(2.0))
TWO_PI = F(6.2831853071795864769252867665590057683943)
SQRT13 = F(0.5773502691896257645091487805019574556476)
U32 = 0xFFFFFFFF


def hash32(x):
    x = (x + (x << 10)) & U32
    x ^= x >> 6
    x = (x + (x << 3)) & U32
    x ^= x >> 11
    x = (x + (x << 15)) & U32
    return x


class Rng:  # random(), include/random.glsl:37-46, with globalInvocationSMP = the path key
    def __init__(self, smp, time):
        self.smp, self.clocks, self.t5 = smp & U32, 0, (time << 5) & U32

    def next(self):
        hs = self.clocks
        self.clocks = hash32((self.clocks + 1) & U32)
        h = hash32(self.smp ^ hash32(hs) ^ hash32(self.t5))
        f = np.array([(h & 0x007FFFFF) | 0x3F800000], np.uint32).view(np.float32)[0]
        return F(f - np.floor(f))  # fract


def v3(x, y, z):
    return np.array([x, y, z], np.float32)


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def normalize(a):
    return (a * F(F(1.0) / F(np.sqrt(dot(a, a))))).astype(np.float32)


def cross(a, b):
    return v3(F(a[1] * b[2]) - F(b[1] * a[2]), F(a[2] * b[0]) - F(b[2] * a[0]), F(a[0] * b[1]) - F(b[0] * a[1]))


def mix(x, y, a):
    return (x * (F(1.0) - a) + y * a).astype(np.float32) if isinstance(x, np.ndarray) else F(F(x * F(F(1.0) - a)) + F(y * a))


def clamp(x, lo, hi):
    return np.minimum(np.maximum(x, F(lo)), F(hi)).astype(np.float32) if isinstance(x, np.ndarray) else min(max(x, F(lo)), F(hi))


def mlength(c):
    return max(c[0], max(c[1], c[2]))


def half(x):  # packHalf + unpackHalf round trip
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def bf(b, off, bits):
    return (b >> off) & ((1 << bits) - 1)


def bfs(b, v, off, bits):
    m = ((1 << bits) - 1) << off
    return (b & ~m) | ((v << off) & m)


ACT, TYPE, DL, TARGET, BOUNCE, BASIS = (0, 1), (1, 2), (3, 1), (4, 4), (8, 4), (12, 1)


class Ray:
    def __init__(self, origin, direct, color, final, b, texel):
        self.origin, self.direct, self.color, self.final, self.b, self.texel = origin, direct, color, final, int(b), int(texel)

    def copy(self):
        return Ray(self.origin.copy(), self.direct.copy(), self.color.copy(), self.final.copy(), self.b, self.texel)

    def get(self, f):
        return bf(self.b, *f)

    def set(self, f, v):
        self.b = bfs(self.b, int(v), *f)


def random_cosine(g, normal):  # random.glsl:48-69
    up = F(np.sqrt(g.next()))
    over = F(np.sqrt(F(F(1.0) - F(up * up))))
    around = F(g.next() * TWO_PI)
    p0 = v3(0, 0, 1)
    if abs(normal[0]) < SQRT13:
        p0 = v3(1, 0, 0)
    elif abs(normal[1]) < SQRT13:
        p0 = v3(0, 1, 0)
    p1 = normalize(cross(normal, p0))
    p2 = cross(normal, p1)
    ca, sa = F(F(np.cos(around)) * over), F(F(np.sin(around)) * over)
    return normalize((normal * up + (p1 * ca + p2 * sa)).astype(np.float32))


def random_direction_in_sphere(g):  # random.glsl:71-76
    up = F(F(g.next() * F(2.0)) - F(1.0))
    over = F(np.sqrt(F(F(1.0) - F(up * up))))
    around = F(g.next() * TWO_PI)
    return normalize(v3(up, F(F(np.cos(around)) * over), F(F(np.sin(around)) * over)))


def light_center(L):  # shadinglib.glsl:22-26
    lv = np.asarray(L["lightVector"], np.float32)
    lvec = normalize(lv[:3]) * (F(-1.0) if lv[1] < 0 else F(1.0))
    return (lvec * lv[3] + np.asarray(L["lightOffset"], np.float32)[:3]).astype(np.float32)


def intersect_sphere(origin, ray, centre, radius):  # shadinglib.glsl:32-48
    ts = (origin - centre).astype(np.float32)
    a = dot(ray, ray)
    b = F(F(2.0) * dot(ts, ray))
    c = F(dot(ts, ts) - F(radius * radius))
    disc = F(F(b * b) - F(F(F(4.0) * a) * c))
    t = INF
    if disc > 0:
        da = F(F(0.5) / a)
        sq = F(np.sqrt(disc))
        t1, t2 = F(F(-b - sq) * da), F(F(-b + sq) * da)
        mn, mx = min(t1, t2), max(t1, t2)
        if mx >= 0:
            t = mn if mn >= 0 else mx
    return t


class Sink:
    """queue + per-texel sums: createRay / storeRay / _collect (rayslib.glsl:59-203)"""

    def __init__(self):
        self.out, self.tex = [], {}

    def collect(self, ray):
        c = np.maximum(ray.final, F(0.0))
        if mlength(c) < F(10000.0) and not np.isnan(c).any() and not np.isinf(c).any():
            s = self.tex.setdefault(ray.texel, [0.0, 0.0, 0.0, 0])
            s[0] += float(c[0]); s[1] += float(c[1]); s[2] += float(c[2]); s[3] += 1
        ray.final = v3(0, 0, 0)

    def create_ray(self, ray, pkey):  # createRay -> createRayStrict (`in` parameter: works on a copy)
        ray = ray.copy()
        invalid = ray.get(ACT) == 0 or ray.get(BOUNCE) <= 0 or mlength(ray.color) < F(0.0001)
        if mlength(ray.final) >= F(0.0001) and ray.get(ACT) == 0:
            self.collect(ray)
        ray.set(BASIS, 0)
        if invalid:
            return
        ray.set(BOUNCE, ray.get(BOUNCE) - 1)
        self.out.append((ray.origin.copy(), ray.direct.copy(), ray.color.copy(), ray.b, ray.texel, pkey))


def child_key(pkey, site):
    return hash32(pkey ^ hash32(site))


def shade_round(tris, normals, tri_mats, materials, mat_offset, lights, sky, time, rays, hits, counts):
    sink = Sink()
    nmat = len(materials)
    for it in range(rays.shape[0]):
        r = rays[it]
        ray = Ray(np.array(r["origin"], np.float32), np.array(r["direct"], np.float32), np.array(r["color"], np.float32),
                  v3(0, 0, 0), r["bitfield"], r["texel"])
        pkey = int(r["pkey"])
        g = Rng(pkey, time)
        n = int(counts[it])
        assert n <= 1, "this restatement covers chains of at most one hit"
        skipping = False
        # ---- hit: interpolateMeshData (directTraverse.comp:116-217) + surface.comp:165-195, no textures
        albedo = emission = mr = np.zeros(4, np.float32)
        normal_h = v3(0, 0, 0)
        uvt_t = INF
        if n == 1:
            h = hits[it, 0]
            u, v, uvt_t, tri = F(h["u"]), F(h["v"]), F(h["t"]), int(h["tri"])
            p = tris[tri]
            vs = v3(F(F(F(1.0) - u) - v), u, v)
            d1, d2 = (p[1] - p[0]).astype(np.float32), (p[2] - p[0]).astype(np.float32)
            nor = normalize(cross(d1, d2))
            tn = normals[tri]
            nrm = v3(*[F(F(F(vs[0] * tn[0][k]) + F(vs[1] * tn[1][k])) + F(vs[2] * tn[2][k])) for k in range(3)])
            nrm = normalize(nrm)  # lessF(length, 0) never holds
            nrm = (nrm * np.sign(dot(nrm, nor))).astype(np.float32)
            # tangent from all-zero texcoords: deltas fall back to (1,0),(1,0); f = 1
            tang = (d1 * F(1.0) + d2 * F(0.0)).astype(np.float32)
            tangent = normalize((tang - nrm * np.sign(dot(tang, nor))).astype(np.float32))
            mat_id = int(tri_mats[tri]) - mat_offset
            active = 0 <= mat_id < nmat
            nh = normalize(nrm)
            if active:
                m = materials[mat_id]
                tg = normalize(tangent)
                bt = normalize(cross(nh, tg))
                nm = normalize(normalize(v3(0, 0, 1)))  # getNormalMapping of the default (0.5, 0.5, 1) texel
                shn = normalize((tg * nm[0] + bt * nm[1] + nh * nm[2]).astype(np.float32))
                diffuse = np.maximum(np.array([m["diffuse"][0], m["diffuse"][1], m["diffuse"][2], 1.0], np.float32), F(0.0))
                albedo = half(diffuse)
                emission = half(np.array([0.0, 0.0, 0.0, 1.0], np.float32))  # fetchEmissive = 0 without a texture, x2, w = 1
                spc = np.asarray(m["specular"], np.float32)
                mr = half(np.array([spc[1], spc[2], 0.0, 0.0], np.float32))
                normal_h = shn
            else:
                normal_h = nrm  # not actived: transparent, keeps the traversal normal (rayshading.comp:73-79)
        # ---- physical lights (:119-138)
        lc = -1
        typ = ray.get(TYPE)
        if ray.get(DL) > 0 and typ in (1, 2) and not skipping:
            for i in range(min(len(lights), 16)):
                dt = intersect_sphere(ray.origin, ray.direct, light_center(lights[i]), F(F(lights[i]["lightColor"][3]) + GAP))
                t = F(F(1.0) * dt)
                if F(INF - dt) >= PZERO and F(uvt_t - t) > -PZERO:
                    lc = i
        if lc >= 0 and (ray.get(TARGET) == lc or typ != 2):
            ray.final = (ray.color * np.maximum(np.asarray(lights[lc]["lightColor"], np.float32)[:3], F(0.0))).astype(np.float32)
            ray.color = (ray.color * F(0.0)).astype(np.float32)
            ray.set(ACT, 0)
            skipping = True
        # ---- background (:141-152)
        if F(uvt_t - INF) > -PZERO and typ != 2 and not skipping:
            ray.final = (ray.color * np.asarray(sky, np.float32)[:3]).astype(np.float32)
            ray.color = (ray.color * F(0.0)).astype(np.float32)
            ray.set(ACT, 0)
            skipping = True
        ray.direct = normalize(ray.direct)
        ray.origin = (ray.origin + ray.direct * uvt_t).astype(np.float32)
        if ray.get(ACT) < 1 or n == 0:
            skipping = True
        surfacenormal = normal_h
        normal = surfacenormal if dot(surfacenormal, ray.direct) < 0 else (-surfacenormal).astype(np.float32)
        refly = mr[0]
        pw = clamp(F(np.power(abs(dot(ray.direct, normal)), F(F(1.4) - F(1.0)), dtype=np.float32)), 0.0, 1.0)
        dielectric = mix(F(1.0), F(0.05), pw)
        sm = F(np.sqrt(mr[1]))
        sc = v3(*[mix(dielectric, albedo[k], sm) for k in range(3)])
        emis = mlength(emission[:3])
        spca = clamp(mlength(sc), 0.0, 1.0)
        prom = F(F(1.0) - albedo[3])
        aprom = prom if typ == 2 else (F(1.0) if g.next() < prom else F(0.0))
        diffuse_ray, reflection_ray, emissive_ray = ray.copy(), ray.copy(), ray.copy()
        for q in (diffuse_ray, reflection_ray, emissive_ray):
            q.final = (q.final * F(0.0)).astype(np.float32)
        if not skipping:
            ray.final = (ray.final * F(0.0)).astype(np.float32)
        if ray.get(ACT) > 0 and not skipping:
            # diffuse(), shadinglib.glsl:106-119 (DIRECT_LIGHT_ENABLED)
            q = diffuse_ray
            q.color = (q.color * albedo[:3]).astype(np.float32)
            q.direct = normalize(random_cosine(g, normal))
            q.origin = (q.direct * GAP + q.origin).astype(np.float32)
            q.set(ACT, 0 if q.get(TYPE) == 2 else q.get(ACT))
            q.set(BOUNCE, min(2, q.get(BOUNCE)))
            q.set(TYPE, 1)
            q.set(DL, 0)
            # reflection(), :139-148 (SUNLIGHT_CAUSTICS false)
            q = reflection_ray
            col = clamp((sc / spca).astype(np.float32), 0.0, 1.0)
            dn = dot(normal, q.direct)
            refl = (q.direct - normal * F(F(2.0) * dn)).astype(np.float32)
            rc = random_cosine(g, normal)
            al = clamp(F(refly * g.next()), 0.0, 1.0)
            q.direct = normalize(mix(refl, rc, al))
            q.color = (q.color * col).astype(np.float32)
            q.origin = (q.direct * GAP + q.origin).astype(np.float32)
            q.set(DL, 0 if q.get(TYPE) == 1 else 1)
            q.set(TYPE, 0)
            q.set(BOUNCE, min(3, q.get(BOUNCE)))
            q.set(ACT, 0 if q.get(TYPE) == 2 else q.get(ACT))
            # emissive(), :127-137
            q = emissive_ray
            q.final = np.maximum((q.color * emission[:3]).astype(np.float32), F(0.0))
            if q.get(TYPE) == 1:
                q.final = v3(0, 0, 0)
            q.color = (q.color * F(0.0)).astype(np.float32)
            q.direct = normalize(random_cosine(g, normal))
            q.origin = (q.direct * GAP + q.origin).astype(np.float32)
            q.set(BOUNCE, 0); q.set(ACT, 0); q.set(DL, 0)
            # promised(), :121-125
            ray.set(BOUNCE, ray.get(BOUNCE) + 1)
            ray.origin = (ray.direct * GAP + ray.origin).astype(np.float32)
            ray.color = (ray.color * aprom).astype(np.float32)
            ray.final = (ray.final * aprom).astype(np.float32)
        else:
            for q in (reflection_ray, emissive_ray, diffuse_ray):
                q.color = (q.color * F(0.0)).astype(np.float32)
            diffuse_ray.final = (diffuse_ray.final * F(0.0)).astype(np.float32)
        if not skipping:
            om = F(F(1.0) - aprom)
            diffuse_ray.color = (diffuse_ray.color * om).astype(np.float32)
            diffuse_ray.final = (diffuse_ray.final * om).astype(np.float32)
            reflection_ray.color = (reflection_ray.color * om).astype(np.float32)
            emissive_ray.final = (emissive_ray.final * F(om * F(F(1.0) - clamp(spca, 0.0, 1.0)))).astype(np.float32)
        if ray.get(BASIS) == 1 and aprom < F(0.1):
            ray.set(BASIS, 0)
        # reclaim the current ray (:235-251): storeRay deposits, addRayToList queues
        bounce = ray.get(BOUNCE) - 1
        # max(vec3(0.0f), v): GLSL's max(x, y) is `x < y ? y : x`, so a NaN component of v gives x = 0 -- numpy's maximum would hand the NaN
        # on and the ray would stay active with a NaN colour (round 5: the fuzzer's black full-metal surfaces,
        # tests/test_oracle_cpu.py::test_fuzzed_scenes_shade_against_the_independent_reading)
        ray.final = np.where(F(0.0) < ray.final, ray.final, F(0.0)).astype(np.float32)
        ray.color = np.where(F(0.0) < ray.color, ray.color, F(0.0)).astype(np.float32)
        if bounce < 0 or mlength(ray.color) < F(0.0001) or n == 0:
            ray.set(ACT, 0)
        ray.set(BOUNCE, bounce if bounce >= 0 else 0)
        if mlength(ray.final) >= F(0.0001) and ray.get(ACT) == 0:
            sink.collect(ray)
        if ray.get(ACT) == 1:
            sink.out.append((ray.origin.copy(), ray.direct.copy(), ray.color.copy(), ray.b, ray.texel, pkey))
        # emit new rays (:263-275)
        if not skipping:
            coef = clamp(F(1.0) if g.next() < spca else F(0.0), 0.0, 1.0)
            reflection_ray.color = (reflection_ray.color * coef).astype(np.float32)
            diffuse_ray.color = (diffuse_ray.color * F(F(1.0) - coef)).astype(np.float32)
            # directLight(0, diffuseRay, 1, normal), shadinglib.glsl:75-93
            sh = diffuse_ray.copy()
            sh.set(ACT, 0 if sh.get(TYPE) == 2 else sh.get(ACT))
            sh.set(DL, 1); sh.set(TYPE, 2); sh.set(TARGET, 0)
            sh.set(BOUNCE, min(1, sh.get(BOUNCE)))
            L0 = lights[0]
            ctr = light_center(L0)
            radius = F(L0["lightColor"][3])
            sl = (random_direction_in_sphere(g) * F(radius - F(0.0001)) + ctr).astype(np.float32)
            ldirect = normalize((sl - sh.origin).astype(np.float32))
            d = (ctr - sh.origin).astype(np.float32)
            dist = F(np.sqrt(dot(d, d)))
            q2 = F(np.power(F(radius / dist), F(2.0), dtype=np.float32))
            weight = F(F(1.0) - F(np.sqrt(F(F(1.0) - clamp(F(F(dot(ldirect, normal) * F(2.0)) * q2), 0.0, 1.0)))))
            sh.origin = (sh.direct * (-GAP) + sh.origin).astype(np.float32)
            sh.direct = ldirect
            sh.color = (sh.color * F(F(1.0) * weight)).astype(np.float32)
            sh.final = (sh.final * F(0.0)).astype(np.float32)
            sh.origin = (sh.direct * GAP + sh.origin).astype(np.float32)
            sink.create_ray(diffuse_ray, child_key(pkey, 1))
            sink.create_ray(reflection_ray, child_key(pkey, 2))
            ce = clamp(emis, 0.0, 1.0)
            emissive_ray.color = (emissive_ray.color * ce).astype(np.float32)
            emissive_ray.final = (emissive_ray.final * ce).astype(np.float32)
            sink.create_ray(emissive_ray, child_key(pkey, 4))
            # applyLight, shadinglib.glsl:181-189
            if diffuse_ray.get(TYPE) == 2 or dot(surfacenormal, sh.direct) < 0:
                sh.set(ACT, 0)
            sink.create_ray(sh, child_key(pkey, 3))
    return sink.out, sink.tex
