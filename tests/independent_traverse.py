"""An independent second opinion on the BVH traversal (test infrastructure).

Written from the reference's GLSL alone -- raytracing/directTraverse.comp:49-112 (stack, reorderTriangles), :261-309
(testIntersectionPacked), :333-484 (traverse), include/mathlib.glsl:10-14,107-126,129-193 (tolerant compares,
intersectCubeSingle, intersectCubeDual: the fp32 branch, no AMD_F16_BVH, no ENABLE_AMD_INSTRUCTION_SET),
include/vertex.glsl:140-189 (intersectTriangle) -- as a plain Python state machine on numpy float32 scalars, NOT from
oracle/psm_oracle.c. It walks the oracle's canonical node array (whose construction has its own second opinion,
independent_build.py). Canonical rules it shares with the oracle by construction: min / max
ignore a NaN operand, fma is fused, M v is summed as ((m0 x + m1 y) + m2 z) + m3 w, a 16-entry stack whose overflowing
pushes are dropped, an equal-distance list of 8 entries whose overflowing writes are dropped.

traverse(nodes, tris, M, origin, direct) -> (chain [(tri, t, u, v), ...] in list order, node visits, triangle tests)
"""
import numpy as np

F = np.float32
PZERO = F(0.0005)
INF = F(10000.0)
LONGEST = -1
STACK = 16          # STACK_SIZE + GLOBAL_STACK_SIZE, :40-41
BAKED = 8           # BAKED_STACK_SIZE, :42


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def fmin(a, b):
    return F(np.fmin(a, b))


def fmax(a, b):
    return F(np.fmax(a, b))


def less_equal_f(a, b):
    return F(b - a) > -PZERO


def less_f(a, b):
    return F(b - a) >= PZERO


def greater_equal_f(a, b):
    return F(a - b) > -PZERO


def equal_f(a, b):
    return abs(F(a - b)) < PZERO


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def cross(a, b):
    return (F(F(a[1] * b[2]) - F(b[1] * a[2])), F(F(a[2] * b[0]) - F(b[2] * a[0])), F(F(a[0] * b[1]) - F(b[0] * a[1])))


def normalize(a):
    inv = F(F(1.0) / F(np.sqrt(dot(a, a))))
    return (F(a[0] * inv), F(a[1] * inv), F(a[2] * inv))


def length(a):
    return F(np.sqrt(dot(a, a)))


def unpack_box(words):
    """bboxf16 {uvec2 mn, mx} -> two triples of float32 (unpackHalf, mathlib.glsl:323-338)"""
    w = np.asarray(words, np.uint32)
    h = np.array([w[0] & 0xFFFF, w[0] >> 16, w[1] & 0xFFFF, w[2] & 0xFFFF, w[2] >> 16, w[3] & 0xFFFF], np.uint16).view(np.float16)
    f = h.astype(np.float32)
    return (f[0], f[1], f[2]), (f[3], f[4], f[5])


def cube_single(o, ray, cmin, cmax):  # mathlib.glsl:107-126
    t1, t2 = [], []
    for k in range(3):
        dr = F(F(1.0) / ray[k])
        norig = F(-o[k] * dr)
        a, b = fma(cmin, dr, norig), fma(cmax, dr, norig)
        t1.append(fmin(a, b))
        t2.append(fmax(a, b))
    t_near = fmax(fmax(t1[0], t1[1]), t1[2])
    t_far = fmin(fmin(t2[0], t2[1]), t2[2])
    is_cube = greater_equal_f(t_far, t_near) and greater_equal_f(t_far, F(0.0))
    near = fmin(t_near, t_far) if is_cube else INF
    far = fmax(t_near, t_far) if is_cube else INF
    d = (far if less_f(near, F(0.0)) else near) if is_cube else INF
    return d, near, far


def cube_child(o, dr, bmin, bmax):  # one column of intersectCubeDual, mathlib.glsl:166-192
    t1, t2 = [], []
    for k in range(3):
        norig = F(-o[k] * dr[k])
        a, b = fma(bmin[k], dr[k], norig), fma(bmax[k], dr[k], norig)
        t1.append(fmin(a, b))
        t2.append(fmax(a, b))
    t_near = fmax(fmax(t1[0], t1[1]), t1[2])
    t_far = fmin(fmin(t2[0], t2[1]), t2[2])
    is_cube = (F(t_far + PZERO) >= t_near) and (F(t_far + PZERO) >= F(0.0))
    near = fmin(t_near, t_far) if is_cube else INF
    far = fmax(t_near, t_far) if is_cube else INF
    hit = far if F(near + PZERO) <= F(0.0) else near
    return hit, near


def intersect_triangle(tris, orig, dr, tri):  # vertex.glsl:140-189; returns (T, u, v)
    v0, v1, v2 = tris[tri][0], tris[tri][1], tris[tri][2]
    e1 = (F(v1[0] - v0[0]), F(v1[1] - v0[1]), F(v1[2] - v0[2]))
    e2 = (F(v2[0] - v0[0]), F(v2[1] - v0[1]), F(v2[2] - v0[2]))
    pvec = cross(dr, e2)
    det = dot(e1, pvec)
    if abs(det) <= F(0.0):
        return INF, F(0), F(0)
    sign = F(1.0) if det > 0 else (F(-1.0) if det < 0 else F(0.0))
    inv = F(F(1.0) / F(max(abs(det), F(0.000001)) * sign))
    tvec = (F(orig[0] - v0[0]), F(orig[1] - v0[1]), F(orig[2] - v0[2]))
    u = F(dot(tvec, pvec) * inv)
    if u < F(-0.00001) or u > F(1.00001):
        return INF, F(0), F(0)
    qvec = cross(tvec, e1)
    v = F(dot(dr, qvec) * inv)
    if v < F(-0.00001) or F(u + v) > F(1.00001):
        return INF, F(0), F(0)
    t = F(dot(e2, qvec) * inv)
    if greater_equal_f(t, F(0.0)):
        return t, u, v
    return INF, F(0), F(0)


def traverse(nodes, tris, M, origin, direct):
    pdata, boxes = nodes["pdata"], nodes["box"]
    M = np.asarray(M, np.float32).reshape(4, 4)
    st = {"predist": INF, "tri": LONGEST, "baked": [None] * BAKED, "count": 0, "tests": 0}
    direct = normalize(tuple(F(x) for x in direct))                                                   # :350
    origin = tuple(F(x) for x in origin)
    torig = tuple(F(F(F(F(M[i][0] * origin[0]) + F(M[i][1] * origin[1])) + F(M[i][2] * origin[2])) + M[i][3]) for i in range(3))   # :353
    tdir = tuple(F(F(F(F(M[0][i] * direct[0]) + F(M[1][i] * direct[1])) + F(M[2][i] * direct[2])) + M[3][i]) for i in range(3))    # :354
    dirlen = F(length(tdir) / F(max(length(direct), F(0.000001))))
    dirlen_inv = F(F(1.0) / F(max(dirlen, F(0.000001))))
    dirproj = normalize(tdir)
    d, near, far = cube_single(torig, dirproj, F(-0.00001), F(1.00001))                              # :365
    toffset = fmax(near, F(0.0))
    origined = tuple(F(torig[k] + F(dirproj[k] * toffset)) for k in range(3))                         # :372
    divident = tuple(F(F(1.0) / dirproj[k]) for k in range(3))

    def test_packed(t0, t1, valid0, valid1):                                                          # :261-309
        tri = [t0, t1]
        valid = [t0 >= 0 and t0 != LONGEST and t0 != st["tri"] and valid0,
                 t1 >= 0 and t1 != LONGEST and t1 != st["tri"] and valid1]
        valid[1] = valid[1] and tri[0] != tri[1]
        if not valid[0]:
            tri.reverse()
            valid.reverse()
        if not (valid[0] or valid[1]):
            return
        res = []
        for k in range(2):
            if valid[k] and tri[k] != LONGEST:
                st["tests"] += 1
                res.append(intersect_triangle(tris, origin, direct, tri[k]))
            else:
                res.append((INF, F(0), F(0)))
        for k in range(2):
            t, u, v = res[k]
            if valid[k] and less_f(t, INF) and less_equal_f(t, st["predist"]) and greater_equal_f(t, F(0.0)):
                if not equal_f(t, st["predist"]):
                    st["count"] = 0
                st["predist"] = t
                st["tri"] = tri[k]
                at = st["count"]
                st["count"] += 1
                if at < BAKED:                       # the reference writes past its 8 entries here (:294); canonical: dropped
                    st["baked"][at] = (u, v, t, tri[k])

    idx, esc, level, found = 0, -1, 0, -1
    valid_box = less_f(d, INF) and less_f(F(d * dirlen_inv), INF) and greater_equal_f(d, F(0.0))    # :377
    node = 0
    stack = [-1] * STACK
    ptr_def = 0
    skip_up, skip_int = False, False
    visits = 0
    for _ in range(8192):                                                                             # :383
        if not valid_box:
            break
        nx, ny = int(pdata[node][0]), int(pdata[node][1])
        not_leaf = nx != ny and valid_box
        if not_leaf:
            visits += 1
            lmn, lmx = unpack_box(boxes[nx])
            rmn, rmx = unpack_box(boxes[ny])
            hl, nl = cube_child(origined, divident, lmn, lmx)
            hr, nr = cube_child(origined, divident, rmn, rmx)
            left_near = less_equal_f(nl, nr)                                                          # :414
            lim = F(INF - PZERO)

            def general(hit, near_, child):                                                           # :416-430
                return (hit <= lim and F(hit * dirlen_inv) <= lim and hit > -PZERO and near_ <= lim and F(near_ * dirlen_inv) <= lim
                        and F(F(F(near_ + toffset) * dirlen_inv) - PZERO) <= st["predist"] and child != -1 and nx != ny
                        and (ny if left_near else nx) != esc and child != esc)
            og = [general(hl, nl, nx), general(hr, nr, ny)]
            lp = pdata[nx] if og[0] else np.array([-1, -1, -1, -1])
            rp = pdata[ny] if og[1] else np.array([-1, -1, -1, -1])
            is_node = [og[0] and lp[0] != lp[1], og[1] and rp[0] != rp[1]]
            is_leaf = [og[0] and lp[0] == lp[1], og[1] and rp[0] == rp[1]]
            any_leaf = is_leaf[0] or is_leaf[1]
            skip_int = skip_int or any_leaf
            if any_leaf:                                                                              # :441-448
                left_order = left_near if (is_leaf[0] and is_leaf[1]) else is_leaf[0]
                ov = is_leaf if left_order else is_leaf[::-1]
                first, second = (lp, rp) if left_order else (rp, lp)
                test_packed(int(first[3]), int(second[3]), ov[0], ov[1])
                skip_int = False
            any_overlap = is_node[0] or is_node[1]
            if any_overlap:                                                                           # :451-462
                left_order = left_near if (is_node[0] and is_node[1]) else is_node[0]
                lr = [nx if is_node[0] else -1, ny if is_node[1] else -1]
                if not left_order:
                    lr.reverse()
                if any_overlap and not skip_int:
                    if ptr_def < STACK and lr[1] != -1 and lr[0] != lr[1]:
                        stack[ptr_def] = lr[1]
                        ptr_def += 1
                    found = lr[0]
            skip_up = skip_up or any_overlap
        if not skip_int and valid_box:                                                                # :466-476
            if skip_up:
                ptr = ptr_def
            else:
                ptr_def -= 1
                ptr = ptr_def
            esc = -1 if skip_up else idx
            if ptr >= 0:
                if skip_up:
                    idx = found
                else:
                    idx = stack[ptr] if ptr < STACK else 0
                    if ptr < STACK:
                        stack[ptr] = -1
            else:
                idx = -1
            valid_box = valid_box and idx >= 0 and level >= 0 and ptr >= 0
            if valid_box:
                node = idx
            idx = idx if not valid_box else -1
        skip_up = False
    # choiceBaked -> reorderTriangles, :74-112
    n = min(st["count"], BAKED)
    bk = st["baked"][:n]
    for iround in range(1, n):
        for index in range(0, n - iround):
            a, b = bk[index], bk[index + 1]
            if a[3] <= b[3] or less_f(a[2], b[2]):
                bk[index], bk[index + 1] = b, a
    clean = []
    for iround in range(BAKED):
        if iround >= n - 1:
            break
        if bk[iround + 1][3] != bk[iround][3]:
            clean.append(bk[iround])
    if n > 0 and len(clean) <= BAKED:
        clean.append(bk[n - 1])
    return [(c[3], c[2], c[0], c[1]) for c in clean], visits, st["tests"]
