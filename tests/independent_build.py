"""An independent second opinion on the HLBVH build (test infrastructure).

Written from the reference's GLSL alone -- hlbvh/aabbmaker.comp:142-232 (splitLimit = 0: one box per triangle),
include/morton.glsl:37-51, hlbvh/build-new.comp:33-56,70-117 (findSplit, splitNode), hlbvh/child-link.comp:16-59,
hlbvh/refit.comp:21-114 -- in plain Python / numpy, NOT from oracle/psm_oracle.c. Morton codes by a bit loop, the
fp16 boxes by numpy's own float16 (round to nearest even), the tree by recursion instead of level queues. Canonical
rules it shares with the oracle by construction: leaf slots = rank among the non-degenerate triangles
in triangle order, M v summed as ((m0 x + m1 y) + m2 z) + m3, nodes numbered breadth first (root 0, the i-th internal
node of a level gets children base + 2 i, base + 2 i + 1).

leaves(tris, M) -> keys (n,) uint64, boxes (n, 2, 3) float16 [min, max], tri ids (n,)
tree(sorted_keys, sorted_slots, boxes, tri_ids) -> pdata (2n-1, 4) int32, box_min / box_max (2n-1, 3) float16
leaf_box(tri, M) -> (2, 3) float16: one triangle's leaf box (a refit-only update: tree() again with the build's keys and new boxes)
"""
import numpy as np

F = np.float32
PZERO = F(0.0005)


def morton3(x, y, z):  # morton.glsl:37-51, as a plain bit loop: x lowest
    code = 0
    for b in range(21):
        code |= ((int(x) >> b) & 1) << (3 * b)
        code |= ((int(y) >> b) & 1) << (3 * b + 1)
        code |= ((int(z) >> b) & 1) << (3 * b + 2)
    return code


def transform(M, v):  # mult4(transform, vec4(v, 1)).xyz, canonical sum order
    M = np.asarray(M, np.float32).reshape(4, 4)
    return np.array([F(F(F(F(M[i][0] * v[0]) + F(M[i][1] * v[1])) + F(M[i][2] * v[2])) + F(M[i][3] * F(1.0))) for i in range(3)], np.float32)


def leaves(tris, M):
    keys, boxes, ids = [], [], []
    third = F(0.33333333333333)
    for t in range(tris.shape[0]):
        v = [transform(M, tris[t][k]) for k in range(3)]                       # aabbmaker.comp:155-157
        c = (((v[0] + v[1]).astype(np.float32) + v[2]).astype(np.float32) * third).astype(np.float32)   # :159, left to right
        s = (np.abs(v[0] - c) + np.abs(v[1] - c)).astype(np.float32)
        s = (s + np.abs(v[2] - c)).astype(np.float32)
        if F(np.sqrt(F(F(F(s[0] * s[0]) + F(s[1] * s[1])) + F(s[2] * s[2])))) < F(1e-5):   # :160 degenerate: skipped
            continue
        mn = np.minimum(np.minimum(v[0], v[1]), v[2])                           # calcTriBox, :38-49
        mx = np.maximum(np.maximum(v[0], v[1]), v[2])
        # :176, greaterEqualF(branges, 0) = (range - 0) > -PZERO (mathlib.glsl:12) on all three axes: true for every finite box, false
        # for a NaN one (a singular fit transform: a scene so flat, so far out, that its padded extent rounds to zero). Round 5's
        # fuzzer found this reading without the test (tests/test_oracle_cpu.py::test_fuzzed_soups_oracle_against_the_independent_readings)
        if not all(F(F(mx[k] - mn[k]) - F(0.0)) > -PZERO for k in range(3)):
            continue
        q = np.floor(np.clip(c, F(0.0), F(0.99999)).astype(np.float32) * F(2097152.0))     # :187-189
        q = np.clip(q.astype(np.int64), 0, 0x1FFFFF)
        keys.append(morton3(q[0], q[1], q[2]))
        boxes.append([(mn - PZERO).astype(np.float32).astype(np.float16), (mx + PZERO).astype(np.float32).astype(np.float16)])  # :197-202
        ids.append(t)
    return np.array(keys, np.uint64), np.array(boxes, np.float16).reshape(-1, 2, 3), np.array(ids, np.int32)


def leaf_box(tri, M):
    """calcTriBox + the PZERO pad + packHalf (aabbmaker.comp:38-49,193-202) of ONE triangle: what a refit-only update
    recomputes for every leaf the build kept (no degeneracy test: that decides what a BUILD keeps)."""
    v = [transform(M, tri[k]) for k in range(3)]
    mn = np.minimum(np.minimum(v[0], v[1]), v[2])
    mx = np.maximum(np.maximum(v[0], v[1]), v[2])
    return np.array([(mn - PZERO).astype(np.float32).astype(np.float16), (mx + PZERO).astype(np.float32).astype(np.float16)], np.float16)


def nlz64(x):
    return 64 - int(x).bit_length()


def find_split(keys, first, last):  # build-new.comp:33-56
    fc, lc = int(keys[first]), int(keys[last])
    split = (first + last) >> 1
    if fc != lc:
        split = first
        common = nlz64(fc ^ lc)
        step = last - first
        while True:
            step = (step + 1) >> 1
            ns = split + step
            if ns < last and nlz64(fc ^ int(keys[ns])) > common:
                split = ns
            if step <= 1:
                break
    return min(max(split, first), last - 1)


def tree(keys, slots, boxes, tri_ids):
    """keys / slots: Morton codes and leaf slots in sorted order; boxes / tri_ids indexed by leaf slot."""
    n = len(keys)
    pdata = np.full((2 * n - 1, 4), -1, np.int32)
    bmin = np.zeros((2 * n - 1, 3), np.float16)
    bmax = np.zeros((2 * n - 1, 3), np.float16)
    level = [(0, 0, n - 1)]        # (node id, first, last)
    nxt = 1
    ranges = {}
    while level:
        new = []
        for nid, f, l in level:
            ranges[nid] = (f, l)
            if f == l:             # child-link.comp:34-53: a leaf node: (range end, range end, parent, triangle)
                slot = int(slots[f])
                pdata[nid][0] = pdata[nid][1] = l
                pdata[nid][3] = tri_ids[slot]
                bmin[nid], bmax[nid] = boxes[slot][0], boxes[slot][1]
                continue
            s = find_split(keys, f, l)
            a, b = nxt, nxt + 1    # splitNode: two consecutive ids, left = the lower key range (:77-115)
            nxt += 2
            pdata[nid][0], pdata[nid][1] = a, b
            pdata[a][2] = pdata[b][2] = nid
            new.append((a, f, s))
            new.append((b, s + 1, l))
        level = new
    for nid in range(2 * n - 2, -1, -1):   # refit.comp:91-98: children have larger ids than their parent
        if pdata[nid][0] != pdata[nid][1]:
            a, b = pdata[nid][0], pdata[nid][1]
            bmin[nid] = np.minimum(bmin[a], bmin[b])
            bmax[nid] = np.maximum(bmax[a], bmax[b])
    return pdata, bmin, bmax
