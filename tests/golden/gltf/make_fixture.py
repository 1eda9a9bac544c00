#!/usr/bin/env python3
"""Writes tests/golden/gltf/court.gltf + court.bin + two PNGs (+ their pre-decoded .npy): a glTF 2.0 scene of this repo's own
that holds one of everything the reference's viewer does with a glTF file (Source/Examples/Viewer.cpp:66-279) -- and of what it
does not do. The reference ships no glTF file (its viewer takes one on the command line), so there is no file of its authors' to
load; the expected node transforms come from the reference's vendored glm (tests/golden/make_glm_golden.py), everything else is
checked against the oracle's loader restatement and an independent numpy evaluation of the file (tests/test_gltf_cpu.py).

  buffer 0 (court.bin)
    box      24 vertices INTERLEAVED (position | normal | texcoord, byteStride 32), 16-bit indices
    floor    9 x 9 grid, PLANAR views without byteStride (tightly packed), 32-bit indices
    pyramid  positions with byteStride 16 (4 bytes of padding per vertex), NO normals (face-normal fallback,
             loader.comp:101-113), 16-bit indices at a byte offset that is not a multiple of 4
    lone     three vertices WITHOUT indices (the viewer loads nodeCount = 1), and a LINES primitive (never loaded)
  buffer 1 (a base64 data URI)  pillar: a second box, so that primitives read different buffers
  nodes    T / R / S with a non-uniform scale (the viewer composes T * S * R), a `matrix` node, an instance of the box, a node with
           a mesh AND children (its children are not visited), a chain deeper than the walk goes, a second scene (not loaded)
  materials  factors only; base colour + normal texture; emissive; an empty material; one primitive without material
  textures   two images, one used by two textures (one slot), one that does not exist (slot 0)
"""
import base64
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def box(sx=0.5, sy=0.5, sz=0.5):
    """24 vertices (position, normal, uv), 12 triangles."""
    P, N, T, I = [], [], [], []
    axes = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
            ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    s = np.array([sx, sy, sz])
    for n, u, v in axes:
        n, u, v = np.array(n, float), np.array(u, float), np.array(v, float)
        b = len(P)
        for (a, c) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            P.append((n + a * u + c * v) * s)
            N.append(n)
            T.append(((a + 1) / 2 * 1.5 - 0.25, (c + 1) / 2 * 1.5 - 0.25))   # beyond [0, 1]: GL_REPEAT
        I += [b, b + 1, b + 2, b, b + 2, b + 3]
    return np.array(P, np.float32), np.array(N, np.float32), np.array(T, np.float32), np.array(I, np.uint32)


def grid(n=8):
    xs = np.linspace(-1, 1, n + 1)
    P = np.array([(x, 0.02 * np.sin(5 * x) * np.cos(4 * z), z) for z in xs for x in xs], np.float32)
    N = np.tile(np.array((0, 1, 0), np.float32), (P.shape[0], 1))
    T = np.array([((x + 1) * 2, (z + 1) * 2) for z in xs for x in xs], np.float32)
    I = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            I += [a, a + n + 1, a + 1, a + 1, a + n + 1, a + n + 2]
    return P, N, T, np.array(I, np.uint32)


def pyramid():
    P = np.array([(-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5), (0, 0.9, 0)], np.float32)
    I = np.array([0, 4, 1, 1, 4, 2, 2, 4, 3, 3, 4, 0, 0, 1, 2, 0, 2, 3], np.uint32)
    return P, I


class Buf:
    def __init__(self):
        self.b = bytearray()
        self.views = []

    def view(self, raw, stride=None, align=4):
        while len(self.b) % align:
            self.b += b"\0"
        v = {"buffer": 0, "byteOffset": len(self.b), "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        self.b += raw
        self.views.append(v)
        return len(self.views) - 1


def main():
    buf = Buf()
    acc = []

    def accessor(view, ctype, count, typ, off=0, **kw):
        a = dict({"bufferView": view, "componentType": ctype, "count": count, "type": typ}, **kw)
        if off:
            a["byteOffset"] = off
        acc.append(a)
        return len(acc) - 1

    F, U16, U32 = 5126, 5123, 5125
    # box: interleaved
    P, N, T, I = box()
    inter = np.concatenate([P, N, T], 1).astype(np.float32)
    v_box = buf.view(inter.tobytes(), stride=32)
    a_box_p = accessor(v_box, F, 24, "VEC3", min=P.min(0).tolist(), max=P.max(0).tolist())
    a_box_n = accessor(v_box, F, 24, "VEC3", off=12)
    a_box_t = accessor(v_box, F, 24, "VEC2", off=24)
    v_box_i = buf.view(I.astype(np.uint16).tobytes())
    a_box_i = accessor(v_box_i, U16, I.size, "SCALAR")
    # floor: planar
    P, N, T, I = grid()
    v_fp, v_fn, v_ft = buf.view(P.tobytes()), buf.view(N.tobytes()), buf.view(T.tobytes())
    a_fp = accessor(v_fp, F, P.shape[0], "VEC3", min=P.min(0).tolist(), max=P.max(0).tolist())
    a_fn = accessor(v_fn, F, P.shape[0], "VEC3")
    a_ft = accessor(v_ft, F, P.shape[0], "VEC2")
    v_fi = buf.view(I.tobytes())
    a_fi = accessor(v_fi, U32, I.size, "SCALAR")
    # pyramid: stride 16, indices at 2 mod 4
    P, I = pyramid()
    padded = np.concatenate([P, np.full((P.shape[0], 1), 77.0, np.float32)], 1)
    v_pp = buf.view(padded.tobytes(), stride=16)
    a_pp = accessor(v_pp, F, P.shape[0], "VEC3", min=P.min(0).tolist(), max=P.max(0).tolist())
    v_pi = buf.view(struct.pack("<H", 0xBEEF) + I.astype(np.uint16).tobytes())
    a_pi = accessor(v_pi, U16, I.size, "SCALAR", off=2)
    # lone triangle (no indices) and a line strip
    L = np.array([(-0.3, 0.0, 0.0), (0.3, 0.0, 0.0), (0.0, 0.5, 0.0), (0.0, 0.5, 0.4), (0.3, 0.0, 0.4), (-0.3, 0.0, 0.4)], np.float32)
    v_lp = buf.view(L.tobytes())
    a_lp = accessor(v_lp, F, 6, "VEC3", min=L.min(0).tolist(), max=L.max(0).tolist())
    # buffer 1: pillar, embedded
    P, N, T, I = box(0.15, 1.0, 0.15)
    b1 = bytearray()
    views1 = []
    for raw in (P.tobytes(), N.tobytes(), I.astype(np.uint16).tobytes()):
        while len(b1) % 4:
            b1 += b"\0"
        views1.append({"buffer": 1, "byteOffset": len(b1), "byteLength": len(raw)})
        b1 += raw
    nv0 = len(buf.views)
    a_qp = accessor(nv0 + 0, F, 24, "VEC3", min=P.min(0).tolist(), max=P.max(0).tolist())
    a_qn = accessor(nv0 + 1, F, 24, "VEC3")
    a_qi = accessor(nv0 + 2, U16, I.size, "SCALAR")

    h = np.sqrt(0.5)
    g = {
        "asset": {"version": "2.0", "generator": "tests/golden/gltf/make_fixture.py"},
        "scene": 0,
        "scenes": [{"nodes": [0]}, {"nodes": [5]}],
        "nodes": [
            {"name": "root", "children": [1, 2, 3, 4, 8, 9], "scale": [1.0, 1.0, 1.0]},
            {"name": "floor", "mesh": 1, "translation": [0.0, -1.0, 0.0], "scale": [6.0, 1.0, 6.0]},
            {"name": "box", "mesh": 0, "translation": [-1.5, -0.25, 0.5], "rotation": [0.0, 0.25881904510252074, 0.0, 0.9659258262890683],
             "scale": [1.0, 1.5, 0.75]},
            {"name": "pyramid", "mesh": 2, "children": [5],
             "matrix": [0.8, 0.0, -0.6, 0.0, 0.1, 1.2, 0.0, 0.0, 0.6, 0.0, 0.8, 0.0, 1.6, -1.0, -0.7, 1.0]},
            {"name": "carrier", "translation": [0.2, 0.0, -2.2], "children": [6]},
            {"name": "never: child of a mesh node", "mesh": 0, "translation": [0.0, 3.0, 0.0]},
            {"name": "turn", "rotation": [0.0, h, 0.0, h], "children": [7]},
            {"name": "pillar", "mesh": 3, "translation": [0.5, 0.0, 0.25]},
            {"name": "box again", "mesh": 0, "translation": [2.1, -0.5, -0.4], "scale": [0.8, 0.8, 0.8]},
            {"name": "chain 1", "translation": [0.0, 0.1, 0.0], "children": [10]},
            {"name": "chain 2", "translation": [0.0, 0.1, 0.0], "children": [11]},
            {"name": "chain 3", "translation": [0.0, 0.1, 0.0], "children": [12]},
            {"name": "chain 4: deeper than the viewer walks", "mesh": 0, "translation": [-3.0, 0.0, -2.0]},
        ],
        "meshes": [
            {"name": "box", "primitives": [{"attributes": {"POSITION": a_box_p, "NORMAL": a_box_n, "TEXCOORD_0": a_box_t},
                                            "indices": a_box_i, "material": 0}]},
            {"name": "floor", "primitives": [{"attributes": {"TEXCOORD_0": a_ft, "POSITION": a_fp, "NORMAL": a_fn, "COLOR_0": a_fn},
                                              "indices": a_fi, "material": 1, "mode": 4}]},
            {"name": "pyramid + lone triangle + lines", "primitives": [
                {"attributes": {"POSITION": a_pp}, "indices": a_pi, "material": 2},
                {"attributes": {"POSITION": a_lp}},
                {"attributes": {"POSITION": a_lp}, "mode": 1, "material": 0}]},
            {"name": "pillar", "primitives": [{"attributes": {"POSITION": a_qp, "NORMAL": a_qn}, "indices": a_qi, "material": 3}]},
        ],
        "materials": [
            {"name": "brick", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.3, 0.2, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
            {"name": "tiles", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.1, "roughnessFactor": 0.8},
             "normalTexture": {"index": 1, "scale": 1.0}, "doubleSided": True},
            {"name": "lamp", "pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.1, 0.1, 1.0], "roughnessFactor": 1.0, "metallicFactor": 0.0},
             "emissiveFactor": [4.0, 3.0, 2.0], "emissiveTexture": {"index": 3}},
            {"name": "default"},
            {"name": "unused", "pbrMetallicRoughness": {"baseColorTexture": {"index": 2}}},
        ],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1}, {"source": 0}, {"source": 2}],
        "images": [{"uri": "court_tiles.png"}, {"uri": "court_bumps.png"}, {"uri": "court_missing.png"}],
        "samplers": [{"magFilter": 9728, "wrapS": 33071}],
        "accessors": acc,
    }
    g["bufferViews"] = buf.views + views1
    g["buffers"] = [{"uri": "court.bin", "byteLength": len(buf.b)},
                    {"uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(b1)).decode(), "byteLength": len(b1)}]
    open(os.path.join(HERE, "court.bin"), "wb").write(bytes(buf.b))
    json.dump(g, open(os.path.join(HERE, "court.gltf"), "w"), indent=1)

    # images: RGBA8, top row first as files have them
    from PIL import Image
    y, x = np.mgrid[0:32, 0:48]
    tiles = np.zeros((32, 48, 4), np.uint8)
    tiles[..., 0] = np.where(((x // 6) + (y // 4)) % 2, 210, 60) + (y * 1).astype(np.uint8)
    tiles[..., 1] = np.where(((x // 6) + (y // 4)) % 2, 190, 70)
    tiles[..., 2] = 40 + 4 * x
    tiles[..., 3] = 255
    bumps = np.zeros((16, 16, 4), np.uint8)
    yy, xx = np.mgrid[0:16, 0:16]
    bumps[..., 0] = 128 + (60 * np.sin(xx * 0.8)).astype(int)
    bumps[..., 1] = 128 + (60 * np.cos(yy * 0.7)).astype(int)
    bumps[..., 2] = 230
    bumps[..., 3] = 255
    for name, a in (("court_tiles.png", tiles), ("court_bumps.png", bumps)):
        Image.fromarray(a, "RGBA").save(os.path.join(HERE, name))
        back = np.asarray(Image.open(os.path.join(HERE, name)).convert("RGBA"))
        assert np.array_equal(back, a)
        np.save(os.path.join(HERE, name + ".npy"), np.ascontiguousarray(back))
    print("court.gltf: %d bytes of buffer 0, %d of buffer 1, %d accessors" % (len(buf.b), len(b1), len(acc)))


if __name__ == "__main__":
    main()
