#!/usr/bin/env python3
"""Generates tests/golden/glm_host_formulas.npz: inputs and outputs of the reference's two host-side formulas of
the hot path, evaluated by the reference's OWN vendored glm (oracle/_ref/libglm_pin.so, built by `make -C oracle
ref` from oracle/ref_glm/glm_pin.cpp against /root/reference/External/include/glm):
  * fit transform (TriangleHierarchy.inl:226-232,257-267): bounds + optimisation matrix -> transform as uploaded
  * camera matrices (Pipeline.inl:279-312): eye, view, display size -> camInv, projInv as uploaded
These are the only outputs of the reference's own code that can be produced offline (the kernels are GLSL); they pin
oracle.fit_transform / inverse_opt, scenes.camera_matrices and include/Prismarine/psm_glm.hpp bit for bit.
And tests/golden/glm_gltf_transforms.npz, the same for the viewer's scene loading (Source/Examples/Viewer.cpp:240-258):
  * node transforms: parent * (matrix * T * S * R) in glm doubles, and the root scale(mscale)
  * what TriangleArrayInstance::setTransform uploads for them (VertexInstance.inl:54-58): transpose(mat4(t)), inverse(mat4(t))
which pins prismarine-core_amd/gltf.py's node_transform / root_transform / mesh_transform.
Needs /root/reference; the fixture it writes travels with the repo."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))


def pin_lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    return C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libglm_pin.so"))


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def cases(seed=20260204, n_fit=64, n_cam=64):
    rng = np.random.RandomState(seed)
    mn = rng.uniform(-50, 0, (n_fit, 4)).astype(np.float32)
    mx = (mn + rng.uniform(0.01, 100, (n_fit, 4))).astype(np.float32)
    mn[0], mx[0] = (-15.00001, -0.50001, -9.00001, 0), (15.00001, 12.00001, 9.00001, 0)   # Sponza-like extents
    opt = np.tile(np.eye(4), (n_fit, 1, 1))
    for t in range(1, n_fit, 2):  # every other case: rotation about y, anisotropic scale, translation (row-major)
        a = rng.uniform(0, 3)
        opt[t] = [[np.cos(a), 0, np.sin(a), rng.uniform(-2, 2)], [0, rng.uniform(0.5, 2), 0, -1.0],
                  [-np.sin(a), 0, np.cos(a), 2.0], [0, 0, 0, 1]]
    eye = rng.uniform(-20, 20, (n_cam, 3)).astype(np.float32)
    view = rng.uniform(-5, 5, (n_cam, 3)).astype(np.float32)
    eye[0], view[0] = (0, 6, 6), (0, 2, 0)   # the reference's default camera (Application.hpp:101-102)
    size = np.array([[(1920, 1080), (1280, 720), (3840, 2160), (160, 90), (256, 256)][t % 5] for t in range(n_cam)], np.int32)
    return mn, mx, opt, eye, view, size


def main():
    L = pin_lib()
    mn, mx, opt, eye, view, size = cases()
    n_fit, n_cam = mn.shape[0], eye.shape[0]
    transform = np.zeros((n_fit, 16), np.float32)
    transform_inv = np.zeros((n_fit, 16), np.float32)
    first_pass = np.zeros((n_fit, 16), np.float32)
    for t in range(n_fit):
        o = np.ascontiguousarray(opt[t].T.reshape(16))  # glm memory order is column-major
        L.glm_pin_fit(p(mn[t]), p(mx[t]), p(o), p(transform[t]), p(transform_inv[t]))
        L.glm_pin_inverse_opt(p(o), p(first_pass[t]))
    cam_inv = np.zeros((n_cam, 16), np.float32)
    proj_inv = np.zeros((n_cam, 16), np.float32)
    for t in range(n_cam):
        L.glm_pin_camera(p(eye[t]), p(view[t]), int(size[t, 0]), int(size[t, 1]), p(cam_inv[t]), p(proj_inv[t]))
    np.savez_compressed(os.path.join(HERE, "glm_host_formulas.npz"), mn=mn, mx=mx, opt=opt, transform=transform,
                        transform_inv=transform_inv, first_pass=first_pass, eye=eye, view=view, size=size,
                        cam_inv=cam_inv, proj_inv=proj_inv)
    print("wrote glm_host_formulas.npz: %d fit cases, %d camera cases" % (n_fit, n_cam))
    gltf_cases(L)


def gltf_cases(L, seed=20261005, n=96):
    """Random node descriptions (every combination of matrix / translation / scale / rotation present or absent) under random
    parents, plus the nodes of tests/golden/gltf/court.gltf under the identity."""
    import json
    rng = np.random.RandomState(seed)
    L.glm_pin_gltf_root.argtypes = [C.c_double, C.c_void_p]
    parent = np.zeros((n, 16)); has = np.zeros((n, 4), np.int32)
    matrix = np.zeros((n, 16)); trans = np.zeros((n, 3)); scale = np.ones((n, 3)); rot = np.zeros((n, 4))
    for t in range(n):
        a, b = rng.uniform(0, 6.28, 2)
        pm = np.array([[np.cos(a), 0, np.sin(a), rng.uniform(-3, 3)], [0, rng.uniform(0.2, 3), 0, rng.uniform(-3, 3)],
                       [-np.sin(a), 0, np.cos(a), rng.uniform(-3, 3)], [0, 0, 0, 1]]) if t % 3 else np.eye(4)
        parent[t] = pm.T.reshape(16)                                   # glm memory: column-major
        has[t] = [(t >> k) & 1 for k in range(4)] if t < 16 else rng.randint(0, 2, 4)
        mm = np.array([[np.cos(b), -np.sin(b), 0.1, rng.uniform(-2, 2)], [np.sin(b), np.cos(b), 0, rng.uniform(-2, 2)],
                       [0, 0.2, rng.uniform(0.5, 2), rng.uniform(-2, 2)], [0, 0, 0, 1]])
        matrix[t] = mm.T.reshape(16)
        trans[t] = rng.uniform(-5, 5, 3)
        scale[t] = rng.uniform(0.1, 4, 3) * rng.choice([-1, 1], 3)
        q = rng.normal(size=4)
        rot[t] = q / np.linalg.norm(q) if t % 5 else q                 # every fifth: not normalised (the viewer does not care)
    court = json.load(open(os.path.join(HERE, "gltf", "court.gltf")))["nodes"]
    extra = len(court)
    parent = np.concatenate([parent, np.tile(np.eye(4).reshape(16), (extra, 1))])
    has = np.concatenate([has, np.zeros((extra, 4), np.int32)])
    matrix = np.concatenate([matrix, np.zeros((extra, 16))]); trans = np.concatenate([trans, np.zeros((extra, 3))])
    scale = np.concatenate([scale, np.ones((extra, 3))]); rot = np.concatenate([rot, np.zeros((extra, 4))])
    for k, nd in enumerate(court):
        t = n + k
        if "matrix" in nd:
            has[t, 0], matrix[t] = 1, nd["matrix"]
        else:
            for j, key, arr in ((1, "translation", trans), (2, "scale", scale), (3, "rotation", rot)):
                if key in nd:
                    has[t, j], arr[t] = 1, nd[key]
    total = n + extra
    out = np.zeros((total, 16)); tf = np.zeros((total, 16), np.float32); tfi = np.zeros((total, 16), np.float32)
    for t in range(total):
        hm = bool(has[t, 0])   # tinygltf: T / R / S are read only where there is no matrix
        L.glm_pin_gltf_node(p(parent[t]), p(matrix[t]) if hm else None, p(trans[t]) if has[t, 1] and not hm else None,
                            p(scale[t]) if has[t, 2] and not hm else None, p(rot[t]) if has[t, 3] and not hm else None, p(out[t]))
        L.glm_pin_mesh_transform(p(out[t]), p(tf[t]), p(tfi[t]))
    mscale = np.array([1.0, 0.01, 2.5, 100.0, 1.0 / 3.0])
    roots = np.zeros((mscale.size, 16))
    for t in range(mscale.size):
        L.glm_pin_gltf_root(C.c_double(mscale[t]), p(roots[t]))
    np.savez_compressed(os.path.join(HERE, "glm_gltf_transforms.npz"), parent=parent, has=has, matrix=matrix, translation=trans,
                        scale=scale, rotation=rot, transform=out, mesh_transform=tf, mesh_transform_inv=tfi, mscale=mscale, roots=roots)
    print("wrote glm_gltf_transforms.npz: %d node cases, %d root scales" % (total, mscale.size))


if __name__ == "__main__":
    sys.exit(main())
