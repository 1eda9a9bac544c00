#!/usr/bin/env python3
"""Generates tests/golden/glm_host_formulas.npz: inputs and outputs of the reference's two host-side formulas of
the hot path, evaluated by the reference's OWN vendored glm (oracle/_ref/libglm_pin.so, built by `make -C oracle
ref` from oracle/ref_glm/glm_pin.cpp against /root/reference/External/include/glm):
  * fit transform (TriangleHierarchy.inl:226-232,257-267): bounds + optimisation matrix -> transform as uploaded
  * camera matrices (Pipeline.inl:279-312): eye, view, display size -> camInv, projInv as uploaded
These are the only outputs of the reference's own code that can be produced offline (the kernels are GLSL); they pin
oracle.fit_transform / inverse_opt, scenes.camera_matrices and include/Prismarine/psm_glm.hpp bit for bit.
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


if __name__ == "__main__":
    sys.exit(main())
