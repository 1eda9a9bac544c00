#!/usr/bin/env python3
"""Fills tests/golden/obj/ (run in the container that has /root/reference and PIL; the GPU box has neither).

  box, rbox, sphere, Cow  .obj + .mtl   DATA files of the reference (Resources/toys/): models somebody else wrote -- v / vt / vn
                                        faces, quads, four materials, `g` / `s` / `o` groups, Ka / Ns / illum / d statements --
                                        copied byte for byte. The reference's viewer loads glTF (Source/Examples/Viewer.cpp:66-69),
                                        so nothing in the reference reads them: they are inputs without expected outputs.
  shelf .obj + .mtl + three PNGs        a textured model of this repo's own (the open Cornell box through scenes.write_obj with
                                        box-projected texcoords): map_Kd / map_Bump / map_Ke / map_Ks, so that the path
                                        file -> TextureSet slots -> surface.comp:81-161 is fed from files.
  *.png.npy                             every PNG decoded with PIL HERE into uint8 [h, w, 4] (scenes.load_image_rgba8 prefers
                                        it): nothing is decoded on the GPU box.
"""
import importlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
scenes = importlib.import_module("prismarine-core_amd.scenes")

REF = "/root/reference/Resources/toys"
for stem in ("box", "rbox", "sphere", "Cow"):
    for ext in (".obj", ".mtl"):
        shutil.copyfile(os.path.join(REF, stem + ext), os.path.join(HERE, stem + ext))
        os.chmod(os.path.join(HERE, stem + ext), 0o644)

sc = scenes.textured(scenes.cornell(open_top=True))
scenes.write_obj(os.path.join(HERE, "shelf.obj"), sc)
from PIL import Image
for f in sorted(os.listdir(HERE)):
    if f.endswith(".png"):
        a = np.asarray(Image.open(os.path.join(HERE, f)).convert("RGBA"))
        np.save(os.path.join(HERE, f + ".npy"), np.ascontiguousarray(a))
        print(f, a.shape)
