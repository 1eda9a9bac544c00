"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/psm_hip.h declares,
and the product path fails loudly (no CPU fallback) when there is no gfx950 device."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psm_hip.h")


def declared_symbols():
    """the entry points include/psm_hip.h declares"""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psm_[a-z0-9_]+)\s*\(", src)))


def test_header_compiles_as_c_and_cpp(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "psm_hip.h"\nint main(void){ psm_bvh_info i; psm_stats s; (void)i; (void)s; return sizeof(psm_material) == 128 ? 0 : 1; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++17")):
        exe = str(tmp_path / ("a_" + cc))
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c" if cc == "gcc" else "c++", str(c), "-o", exe])
        assert subprocess.call([exe]) == 0


def test_library_exports_every_declared_symbol(psm):
    lib = psm.lib()
    syms = declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "libpsm_hip.so does not export %s" % s
    # and the python binding lists exactly the header's entry points
    assert sorted(psm.EXPORTS) == syms
    # the schedules that lost (the tag experimental-r04 still has them) are gone from the sources
    assert "EXPERIMENTAL" not in open(HEADER).read()
    for gone in ("psm_arena_create", "psm_rt_traverse_group", "psm_rt_set_traverse_refill", "psm_lanes_render_grouped",
                 "psm_lanes_render_split", "psm_rt_share_texels"):
        assert not hasattr(lib, gone), gone


def test_every_entry_point_cites_the_reference():
    src = open(HEADER).read()
    for token in ("Radix.hpp:47-74", "TriangleHierarchy.inl:206-329", "Pipeline.inl:385-405", "Pipeline.inl:279-296",
                  "Pipeline.inl:251-277", "Utils.hpp:140-178", "MaterialSet.inl:13-23"):
        assert token in src, token


def test_no_device_fails_loudly(psm):
    if psm.lib().psm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(psm.PsmError):
        psm.Context(0)
    h = ctypes.c_void_p()
    assert psm.lib().psm_ctx_create(0, ctypes.byref(h)) < 0 and not h.value
    # NULL handles are rejected, never dereferenced
    assert psm.lib().psm_ctx_sync(None) < 0
    assert psm.lib().psm_bvh_build(None, None) < 0
    assert psm.lib().psm_rt_traverse(None, None) < 0


def test_product_does_not_use_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    banned = ("psmo_", "libpsm_oracle", "from oracle", "import oracle", "oracle/")
    for top in ("prismarine-core_amd", "include", "tools", "profiles"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    for b in banned:
                        assert b not in txt, "%s references %r" % (os.path.join(dirpath, f), b)


def test_dropin_headers_compile_and_link(tmp_path):
    """The GL-free Include/Prismarine look-alike compiles (with its own mini glm) and links against libpsm_hip.so."""
    exe = str(tmp_path / "viewer_order")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-DPSM_NO_SYSTEM_GLM",
                           os.path.join(ROOT, "tests", "cpp", "viewer_order.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "prismarine-core_amd"), "-lpsm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "prismarine-core_amd")])
    assert os.path.exists(exe)


def test_sharded_frame_batch_compiles_and_links(tmp_path):
    """psm::FrameBatch::renderSharded (the C++ mirror of the tile-sharded path, psm_dist_render_frames underneath) compiles
    warning-free against the C ABI's declarations and links against libpsm_hip.so (no GPU needed to build it)."""
    exe = str(tmp_path / "frame_batch_sharded")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-DPSM_NO_SYSTEM_GLM",
                           os.path.join(ROOT, "tests", "cpp", "frame_batch_sharded.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "prismarine-core_amd"), "-lpsm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "prismarine-core_amd")])
    assert os.path.exists(exe)
