"""Code-generation guards that need no GPU: hipcc's gfx950 assembly of the traversal kernels (tools/step_isa.py).
The node step's hot path is the frame's critical resource: a compiler flag lost from the Makefile
(-fno-slp-vectorize), a flag that moves back into a lane mask or a spill that lands in the loop shows here first."""
import importlib.util
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asm(tmp_path):
    flags = open(os.path.join(ROOT, "prismarine-core_amd", "csrc", "Makefile")).read()
    cxx = re.search(r"^CXXFLAGS := (.*)$", flags, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    assert "-fno-slp-vectorize" in cxx and "-ffp-contract=off" in cxx
    out = str(tmp_path / "trace.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + [f for f in cxx if not f.startswith("-W")] +
                          ["-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "prismarine-core_amd", "csrc", "trace.hip")],
                          stderr=subprocess.DEVNULL)
    return out


def test_traversal_step_stays_lean_and_out_of_scratch(tmp_path):
    path = _asm(tmp_path)
    spec = importlib.util.spec_from_file_location("step_isa", os.path.join(ROOT, "tools", "step_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN3psm\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    seen = 0
    for (st, name), en in zip(starts, ends):
        if "rt_traverseILb0ELb0ELb1E" not in name and "rt_traverseILb0ELb0ELb0E" not in name:
            continue   # the timed kernels: hand-over and single launch, plain
        body = lines[st:en]
        first = [i for i, l in enumerate(body) if "v_fma_mix_f32" in l][0]
        a = first
        while not body[a].startswith(".LBB"):
            a -= 1
        valu, salu, branches, mem = mod.walk(body, a)
        assert 60 <= valu <= 84, (name, valu)        # 80 at the end of round 3 (94 at the end of round 2)
        assert salu <= 54, (name, salu)              # hand-over 50 / single launch 37: the exit to the solo gear is in, the parked / left-near flags are no longer lane masks (round 3: 61 / 45; round 2: 86)
        assert branches <= 6, (name, branches)       # 5 / 4
        assert mem == 4, (name, mem)                 # two record loads, the stack's push and pop: no scratch in the step
        assert not any("v_pk_" in l for l in body), name   # no packed-fp32 pairs (the SLP vectoriser's)
        seen += 1
    assert seen == 2
    # occupancy: 64 VGPRs at most (8 waves per SIMD), nothing spilled
    meta = open(path).read()
    for kern in ("_ZN3psm11rt_traverseILb0ELb0ELb1EEEvNS_8TravArgsE", "_ZN3psm11rt_traverseILb0ELb0ELb0EEEvNS_8TravArgsE"):
        blk = meta[meta.index(".name:           " + kern):]
        blk = blk[:blk.index(".wavefront_size")]
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 64, kern
        # (round 4's hand-over kernel parked two vector registers of its prologue in scratch; one of them was `tid & ~63`, which is 0
        # with one wave per workgroup -- the launch bound did not tell the optimiser, a __builtin_assume does: no spill in either kernel)
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0, kern
        # (one scalar of the hand-over kernel's prologue is parked in a vector lane since the solo gear came: lines 105 / 244 of
        # its assembly, long before the loop; the walk above would count a reload inside the step)
        assert int(re.search(r"\.sgpr_spill_count:\s+(\d+)", blk).group(1)) <= 1, kern


def test_solo_gear_step_is_short(tmp_path):
    """The solo gear's step (trace.hip: solo_ray): one ray on all lanes of the wave. A lone wave pays ~10 cycles for every
    instruction it issues (profiles/r04_solo_step.txt), so the step's worth is its instruction count: one word of the record
    per lane, two v_fma_mix, quad-permute DPP moves, the decisions in scalar registers -- about half of the lane-per-ray step
    (80 vector + 49-63 scalar instructions) and free of scratch and LDS traffic."""
    path = _asm(tmp_path)
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN3psm\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    seen = 0
    for (st, name), en in zip(starts, ends):
        if "rt_traverseILb0ELb0ELb1E" not in name and "rt_traverseILb0ELb0ELb0E" not in name:
            continue
        body = lines[st:en]
        dpp = [i for i, l in enumerate(body) if "quad_perm:" in l]
        assert len(dpp) == 5 and sum(1 for l in body if "row_shl:4" in l) == 1, (name, len(dpp))   # the neighbour's word, two moves each for tNear and tFar, and the right child's near value: once in the kernel
        # the straight-line part of the step: from the record's load to the first branch after the reductions
        a = dpp[0]
        while "global_load_dword " not in body[a]:
            a -= 1
        b = dpp[-1]
        while not re.match(r"\ts_cbranch", body[b]):
            b += 1
        seg = body[a:b + 1]
        valu = sum(1 for l in seg if l.startswith("\tv_"))
        salu = sum(1 for l in seg if l.startswith("\ts_") and not l.startswith("\ts_waitcnt") and not l.startswith("\ts_nop"))
        mem = sum(1 for l in seg if re.match(r"\t(ds_|global_|scratch_|buffer_)", l))
        assert valu <= 30, (name, valu)              # 27
        assert salu <= 20, (name, salu)              # 16 up to the push / pop branches
        assert mem == 1, (name, mem)                 # the record: one word per lane
        seen += 1
    assert seen == 2
