#!/usr/bin/env python3
"""Instruction counts of the traversal kernels' node step, from hipcc's assembly (no GPU needed).
usage: python tools/step_isa.py [trace.s]   (default: compiles prismarine-core_amd/csrc/trace.hip to /tmp)
The step is taken as the innermost loop region from the block that issues the node record's loads (the first
v_fma_mix_f32 of the kernel's hot loop) to the wave-level decision's back branch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _walk(body, start, skip):
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    i, valu, salu, br, mem = start, 0, 0, 0, 0
    seen_counts = took_continue = False
    for _ in range(2000):
        l = body[i]
        if l.startswith("\tv_"):
            valu += 1
        elif l.startswith("\ts_") and not l.startswith("\ts_waitcnt") and not l.startswith("\ts_nop"):
            salu += 1
        elif re.match(r"\t(ds_|global_|scratch_|buffer_)", l):
            mem += 1
        if "s_bcnt1" in l:
            seen_counts = True
        m = re.match(r"\ts_(cbranch_\w+|branch) (\.LBB\w+)", l)
        if m:
            br += 1
            kind, tgt = m.group(1), m.group(2)
            take = kind in ("branch", "cbranch_execnz")
            if kind.startswith("cbranch_scc") or kind.startswith("cbranch_vcc"):
                take = seen_counts and not took_continue
                if take and skip > 0:
                    skip -= 1
                    take = False
                took_continue = took_continue or take
            if take:
                i = labels[tgt]
                if i <= start <= i + 3:
                    return (valu, salu, br, mem), True
                continue
        i += 1
        if i == start:
            return (valu, salu, br, mem), True
    return (valu, salu, br, mem), False


def walk(body, start):
    """Follow the hot path of one wave-step: from the block that loads the node record, through the divergent regions
    (s_cbranch_execz falls through: somebody is in them), taking the wave-level `continue` and every execnz / unconditional
    branch, until the walk is back at the start. The `continue` is the first scalar conditional branch after the lane counts
    that leads back to the start: since round 4 the exit to the solo gear (at most solo_max lanes with work) may stand in
    front of it, so the walk is tried with 0, 1, 2 such branches passed by and the first one that closes the loop counts."""
    for skip in range(3):
        counts, closed = _walk(body, start, skip)
        if closed:
            return counts
    return counts


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/psm_trace_step.s"
        src = os.path.join(ROOT, "prismarine-core_amd", "csrc", "trace.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", path, src], stderr=subprocess.DEVNULL)
    L = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"^_ZN3psm\w+:", l)]
    ends = [i for i, l in enumerate(L) if l.startswith(".Lfunc_end")]
    for (st, name), en in zip(starts, ends):
        body = L[st:en]
        idx = [i for i, l in enumerate(body) if "v_fma_mix_f32" in l]
        if not idx:
            continue
        a = idx[0]
        while a > 0 and not body[a].startswith(".LBB"):
            a -= 1
        b = idx[-1]
        while b < len(body) - 1 and not re.search(r"s_cbranch_vccn?z", body[b]):
            b += 1
        seg = body[a:b + 1]
        valu = [l.split()[0] for l in seg if l.startswith("\tv_")]
        salu = sum(1 for l in seg if l.startswith("\ts_") and not l.startswith("\ts_waitcnt") and not l.startswith("\ts_nop"))
        br = sum(1 for l in seg if l.startswith("\ts_cbranch"))
        mem = sum(1 for l in seg if re.match(r"\t(ds_|global_|scratch_|buffer_)", l))
        scr = sum(1 for l in seg if "scratch_" in l)
        hv, hs, hb, hm = walk(body, a)
        print("%-62s hot path of one wave-step: VALU %3d  SALU %3d (of them %d branches)  memory %d" % (name.replace("_ZN3psm", "")[:62], hv, hs, hb, hm))
        print("%-62s step lines %5d-%5d: VALU %3d  SALU %3d  branches %2d  memory %d (scratch %d); kernel scratch ops %d" % (
            name.replace("_ZN3psm", "")[:62], st + a + 1, st + b + 1, len(valu), salu, br, mem, scr, sum(1 for l in body if "scratch_" in l)))


if __name__ == "__main__":
    main()
