#!/bin/bash
# SQ counters of the traversal kernel on the primary + second-round ray sets (tools/trav_bench.py).
# usage: tools/pmc_trav.sh <tag>   (through gpurun) -> gpurun_out/<tag>_sq.txt
set -e
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export REPS=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $REPO/gpurun_out/${TAG}_sq -- python $REPO/tools/trav_bench.py > $REPO/gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $REPO/gpurun_out/${TAG}_sq2 -- python $REPO/tools/trav_bench.py >> $REPO/gpurun_out/${TAG}_sq.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("${TAG}_sq", "${TAG}_sq2"):
    for f in glob.glob("$REPO/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            k = r["Kernel_Name"][:40]
            if "rt_traverse" not in k: continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
        disp = sorted({k[0] for k in agg}, key=int)
        with open("$REPO/gpurun_out/${TAG}_sq.txt", "a") as out:
            for dsp in disp:
                line = "dispatch %s: " % dsp + "  ".join("%s=%.4g" % (c, v) for (d2, c), v in agg.items() if d2 == dsp)
                print(line); out.write(line + "\n")
PY
