#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 5 120 tools/ubench/valu_rate > $OUT/r02d_valu_rate.txt 2>&1; cat $OUT/r02d_valu_rate.txt
for cfg in whole adaptive; do
  PSM_LANES_PROFILE=1 timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02d_prof_$cfg.json 2> $OUT/r02d_prof_$cfg.err
  grep psm_lanes_render $OUT/r02d_prof_$cfg.err | tail -2
done
cd /tmp && export TMPDIR=/tmp
for cfg in whole adaptive; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/r02d_sq_$cfg -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 --traverse $cfg > /dev/null 2> $OUT/r02d_sq_$cfg.err
f=$(find $OUT/r02d_sq_$cfg -name '*counter_collection.csv' | head -1)
python3 $REPO/profiles/summarize.py pmc $f > $OUT/r02d_sq_$cfg.txt
python3 - <<PY
import csv, collections
tot = collections.Counter()
for r in csv.DictReader(open("$f")):
    tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("$cfg totals over the run (all kernels):", {k: "%.4g" % v for k, v in tot.items()})
PY
done
