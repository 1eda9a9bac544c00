#!/bin/bash
# round-2 GPU session B: throughput-mode A/B of the traversal schedules + SQ counters per schedule
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for cfg in "whole" "phased" "adaptive" "adaptive --trav-adaptive 24,8,4096,8,1048576" "adaptive --trav-adaptive 16,8,16384,4,1048576" "adaptive --trav-adaptive 16,8,4096,8,262144" "persistent"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02b_$tag.json 2> $OUT/r02b_$tag.err || { echo "bench $cfg failed"; tail -5 $OUT/r02b_$tag.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02b_$tag.json").read().strip().splitlines()[-1])
print("%-60s lanes4: %.3f ms/frame %.0f Mrays/s  (serial pass traverse %.3f ms)" % ("$cfg", d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["traverse"]))
PY
done
for cfg in "whole" "adaptive" "persistent"; do
  timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --lanes 8 --traverse $cfg > $OUT/r02b_l8_$cfg.json 2> $OUT/r02b_l8_$cfg.err || exit 1
  python - <<PY
import json
d = json.loads(open("$OUT/r02b_l8_$cfg.json").read().strip().splitlines()[-1])
print("%-60s lanes8: %.3f ms/frame %.0f Mrays/s" % ("$cfg", d["ms_per_step"], d["value"]))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/r02b_sq -- python3 $REPO/tools/trav_ab.py --configs whole,live16,live24,pt8 --reps 1 --rounds 1 > $OUT/r02b_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/r02b_sq2 -- python3 $REPO/tools/trav_ab.py --configs whole,live16,live24,pt8 --reps 1 --rounds 1 >> $OUT/r02b_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("r02b_sq", "r02b_sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rt_traverse" not in k: continue
            k = k.replace("void psm::", "").split("(")[0]
            key = (k, r["Counter_Name"])
            a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
        with open("$OUT/r02b_sq.txt", "a") as out:
            for (k, c), (n, v) in agg.items():
                line = "%-40s %-24s dispatches %4d total %.6g" % (k, c, n, v)
                print(line); out.write(line + "\n")
PY
