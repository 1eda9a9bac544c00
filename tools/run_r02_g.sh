#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
echo "== base lib"; timeout -k 10 200 python tools/trav_ab.py --configs whole,live16,live24 --reps 5 --rounds 2 2>&1 | tail -4
echo "== +40 half-rate VALU per box step"; PSM_HIP_LIB=$REPO/tools/ubench/libpsm_hip_x40.so timeout -k 10 200 python tools/trav_ab.py --configs whole,live16,live24 --reps 5 --rounds 2 2>&1 | tail -4
for lib in base x40; do
  L=""; [ $lib = x40 ] && L="PSM_HIP_LIB=$REPO/tools/ubench/libpsm_hip_x40.so"
  for cfg in whole adaptive; do
    env $L timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02g_${lib}_$cfg.json 2> $OUT/r02g_${lib}_$cfg.err
    python - <<PY
import json
d = json.loads(open("$OUT/r02g_${lib}_$cfg.json").read().strip().splitlines()[-1])
print("$lib $cfg lanes4: %.3f ms/frame %.0f Mrays/s (serial traverse %.3f)" % (d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["traverse"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r02g_kt -- python3 $REPO/tools/trav_ab.py --configs whole,live16,live24 --reps 1 --rounds 1 > $OUT/r02g_kt.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/r02g_kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rt_traverse" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-80:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-36s grid %8s  dur %8.1f us  gap %6.1f" % (r["Kernel_Name"].split("(")[0][-36:], r["Grid_Size_X"], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
PY
