#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traverse or traversal or primary or phased or schedule or multi_bvh" > $OUT/r02h_t.log 2>&1; tail -3 $OUT/r02h_t.log
timeout -k 10 200 python tools/trav_ab.py --configs whole,cap96,live16,live16f16k,live16l4 --reps 5 --rounds 2 2>&1 | tail -6
for cfg in "whole" "phased" "adaptive" "adaptive --trav-adaptive 16,8,16384,4,1048576" "whole --lanes 1"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02h_$tag.json 2> $OUT/r02h_$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/r02h_$tag.json").read().strip().splitlines()[-1])
print("%-50s %.3f ms/frame %.0f Mrays/s (serial traverse %.3f)" % ("$cfg", d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["traverse"]))
PY
done
