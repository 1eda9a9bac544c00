#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traverse or traversal or primary or schedule or multi_bvh or shade_rounds" > $OUT/r02q_t.log 2>&1; tail -3 $OUT/r02q_t.log
grep -q " failed\|rror" $OUT/r02q_t.log && exit 1
timeout -k 10 200 python tools/trav_ab.py --configs whole,cap96,live16,live16f16k,live16l4,live24 --reps 5 --rounds 2 2>&1 | tail -7
for cfg in "whole" "phased" "adaptive" "adaptive --trav-adaptive 16,8,16384,4,1048576" "adaptive --trav-adaptive 24,8,16384,4,1048576" "adaptive --trav-adaptive 12,8,16384,3,1048576" "adaptive --lanes 8" "whole --lanes 1"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 200 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02q_$tag.json 2> $OUT/r02q_$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/r02q_$tag.json").read().strip().splitlines()[-1])
print("%-55s %.3f ms/frame %.0f Mrays/s (serial traverse %.3f)" % ("$cfg", d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["traverse"]))
PY
done
