#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "sort" > $OUT/r02j_sort.log 2>&1; tail -4 $OUT/r02j_sort.log
grep -q "passed" $OUT/r02j_sort.log || exit 1
grep -q "failed" $OUT/r02j_sort.log && exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r02j_all.log 2>&1; tail -12 $OUT/r02j_all.log
for cfg in "--lanes 4" "--lanes 1" "--lanes 1 --scene stress --width 3840 --height 2160 --steps 4 --warmup 1"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 300 python bench.py --no-cpu-baseline $cfg > $OUT/r02j_$tag.json 2> $OUT/r02j_$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/r02j_$tag.json").read().strip().splitlines()[-1])
print("%-60s %.3f ms/frame %.0f Mrays/s" % ("$cfg", d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in d["stage_ms_per_frame"].items()})
PY
done
