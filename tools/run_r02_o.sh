#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=4 > $OUT/r02o_all.log 2>&1; tail -10 $OUT/r02o_all.log
grep -q " failed\|rror" $OUT/r02o_all.log && exit 1
for cfg in "--lanes 4" "--lanes 1" "--lanes 8"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline $cfg > $OUT/r02o_$tag.json 2> $OUT/r02o_$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/r02o_$tag.json").read().strip().splitlines()[-1])
print("%-30s %.3f ms/frame %.0f Mrays/s" % ("$cfg", d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in d["stage_ms_per_frame"].items()})
PY
done
