#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for cfg in "--traverse whole" "--traverse adaptive" "--traverse whole --lanes 1" "--traverse adaptive --lanes 1" "--traverse whole --lanes 2" "--traverse adaptive --lanes 8"; do
  tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --diag-clock $cfg > $OUT/r02f_$tag.json 2> $OUT/r02f_$tag.err
  echo "== $cfg"; grep diag-clock $OUT/r02f_$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/r02f_$tag.json").read().strip().splitlines()[-1])
print("   %.3f ms/frame %.0f Mrays/s" % (d["ms_per_step"], d["value"]))
PY
done
