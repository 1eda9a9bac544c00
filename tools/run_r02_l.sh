#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for cfg in "--force-dist --emulate-tile 1/8 --lanes 12" "--force-dist --emulate-tile 1/8 --lanes 16" "--force-dist --emulate-tile 1/8 --lanes 24" "--force-dist --emulate-tile 1/4 --lanes 8" "--force-dist --emulate-tile 1/4 --lanes 12" "--force-dist --emulate-tile 1/2 --lanes 8" "--force-dist --lanes 8" "--emulate-tile 1/8 --lanes 16"; do
  tag=$(echo $cfg | tr ' ,-/' '____')
  timeout -k 10 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline $cfg > $OUT/r02l_$tag.json 2> $OUT/r02l_$tag.err || { echo "FAILED $cfg"; tail -5 $OUT/r02l_$tag.err; continue; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02l_$tag.json").read().strip().splitlines()[-1])
print("%-50s %.3f ms/frame %.0f Mrays/s" % ("$cfg", d["ms_per_step"], d["value"]))
PY
done
