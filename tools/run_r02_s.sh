#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=3 > $OUT/r02s_all.log 2>&1; tail -8 $OUT/r02s_all.log
grep -q " failed\|rror" $OUT/r02s_all.log && exit 1
for cfg in "" "--lanes 1" "--lanes 8" "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1" "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1 --traverse whole" "--force-dist --emulate-tile 1/8 --lanes 8 --steps 48 --warmup 8" "--force-dist --steps 24"; do
  tag=$(echo "x$cfg" | tr ' ,-/' '____')
  timeout -k 10 400 python bench.py --no-cpu-baseline $cfg > $OUT/r02s_$tag.json 2> $OUT/r02s_$tag.err || { echo "FAILED $cfg"; tail -3 $OUT/r02s_$tag.err; continue; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02s_$tag.json").read().strip().splitlines()[-1])
print("%-75s %.3f ms/frame %.0f Mrays/s" % ("[$cfg]", d["ms_per_step"], d["value"]))
PY
done
