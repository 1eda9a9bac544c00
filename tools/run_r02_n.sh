#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "native or tile" > $OUT/r02n_t.log 2>&1; tail -5 $OUT/r02n_t.log
grep -q "failed\|rror" $OUT/r02n_t.log && exit 1
for cfg in "--force-dist" "--force-dist --lanes 8" "--force-dist --emulate-tile 1/8 --lanes 8" "--force-dist --emulate-tile 1/8 --lanes 12" "--force-dist --emulate-tile 1/8 --lanes 16" "--force-dist --emulate-tile 1/4 --lanes 12" "--force-dist --emulate-tile 1/2 --lanes 8" "--force-dist --emulate-tile 1/2 --lanes 12"; do
  tag=$(echo $cfg | tr ' ,-/' '____')
  timeout -k 10 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline $cfg > $OUT/r02n_$tag.json 2> $OUT/r02n_$tag.err || { echo "FAILED $cfg"; tail -5 $OUT/r02n_$tag.err; continue; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02n_$tag.json").read().strip().splitlines()[-1])
print("%-50s %.3f ms/frame %.0f Mrays/s image_mean %.6f" % ("$cfg", d["ms_per_step"], d["value"], d["image_mean"]))
PY
done
PSM_DIST_PIPELINE=0 timeout -k 10 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --force-dist --emulate-tile 1/8 --lanes 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch-synchronous native, 1/8 tile, 16 lanes:', round(d['ms_per_step'],3))"
