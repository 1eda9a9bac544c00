#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "native_rccl or tile" > $OUT/r02k_t.log 2>&1; tail -4 $OUT/r02k_t.log
grep -q "failed\|error" $OUT/r02k_t.log && exit 1
for cfg in "--force-dist" "--force-dist --lanes 1" "--force-dist --emulate-tile 1/8 --lanes 8" "--emulate-tile 1/8 --lanes 8" "--emulate-tile 1/8 --lanes 4" "--emulate-tile 0/8 --lanes 8" "--lanes 4"; do
  tag=$(echo $cfg | tr ' ,-/' '____')
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline $cfg > $OUT/r02k_$tag.json 2> $OUT/r02k_$tag.err || { echo "FAILED $cfg"; tail -5 $OUT/r02k_$tag.err; continue; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02k_$tag.json").read().strip().splitlines()[-1])
print("%-50s %.3f ms/frame %.0f Mrays/s image_mean %.6f" % ("$cfg", d["ms_per_step"], d["value"], d["image_mean"]), {k: round(v, 3) for k, v in d["stage_ms_per_frame"].items() if k in ("build","camera","traverse","shade","sample")})
PY
done
