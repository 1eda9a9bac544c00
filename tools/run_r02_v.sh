#!/bin/bash
# round 2, GPU session v: share of the wall time the lane scheduler's host thread spends issuing work (PSM_LANES_PROFILE)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
export PSM_LANES_PROFILE=1
for cfg in "--emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16" "--emulate-tile 1/8 --lanes 16 --steps 64 --warmup 16" "--emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16 --no-build-graph" "--emulate-tile 1/4 --lanes 12 --steps 48 --warmup 12" "--steps 24 --warmup 4" "--lanes 8 --steps 24 --warmup 8"; do
  tag=$(echo "x$cfg" | tr ' ,-/' '____')
  timeout -k 10 300 python bench.py --no-cpu-baseline $cfg > $OUT/r02v_$tag.json 2> $OUT/r02v_$tag.err
  python3 -c "import json;d=json.loads(open('$OUT/r02v_$tag.json').read().strip().splitlines()[-1]);print('[$cfg] %.3f ms/frame'%d['ms_per_step'])"
  grep "psm_lanes_render" $OUT/r02v_$tag.err | tail -2
done
