#!/bin/bash
# round 2, GPU session ae: does the hand-over schedule pay on a tile's smaller rounds? (min_rays threshold)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for t in "1/8 --lanes 8 --steps 64 --warmup 16" "1/4 --lanes 8 --steps 48 --warmup 8" "1/2 --lanes 8 --steps 32 --warmup 8"; do
for cfg in "" "--traverse whole" "--trav-adaptive 12,8,65536,3,65536" "--trav-adaptive 12,8,16384,3,32768" "--trav-adaptive 12,8,16384,2,65536"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --force-dist --emulate-tile $t $cfg > $OUT/r02ae.json 2> $OUT/r02ae.err || { echo "FAILED $t $cfg"; continue; }
  python3 -c "import json;d=json.loads(open('$OUT/r02ae.json').read().strip().splitlines()[-1]);print('%-45s %-50s %.3f ms/frame'%('$t','$cfg',d['ms_per_step']))"
done
done
