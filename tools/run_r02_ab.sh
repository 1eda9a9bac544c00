#!/bin/bash
# round 2, GPU session ab: aggressive hand-over needs more frames in flight? (C3)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for cfg in "--lanes 8" "--lanes 12 --steps 36 --warmup 12" "--traverse adaptive --trav-adaptive 24,8,16384,4,131072 --lanes 8" "--traverse adaptive --trav-adaptive 24,8,16384,4,131072 --lanes 12 --steps 36 --warmup 12" "--traverse adaptive --trav-adaptive 32,4,8192,6,131072 --lanes 8" "--traverse adaptive --trav-adaptive 32,4,8192,6,131072 --lanes 12 --steps 36 --warmup 12" "--traverse adaptive --trav-adaptive 32,4,8192,6,131072 --lanes 16 --steps 48 --warmup 16" "--traverse adaptive --trav-adaptive 40,4,8192,8,65536 --lanes 16 --steps 48 --warmup 16" "--traverse adaptive --trav-adaptive 16,8,65536,3,131072 --lanes 8" "--traverse whole --lanes 8"; do
  timeout -k 10 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline $cfg > $OUT/r02ab.json 2> $OUT/r02ab.err || { echo "FAILED $cfg"; continue; }
  python3 -c "import json;d=json.loads(open('$OUT/r02ab.json').read().strip().splitlines()[-1]);print('%-100s %.3f ms/frame %.0f Mrays/s'%('$cfg',d['ms_per_step'],d['value']))"
done
