#!/bin/bash
# round 2, GPU session aa: hand-over parameters again after the leaner step (4 frames in flight, C3)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for cfg in "--traverse whole" "--traverse adaptive --trav-adaptive 12,8,65536,3,262144" "--traverse adaptive --trav-adaptive 16,8,16384,4,262144" "--traverse adaptive --trav-adaptive 24,8,16384,4,262144" "--traverse adaptive --trav-adaptive 24,4,8192,6,131072" "--traverse adaptive --trav-adaptive 32,4,8192,6,131072" "--traverse adaptive --trav-adaptive 32,4,4096,8,65536" "--traverse adaptive --trav-adaptive 40,4,4096,8,65536" "--traverse adaptive --trav-adaptive 16,8,65536,3,131072" "--traverse adaptive --trav-adaptive 20,8,65536,3,262144" "--traverse adaptive --trav-adaptive 12,8,65536,2,262144" "--traverse phased --trav-caps 64,64" "--traverse phased --trav-caps 48,48,48" "--traverse adaptive --trav-adaptive 24,8,16384,4,262144 --lanes 6" "--traverse adaptive --trav-adaptive 24,8,16384,4,262144 --lanes 8"; do
  timeout -k 10 200 python bench.py --steps 24 --warmup 4 --no-cpu-baseline $cfg > $OUT/r02aa.json 2> $OUT/r02aa.err || { echo "FAILED $cfg"; continue; }
  python3 -c "import json;d=json.loads(open('$OUT/r02aa.json').read().strip().splitlines()[-1]);print('%-85s %.3f ms/frame %.0f Mrays/s'%('$cfg',d['ms_per_step'],d['value']))"
done
