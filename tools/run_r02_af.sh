#!/bin/bash
# round 2, GPU session af: rounds below how many rays should keep the single-launch schedule? (full frames, two repeats)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
for rep in 1 2; do
for cfg in "--trav-adaptive 12,8,65536,3,262144" "--trav-adaptive 12,8,65536,3,524288" "--trav-adaptive 12,8,65536,3,1048576" "--trav-adaptive 12,8,65536,3,131072"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 24 --warmup 4 $cfg > $OUT/r02af.json 2> $OUT/r02af.err || { echo "FAILED $cfg"; continue; }
  python3 -c "import json;d=json.loads(open('$OUT/r02af.json').read().strip().splitlines()[-1]);print('%-50s %.3f ms/frame'%('$cfg',d['ms_per_step']))"
done
done
timeout -k 10 300 python bench.py --no-cpu-baseline --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 --trav-adaptive 12,8,65536,3,524288 > $OUT/r02af.json 2>/dev/null; python3 -c "import json;d=json.loads(open('$OUT/r02af.json').read().strip().splitlines()[-1]);print('C5 min_rays 2^19 %.3f ms/frame'%(d['ms_per_step']))"
