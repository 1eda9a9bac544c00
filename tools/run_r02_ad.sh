#!/bin/bash
# round 2, GPU session ad: the whole GPU suite, then the round's evidence with the final kernels (as tools/run_r02_x.sh)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r02ad_all.log 2>&1; tail -10 $OUT/r02ad_all.log
grep -q " failed\|rror" $OUT/r02ad_all.log && exit 1
bash tools/run_r02_x.sh
timeout -k 10 120 tools/ubench/valu_rate > $OUT/r02_valu_rate.txt 2>&1
