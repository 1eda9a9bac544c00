#!/bin/bash
# full GPU suite, then the rocprofv3 evidence for C3 (default bench) and C5 (S-stress 10 M tris, 3840x2160)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r02_gputests2.log 2>&1; tail -12 $OUT/r02_gputests2.log
echo "== collect C3"; STEPS=8 bash profiles/collect.sh r02_c3 "" 2>&1 | tail -3
echo "== collect C5"; STEPS=4 bash profiles/collect.sh r02_c5 "--scene stress --width 3840 --height 2160" 2>&1 | tail -3
cd $REPO
python3 profiles/make_traffic.py $OUT/traffic_r02.json r02_c3 sponza_like 1920 1080 r02_c5 stress 3840 2160
