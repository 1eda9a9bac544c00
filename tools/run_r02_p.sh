#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
echo "== collect C3"; STEPS=8 bash profiles/collect.sh r02_c3 "" 2>&1 | tail -2
echo "== collect C5"; STEPS=4 bash profiles/collect.sh r02_c5 "--scene stress --width 3840 --height 2160" 2>&1 | tail -2
cd $REPO
python3 profiles/make_traffic.py $OUT/traffic_r02.json r02_c3 sponza_like 1920 1080 r02_c5 stress 3840 2160
cp $OUT/traffic_r02.json profiles/traffic_r02.json
# final bench lines with the committed traffic file in place
timeout -k 10 300 python bench.py > $OUT/r02_final_c3.json 2> $OUT/r02_final_c3.err; tail -c 600 $OUT/r02_final_c3.json; echo
timeout -k 10 300 python bench.py --lanes 1 --no-cpu-baseline > $OUT/r02_final_c3_l1.json 2>/dev/null
timeout -k 10 600 python bench.py --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 > $OUT/r02_final_c5.json 2> $OUT/r02_final_c5.err
timeout -k 10 120 tools/ubench/valu_rate > $OUT/r02_valu_rate.txt 2>&1
