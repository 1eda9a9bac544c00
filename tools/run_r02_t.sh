#!/bin/bash
# round 2, GPU session t: captured build graph (parity + A/B), tile emulation, dependent-gather rates, 2 ranks on one GPU over gloo
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "build or graph or frame_batch or sharded or lanes or sort" > $OUT/r02t_t.log 2>&1; tail -4 $OUT/r02t_t.log
grep -q " failed\|rror" $OUT/r02t_t.log && exit 1
for cfg in "" "--no-build-graph" "--lanes 1" "--lanes 1 --no-build-graph" \
           "--force-dist --emulate-tile 1/8 --lanes 8 --steps 48 --warmup 8" "--force-dist --emulate-tile 1/8 --lanes 8 --steps 48 --warmup 8 --no-build-graph" \
           "--force-dist --emulate-tile 1/8 --lanes 16 --steps 64 --warmup 16" \
           "--force-dist --emulate-tile 1/4 --lanes 12 --steps 48 --warmup 12" "--force-dist --emulate-tile 1/2 --lanes 8 --steps 32 --warmup 8" \
           "--force-dist --emulate-tile 0/8 --lanes 8 --steps 48 --warmup 8" \
           "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1" "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1 --no-build-graph"; do
  tag=$(echo "x$cfg" | tr ' ,-/' '____')
  timeout -k 10 400 python bench.py --no-cpu-baseline $cfg > $OUT/r02t_$tag.json 2> $OUT/r02t_$tag.err || { echo "FAILED $cfg"; tail -3 $OUT/r02t_$tag.err; continue; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02t_$tag.json").read().strip().splitlines()[-1])
print("%-95s %.3f ms/frame %.0f Mrays/s  build %.3f" % ("[$cfg]", d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["build"]))
PY
done
echo "== 2 ranks sharing the one GPU, gloo collectives, bench.py starting its own ranks"
PSM_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r02t_gloo2.json 2> $OUT/r02t_gloo2.err; echo "rc $?"; tail -c 700 $OUT/r02t_gloo2.json; echo; tail -3 $OUT/r02t_gloo2.err
echo "== gather rate"
timeout -k 10 300 tools/ubench/gather_rate > $OUT/r02_gather_rate.txt 2>&1; cat $OUT/r02_gather_rate.txt
