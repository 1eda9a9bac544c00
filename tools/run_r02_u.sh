#!/bin/bash
# round 2, GPU session u: new edge-case parity tests; where does a 1/8 tile's frame time go (kernel residency from traces)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traverse or shade_rounds or overflow or accumulated" > $OUT/r02u_t.log 2>&1; tail -4 $OUT/r02u_t.log
grep -q " failed\|rror" $OUT/r02u_t.log && exit 1
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
for lanes in 8 16; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/r02u_tile8_l$lanes -- python3 $REPO/bench.py --force-dist --emulate-tile 1/8 --lanes $lanes --steps 64 --warmup 16 --no-cpu-baseline > $OUT/r02u_tile8_l$lanes.json 2> $OUT/r02u_tile8_l$lanes.err
  f=$(find $OUT/r02u_tile8_l$lanes -name '*kernel_trace.csv' | head -1)
  echo "== 1/8 tile, $lanes lanes: $(python3 -c "import json;d=json.loads(open('$OUT/r02u_tile8_l$lanes.json').read().strip().splitlines()[-1]);print('%.3f ms/frame'%d['ms_per_step'])")"
  python3 $REPO/tools/trace_concurrency.py $f 20 > $OUT/r02u_tile8_l${lanes}_conc.txt; cat $OUT/r02u_tile8_l${lanes}_conc.txt
  rm -rf $OUT/r02u_tile8_l$lanes
done
cd $REPO
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --force-dist --emulate-tile 1/8 --lanes 16 --steps 64 --warmup 16 --no-cpu-baseline > $OUT/r02u_q16.json 2>/dev/null; python3 -c "import json;d=json.loads(open('$OUT/r02u_q16.json').read().strip().splitlines()[-1]);print('GPU_MAX_HW_QUEUES=16, 16 lanes: %.3f ms/frame'%d['ms_per_step'])"
GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python bench.py --force-dist --emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16 --no-cpu-baseline > $OUT/r02u_q4.json 2>/dev/null; python3 -c "import json;d=json.loads(open('$OUT/r02u_q4.json').read().strip().splitlines()[-1]);print('GPU_MAX_HW_QUEUES=4, 8 lanes: %.3f ms/frame'%d['ms_per_step'])"
