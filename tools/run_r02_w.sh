#!/bin/bash
# round 2, GPU session w: one queue's timeline in a 1/8-tile run (where does a lane's round latency go?)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/r02w_tile8 -- python3 $REPO/bench.py --emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16 --no-cpu-baseline > $OUT/r02w_tile8.json 2> $OUT/r02w_tile8.err
f=$(find $OUT/r02w_tile8 -name '*kernel_trace.csv' | head -1)
python3 $REPO/tools/trace_concurrency.py $f 20 3 > $OUT/r02w_tile8_conc.txt; head -150 $OUT/r02w_tile8_conc.txt
m=$(find $OUT/r02w_tile8 -name '*memory_copy_trace.csv' | head -1)
[ -n "$m" ] && head -5 $m && wc -l $m
cp $f $OUT/r02w_kernel_trace.csv; [ -n "$m" ] && cp $m $OUT/r02w_memcpy_trace.csv
rm -rf $OUT/r02w_tile8
