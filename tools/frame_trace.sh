#!/bin/bash
# kernel trace of the bench as timed (4 frames in flight) -> residency per 5 ms bin and the timeline of one lane's queue
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rm -rf $REPO/gpurun_out/frame_kt
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/frame_kt -- python3 $REPO/bench.py --no-cpu-baseline --no-obj-roundtrip --lanes ${LANES:-4} --steps 64 --warmup 8 ${EXTRA:-} > $REPO/gpurun_out/frame_kt_bench.json 2> $REPO/gpurun_out/frame_kt.err
f=$(find $REPO/gpurun_out/frame_kt -name '*kernel_trace.csv' | head -1)
python3 $REPO/tools/trace_bins.py $f 5 ${SPAN:-10} > $REPO/gpurun_out/frame_bins.txt
grep -A400 '^queue' $REPO/gpurun_out/frame_bins.txt | grep -v "radix_\|bvh_\|scan_blocks$" | head -${LINES:-120}
find $REPO/gpurun_out/frame_kt -name '*kernel_trace.csv' -delete
