#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   kernel trace + stats of the default bench and of --lanes 1, and separate PMC passes (with --lanes 1: one
#   frame at a time, the launch pattern bench.py's roofline is measured on) for FETCH_SIZE / WRITE_SIZE
#   (gfx950: 4 TCC slots, FETCH_SIZE costs 3, WRITE_SIZE 2 -> separate passes; MI355X_MICROARCH.md).
# usage: profiles/collect.sh <tag>     -> gpurun_out/<tag>_{kt,fetch,write}/ ; summarise with summarize.py
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/${TAG}_kt -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/${TAG}_kt_bench.json 2> $REPO/gpurun_out/${TAG}_kt.err
# the same with one frame at a time: every launch runs alone, so the per-kernel averages are the kernels' own
# durations (with frames in flight, launches of different streams overlap and stretch each other)
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/${TAG}_kt1 -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --lanes 1 > $REPO/gpurun_out/${TAG}_kt1_bench.json 2> $REPO/gpurun_out/${TAG}_kt1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/${TAG}_fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/${TAG}_write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > /dev/null 2>&1
echo collected $TAG
