#!/bin/bash
# Collects the rocprofv3 evidence for one configuration on the GPU box (run through gpurun):
#   kernel trace + stats of the bench as timed (frames in flight) and with --lanes 1 (one frame at a time: every
#   launch runs alone, so per-kernel averages are the kernels' own durations), then SEPARATE PMC passes with
#   --lanes 1 (gfx950: 4 TCC slots, FETCH_SIZE costs 3, WRITE_SIZE 2 -> one pass each; MI355X_MICROARCH.md) and
#   one SQ pass for VALU issue / lane utilisation. PMC passes never carry a trace option.
# usage: profiles/collect.sh <tag> ["<extra bench.py args>"]   -> gpurun_out/<tag>_{kt,kt1,fetch,write,sq,sq4,tcc}/
#        summarise with profiles/summarize.py and copy what is to be judged into profiles/.
set -e
TAG=${1:-r02}
EXTRA=${2:-}
STEPS=${STEPS:-8}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the profiler initialises the runtime before bench.py can set it
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -- python3 $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline $EXTRA > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.err
echo "kt done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt1 -- python3 $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --lanes 1 $EXTRA > $OUT/${TAG}_kt1_bench.json 2> $OUT/${TAG}_kt1.err
echo "kt1 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_write.err
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_sq.err
echo "sq done"
# the same SQ counters with the bench as timed (frames in flight => the hand-over traversal schedule of the timed region;
# the profiler serialises the launches, so these are instruction counts and lane utilisation, not overlap)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_sq4 -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $OUT/${TAG}_sq4.err
echo "sq4 done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_tcc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_tcc.err
echo "tcc done"
for d in kt kt1; do
  f=$(find $OUT/${TAG}_$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python3 $REPO/profiles/summarize.py stats $f > $OUT/${TAG}_${d}_stats.txt
done
for d in fetch write sq sq4 tcc; do
  f=$(find $OUT/${TAG}_$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $REPO/profiles/summarize.py pmc $f > $OUT/${TAG}_${d}.txt
done
echo collected $TAG
