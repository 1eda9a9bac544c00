#!/bin/bash
# Collects the rocprofv3 evidence for one configuration on the GPU box (run through gpurun):
#   kt    kernel trace + stats of the bench AS TIMED (frames in flight: the hand-over traversal kernel, launches overlap)
#   kt1   the same with --lanes 1 (one frame at a time: AUTO runs the single-launch kernel, every launch alone)
#   kt1a  --lanes 1 --traverse adaptive (the hand-over kernel's launches running alone: what the PMC passes see)
#   fetch1 / write1 / tcc1 / sq1  the same passes with --lanes 1 (single-launch kernel over whole rounds)
#   fetch / write / tcc / sq   SEPARATE PMC passes of the bench as timed (gfx950: 4 TCC slots, FETCH_SIZE costs 3,
#         WRITE_SIZE 2 -> one pass each; MI355X_MICROARCH.md). The profiler serialises the launches, so these are
#         per-launch counts of every kernel the bench runs -- the timed region's hand-over kernel and the serial passes'
#         single-launch kernel alike. PMC passes never carry a trace option.
# usage: profiles/collect.sh <tag> ["<extra bench.py args>"]   -> gpurun_out/<tag>_{kt,kt1,kt1a,fetch,write,sq,tcc}/
#        then profiles/make_traffic.py, and copy what is to be judged into profiles/.
set -e
TAG=${1:-r04}
EXTRA=${2:-}
STEPS=${STEPS:-8}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
# what the counters belong to: the sources of the traversal and shading kernels as they are NOW (make_traffic.py puts these into the
# json; bench.py compares them with the sources it runs from and marks replayed counters stale when a kernel has changed since)
( cd $REPO && sha256sum prismarine-core_amd/csrc/trace.hip prismarine-core_amd/csrc/shade.hip prismarine-core_amd/csrc/psm_common.h prismarine-core_amd/csrc/psm_math.h prismarine-core_amd/csrc/psm_internal.h ) > $OUT/${TAG}_sources.sha256
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the profiler initialises the runtime before bench.py can set it
B="python3 $REPO/bench.py --no-cpu-baseline --no-obj-roundtrip --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -- $B --steps $STEPS --warmup 2 $EXTRA > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.err
echo "kt done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt1 -- $B --steps $STEPS --warmup 2 --lanes 1 $EXTRA > $OUT/${TAG}_kt1_bench.json 2> $OUT/${TAG}_kt1.err
echo "kt1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt1a -- $B --steps $STEPS --warmup 2 --lanes 1 --traverse adaptive $EXTRA > $OUT/${TAG}_kt1a_bench.json 2> $OUT/${TAG}_kt1a.err
echo "kt1a done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- $B --steps 4 --warmup 1 $EXTRA > /dev/null 2> $OUT/${TAG}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- $B --steps 4 --warmup 1 $EXTRA > /dev/null 2> $OUT/${TAG}_write.err
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_sq -- $B --steps 4 --warmup 1 $EXTRA > /dev/null 2> $OUT/${TAG}_sq.err
echo "sq done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_tcc -- $B --steps 4 --warmup 1 $EXTRA > /dev/null 2> $OUT/${TAG}_tcc.err
echo "tcc done"
# the same four passes with --lanes 1: the single-launch kernel over whole rounds only (as timed it also runs the rounds
# below phase_min_rays, which would mix small launches into its per-launch averages)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch1 -- $B --steps 2 --warmup 1 --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_fetch1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write1 -- $B --steps 2 --warmup 1 --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_write1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_sq1 -- $B --steps 2 --warmup 1 --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_sq1.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_tcc1 -- $B --steps 2 --warmup 1 --lanes 1 $EXTRA > /dev/null 2> $OUT/${TAG}_tcc1.err
echo "lanes-1 pmc done"
for d in kt kt1 kt1a; do
  f=$(find $OUT/${TAG}_$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python3 $REPO/profiles/summarize.py stats $f > $OUT/${TAG}_${d}_stats.txt
done
for d in fetch write sq tcc fetch1 write1 sq1 tcc1; do
  f=$(find $OUT/${TAG}_$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $REPO/profiles/summarize.py pmc $f > $OUT/${TAG}_${d}.txt
done
echo collected $TAG
