#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
for cfg in whole adaptive; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/r02c_kt_$cfg -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --traverse $cfg > $OUT/r02c_kt_$cfg.json 2> $OUT/r02c_kt_$cfg.err
f=$(find $OUT/r02c_kt_$cfg -name '*kernel_trace.csv' | head -1)
python3 $REPO/tools/trace_concurrency.py $f 40 3 > $OUT/r02c_conc_$cfg.txt
head -20 $OUT/r02c_conc_$cfg.txt
done
