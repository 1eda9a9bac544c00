#!/bin/bash
# kernel trace of the emulated 1/8 tile (native sharded path on one GPU, 8 frames in flight) -> residency summary
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rm -rf $REPO/gpurun_out/tile_kt
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tile_kt -- python3 $REPO/bench.py --no-cpu-baseline --no-obj-roundtrip ${DIST---force-dist} --emulate-tile ${TILE:-1/8} --band-weights none --lanes ${LANES:-8} --steps 96 --warmup 8 > $REPO/gpurun_out/tile_kt_bench.json 2> $REPO/gpurun_out/tile_kt.err
f=$(find $REPO/gpurun_out/tile_kt -name '*kernel_trace.csv' | head -1)
python3 $REPO/tools/trace_bins.py $f 5 6 > $REPO/gpurun_out/tile_bins.txt
python3 $REPO/tools/trace_concurrency.py $f 20 > $REPO/gpurun_out/tile_concurrency.txt
q=$(python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
c=collections.Counter(r["Queue_Id"] for r in rows[-2000:] if "rt_shade" in r["Kernel_Name"])
print(c.most_common(1)[0][0])
PY
)
python3 $REPO/tools/trace_concurrency.py $f 6 $q > $REPO/gpurun_out/tile_queue.txt
grep -A200 '^queue' $REPO/gpurun_out/tile_bins.txt | head -150
find $REPO/gpurun_out/tile_kt -name '*kernel_trace.csv' -delete
