#!/bin/bash
# hardware queues against frames in flight for the 1/8 tile: confirmation runs (192-step and 20-step calls, worker 1/8 and gathering rank 0/8)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
: > gpurun_out/r04_hw_queues2.txt
run() {
  line=$(GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip ${@:2} 2>> gpurun_out/r04_hw_queues2.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f ms/step  %.1f Mrays/s  side by side %s' % (d['ms_per_step'], d['value'], r.get('launches_side_by_side')))")
  echo "GPU_MAX_HW_QUEUES=$1 ${@:2}: $line" | tee -a gpurun_out/r04_hw_queues2.txt
}
for rep in 1 2; do
for tile in 1/8 0/8; do
  T="--force-dist --emulate-tile $tile --band-weights default --repeats 3"
  run 8 $T --lanes 8 --steps 192 --warmup 16
  run 16 $T --lanes 12 --steps 192 --warmup 24
  run 16 $T --lanes 10 --steps 200 --warmup 20
  run 8 $T --lanes 8 --steps 20 --warmup 5
  run 16 $T --lanes 12 --steps 20 --warmup 5
  run 16 $T --lanes 10 --steps 20 --warmup 5
done
done
S="--scene stress --width 3840 --height 2160 --force-dist --emulate-tile 1/8 --band-weights default --repeats 3"
run 8 $S --lanes 8 --steps 48 --warmup 8
run 16 $S --lanes 12 --steps 48 --warmup 12
