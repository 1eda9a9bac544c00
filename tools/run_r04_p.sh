#!/bin/bash
# psm_dist_render_frames serving its lanes during exchanges, gathers and batch starts: the sharded tests, then tile emulation
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -q -m gpu -x -k "dist or sharded or native or world or c4 or c5 or weighted" > gpurun_out/r04_p_tests.log 2>&1 || { tail -40 gpurun_out/r04_p_tests.log; exit 1; }
tail -2 gpurun_out/r04_p_tests.log
: > gpurun_out/r04_tile_polled.txt
run() {
  line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip $@ 2>> gpurun_out/r04_tile_polled.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f ms/step  %.1f Mrays/s  side by side %s' % (d['ms_per_step'], d['value'], r.get('launches_side_by_side')))")
  echo "$@: $line" | tee -a gpurun_out/r04_tile_polled.txt
}
for rep in 1 2; do
for tile in 1/8 0/8; do
  T="--force-dist --emulate-tile $tile --band-weights default --repeats 3"
  run $T --steps 192 --warmup 24
  run $T --steps 20 --warmup 5
done
done
run --scene stress --width 3840 --height 2160 --force-dist --emulate-tile 1/8 --band-weights default --repeats 3 --steps 48 --warmup 12
run --scene stress --width 3840 --height 2160 --force-dist --emulate-tile 0/8 --band-weights default --repeats 3 --steps 48 --warmup 12
run --force-dist --emulate-tile 1/4 --band-weights default --repeats 3 --steps 96 --warmup 12
run --force-dist --emulate-tile 1/2 --band-weights default --repeats 3 --steps 96 --warmup 12
run --force-dist --band-weights none --repeats 3 --steps 48 --warmup 8
