#!/bin/bash
# the solo gear: --solo 0 / 1 / 2 / 4 (rays a wave takes into the gear, one after the other) on a frame alone, 4 frames in flight, the emulated 1/8 tile
cd ${GRAFT_REPO_ROOT:-/root/repo}
for solo in 1 0 2 4; do
  for cfg in "--lanes 1 --steps 16 --warmup 4" "--steps 32 --warmup 8" "--force-dist --emulate-tile 1/8 --lanes 12 --band-weights default --steps 96 --warmup 12"; do
    line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip --repeats 3 --solo $solo $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step (min %.3f)  %.1f Mrays/s' % (d['ms_per_step'], d['timing']['ms_per_step_min'], d['value'] or 0))")
    echo "solo $solo  $cfg: $line"
  done
done
