#!/bin/bash
# round 4, session d: bvh_emit with the wave-cooperative range search -- build parity at every size, then the build's stage times
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "build or c5 or 2m or load_mesh or obj or graph or rebuild or radiance" > gpurun_out/r04_d_tests.log 2>&1 || { tail -40 gpurun_out/r04_d_tests.log; exit 1; }
tail -3 gpurun_out/r04_d_tests.log
for cfg in "--scene stress --width 3840 --height 2160 --steps 8 --warmup 2" "--steps 16 --warmup 4"; do
  for lanes in 1 0; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip --repeats 3 --lanes $lanes $cfg 2>> gpurun_out/r04_d.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['stage_ms_per_frame']
print('$cfg lanes $lanes: %.3f ms/step; build %.3f (bounds %.3f morton %.3f sort %.3f emit %.3f) traverse %.3f shade %.3f sample %.3f' % (d['ms_per_step'], s['build'], s['bounds'], s['morton'], s['sort'], s['emit_refit'], s['traverse'], s['shade'], s['sample']))" | tee -a gpurun_out/r04_d_stages.txt
  done
done
