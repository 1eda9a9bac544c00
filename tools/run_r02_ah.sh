#!/bin/bash
# round 2, GPU session ah: node records on demand: build parity + graph tests, then the build stage's time (C3, C5)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "build or graph or c5 or 2m or lifecycle or load_mesh or golden or multi_bvh" > $OUT/r02ah_t.log 2>&1; tail -3 $OUT/r02ah_t.log
grep -q " failed\|rror" $OUT/r02ah_t.log && exit 1
for cfg in "--lanes 1" "--steps 24 --warmup 4" "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline $cfg > $OUT/r02ah.json 2>/dev/null
  python3 -c "import json;d=json.loads(open('$OUT/r02ah.json').read().strip().splitlines()[-1]);s=d['stage_ms_per_frame'];print('%-70s %.3f ms/frame  build %.3f emit %.3f'%('[$cfg]',d['ms_per_step'],s['build'],s['emit_refit']))"
done
