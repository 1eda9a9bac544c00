#!/bin/bash
# round 2, GPU session z: traversal parity tests with the library as built, then A/B against variants/libpsm_hip_base.so
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traverse or traversal or primary or schedule or multi_bvh or shade_rounds or overflow or golden" > $OUT/r02z_t.log 2>&1; tail -4 $OUT/r02z_t.log
grep -q " failed\|rror" $OUT/r02z_t.log && exit 1
bash tools/run_r02_y.sh libpsm_hip_base.so "--steps 24 --warmup 4" "--steps 24 --warmup 4 --traverse whole" "--lanes 1" "--scene stress --width 3840 --height 2160 --steps 4 --warmup 1"
