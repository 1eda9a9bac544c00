#!/bin/bash
# round 4, session a: the solo gear (psm_rt_set_traverse_solo) -- parity of the traversal tests, then the frame with 0..4 rays
# taken into the gear: 4 frames in flight, one frame at a time, a 1/8 tile
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "travers or chain or multi_bvh or stack or iteration or scaled or cornell or c5 or hits" > gpurun_out/r04_a_tests.log 2>&1 || { tail -40 gpurun_out/r04_a_tests.log; exit 1; }
tail -3 gpurun_out/r04_a_tests.log
A="--steps 96 --warmup 8"
B="--steps 48 --warmup 4 --lanes 1"
T="--force-dist --emulate-tile 1/8 --band-weights none --lanes 8 --steps 192 --warmup 16"
tools/gpu_session.sh sweep r04_a_solo "$A --solo 0;$A --solo 1;$A --solo 2;$A --solo 3;$A --solo 4;$B --solo 0;$B --solo 1;$B --solo 2;$B --solo 3;$B --solo 4;$T --solo 0;$T --solo 1;$T --solo 2;$T --solo 3;$T --solo 4;$A --solo 0;$A --solo 2;$B --solo 0;$B --solo 2"
