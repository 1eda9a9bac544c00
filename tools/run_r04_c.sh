#!/bin/bash
# round 4, session c: the -m gpu suite against the product library, the fenced schedules' tests against the experimental one,
# the driver's SCALE command per tile with the lane count chosen from --steps, and the committed bench lines
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout -k 10 1500 python -m pytest tests -q -m gpu -x --durations=8 ) > gpurun_out/r04_c_tests.log 2>&1 || { tail -60 gpurun_out/r04_c_tests.log; exit 1; }
tail -14 gpurun_out/r04_c_tests.log
PSM_HIP_LIB=$PWD/prismarine-core_amd/csrc/variants/libpsm_experimental.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "refill or split or several_pipelines or grouped or schedule" > gpurun_out/r04_c_tests_exp.log 2>&1 || { tail -40 gpurun_out/r04_c_tests_exp.log; exit 1; }
tail -3 gpurun_out/r04_c_tests_exp.log
D="--force-dist --band-weights default --repeats 3"
tools/gpu_session.sh sweep r04_c_tiles "$D --emulate-tile 1/8 --steps 20 --warmup 5;$D --emulate-tile 0/8 --steps 20 --warmup 5;$D --emulate-tile 1/8 --steps 20 --warmup 5 --lanes 8;$D --emulate-tile 1/8 --steps 200 --warmup 20 --lanes 10;$D --emulate-tile 1/8 --steps 192 --warmup 16 --lanes 8;$D --emulate-tile 1/4 --steps 20 --warmup 5;$D --emulate-tile 1/2 --steps 20 --warmup 5;--steps 20 --warmup 5;--steps 20 --warmup 5 --lanes 1"
