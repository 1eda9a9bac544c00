#!/bin/bash
# round 4, session e: the trimmed solo step -- parity, its cost, the frame
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 200 python tests/studies/solo_debug.py > gpurun_out/r04_solo_debug.txt 2>&1; grep -c " 0 of" gpurun_out/r04_solo_debug.txt; grep -v " 0 of" gpurun_out/r04_solo_debug.txt | head -5
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "travers or chain or multi_bvh or stack or iteration or scaled or cornell or c5 or hits or obj or radiance" > gpurun_out/r04_e_tests.log 2>&1 || { tail -40 gpurun_out/r04_e_tests.log; exit 1; }
tail -3 gpurun_out/r04_e_tests.log
timeout -k 10 300 python tests/studies/solo_step.py > gpurun_out/r04_solo_step_e.txt 2>&1; grep "C3" gpurun_out/r04_solo_step_e.txt
A="--steps 96 --warmup 8 --repeats 3"
B="--steps 48 --warmup 4 --lanes 1 --repeats 3"
T="--force-dist --emulate-tile 1/8 --band-weights none --lanes 8 --steps 192 --warmup 16 --repeats 3"
tools/gpu_session.sh sweep r04_e_solo "$A --solo 0;$A --solo 1;$A --solo 2;$A --solo 3;$B --solo 0;$B --solo 1;$B --solo 2;$B --solo 3;$T --solo 0;$T --solo 1;$T --solo 2;$T --solo 3"
