#!/bin/bash
# round 4, session b: the whole -m gpu suite (with the C4 / sharded C5 tests), the solo gear's step cost, the frame with 0..2
# rays taken into the gear, and the driver's SCALE command (--steps 20 --warmup 5) per emulated tile next to 192-step runs
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 200 python tests/studies/solo_debug.py > gpurun_out/r04_solo_debug.txt 2>&1; grep -c " 0 of" gpurun_out/r04_solo_debug.txt
( time timeout -k 10 1500 python -m pytest tests -q -m gpu -x --durations=15 ) > gpurun_out/r04_b_tests.log 2>&1 || { tail -60 gpurun_out/r04_b_tests.log; exit 1; }
tail -25 gpurun_out/r04_b_tests.log
timeout -k 10 300 python tests/studies/solo_step.py > gpurun_out/r04_solo_step.txt 2>&1; grep "C3" gpurun_out/r04_solo_step.txt
A="--steps 96 --warmup 8"
B="--steps 48 --warmup 4 --lanes 1"
T="--force-dist --emulate-tile 1/8 --band-weights none --lanes 8 --steps 192 --warmup 16"
tools/gpu_session.sh sweep r04_b_solo "$A --solo 0;$A --solo 1;$A --solo 2;$B --solo 0;$B --solo 1;$B --solo 2;$T --solo 0;$T --solo 1;$T --solo 2;$A --solo 0;$A --solo 1;$B --solo 0;$B --solo 1"
D="--force-dist --band-weights default --lanes 8 --repeats 3"
S="--scene stress --width 3840 --height 2160"
tools/gpu_session.sh sweep r04_b_tiles "$D --emulate-tile 1/8 --steps 192 --warmup 16;$D --emulate-tile 1/8 --steps 20 --warmup 5;$D --emulate-tile 0/8 --steps 192 --warmup 16;$D --emulate-tile 0/8 --steps 20 --warmup 5;--steps 20 --warmup 5;--steps 96 --warmup 8;$S $D --emulate-tile 1/8 --steps 48 --warmup 8;$S $D --emulate-tile 1/8 --steps 20 --warmup 5;$S $D --emulate-tile 0/8 --steps 48 --warmup 8;$S $D --emulate-tile 0/8 --steps 20 --warmup 5;$S --steps 20 --warmup 5"
