#!/bin/bash
# round 4, session i: history-ordered dispatch (psm_rt_set_traverse_reorder) -- parity, then on / off
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_i_tests.log 2>&1 || { tail -40 gpurun_out/r04_i_tests.log; exit 1; }
tail -2 gpurun_out/r04_i_tests.log
A="--steps 96 --warmup 8 --repeats 3"
B="--steps 48 --warmup 4 --lanes 1 --repeats 3"
B2="--steps 48 --warmup 4 --lanes 2 --repeats 3"
S="--scene stress --width 3840 --height 2160 --steps 16 --warmup 4 --repeats 3"
tools/gpu_session.sh sweep r04_i_reorder "$B;$B --no-reorder;$A;$A --no-reorder;$B2;$B2 --no-reorder;$S;$S --no-reorder;$S --lanes 1;$S --lanes 1 --no-reorder;$B;$B --no-reorder;$A;$A --no-reorder;--steps 20 --warmup 5;--steps 20 --warmup 5 --no-reorder"
timeout -k 10 300 python tests/studies/tail_study.py > gpurun_out/r04_tail_study_i.txt 2>&1; cat gpurun_out/r04_tail_study_i.txt
