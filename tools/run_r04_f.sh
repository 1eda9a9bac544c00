#!/bin/bash
# round 4, session f: the committed bench lines (C3 as the driver runs it, C3 one frame at a time, C5's scene on one GPU twice over)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python tests/studies/tail_study.py > gpurun_out/r04_tail_study.txt 2>&1; cat gpurun_out/r04_tail_study.txt
tools/gpu_session.sh bench r04_final_bench_c3 --steps 20 --warmup 5
tools/gpu_session.sh bench r04_final_bench_c3_32 --steps 32 --warmup 8 --no-cpu-baseline
tools/gpu_session.sh bench r04_final_bench_c3_lanes1 --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline
tools/gpu_session.sh bench r04_final_bench_c5 --scene stress --width 3840 --height 2160 --steps 20 --warmup 5
tools/gpu_session.sh bench r04_final_bench_c5_lanes1 --scene stress --width 3840 --height 2160 --steps 8 --warmup 2 --lanes 1 --no-cpu-baseline
