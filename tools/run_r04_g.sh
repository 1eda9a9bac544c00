#!/bin/bash
# round 4, session g: the hand-over parameters and the lane counts again, now that the last launch of a round has the solo gear
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
A="--steps 96 --warmup 8 --repeats 3"
L=""
for p in "12,8,65536,4" "8,8,65536,4" "16,8,65536,4" "12,8,16384,4" "12,8,262144,4" "12,8,65536,3" "12,8,65536,5" "6,8,65536,5" "12,4,65536,4" "12,16,65536,4"; do L="$L$A --trav-adaptive $p,524288;"; done
L="$L$A --lanes 3;$A --lanes 5;$A --lanes 6;$A"
tools/gpu_session.sh sweep r04_g_adaptive "$L"
T="--force-dist --emulate-tile 1/8 --band-weights default --steps 192 --warmup 16 --repeats 3"
tools/gpu_session.sh sweep r04_g_tile_lanes "$T --lanes 4;$T --lanes 6;$T --lanes 8;$T --lanes 10;$T --lanes 12;$T --lanes 8 --trav-adaptive 12,8,65536,4,65536 --traverse adaptive;$T --lanes 8 --trav-adaptive 12,8,8192,3,131072 --traverse adaptive"
