#!/bin/bash
# A/B the traversal kernel: run tools/trav_bench.py (full sets + size sweep + tiny sets) for every
# build_variants/libpsm_v*.so.  usage (through gpurun): tools/ab_trav.sh > gpurun_out/ab.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for lib in $REPO/build_variants/libpsm_v*.so; do
  TAG=$(basename $lib .so) PSM_HIP_LIB=$lib REPS=5 SIZE_EXP=1 TINY_EXP=1 python $REPO/tools/trav_bench.py || exit 1
done
