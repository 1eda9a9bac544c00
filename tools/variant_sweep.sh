#!/bin/bash
# A/B of experiment builds of the library (prismarine-core_amd/csrc/variants/libpsm_<name>.so, same ABI) in one GPU session:
#   tools/variant_sweep.sh <tag> "<variant names, '-' = the product build>" "<bench args 1>;<bench args 2>;..."
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; : > gpurun_out/$tag.txt
IFS=';' read -ra ARGS <<< "$3"
for v in $2; do
  if [ "$v" != "-" ]; then export PSM_HIP_LIB=$PWD/prismarine-core_amd/csrc/variants/libpsm_$v.so; else unset PSM_HIP_LIB; fi
  for a in "${ARGS[@]}"; do
    line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip $a 2>> gpurun_out/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  %.1f Mrays/s' % (d['ms_per_step'], d['value']))")
    echo "variant $v $a: $line" | tee -a gpurun_out/$tag.txt
  done
done
