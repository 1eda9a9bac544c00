#!/bin/bash
# round 2, GPU session y: A/B of library variants built into prismarine-core_amd/csrc/variants/ (experiment builds of trace.hip)
# usage: tools/run_r02_y.sh <variant.so> ["<bench args>" ...]   -- runs each config with the shipped library, then the variant
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
V=$1; shift
cp prismarine-core_amd/libpsm_hip.so /tmp/psm_orig.so
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline $2 > $OUT/r02y.json 2> $OUT/r02y.err || { echo "FAILED $1 $2"; tail -3 $OUT/r02y.err; return; }
  python3 -c "import json;d=json.loads(open('$OUT/r02y.json').read().strip().splitlines()[-1]);print('%-10s %-50s %.3f ms/frame %.0f Mrays/s  serial traverse %.3f'%('$1','[$2]',d['ms_per_step'],d['value'],d['stage_ms_per_frame']['traverse']))"
}
for cfg in "$@"; do
  run shipped "$cfg"
  cp prismarine-core_amd/csrc/variants/$V prismarine-core_amd/libpsm_hip.so
  run variant "$cfg"
  cp /tmp/psm_orig.so prismarine-core_amd/libpsm_hip.so
done
