#!/bin/bash
# round 2, GPU session ac: several library variants against the shipped one: tools/run_r02_ac.sh "<bench args>" variant.so ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
ARGS=$1; shift
cp prismarine-core_amd/libpsm_hip.so /tmp/psm_orig.so
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline $ARGS > $OUT/r02ac.json 2> $OUT/r02ac.err || { echo "FAILED $1"; tail -3 $OUT/r02ac.err; return; }
  python3 -c "import json;d=json.loads(open('$OUT/r02ac.json').read().strip().splitlines()[-1]);print('%-28s %-40s %.3f ms/frame %.0f Mrays/s  serial traverse %.3f'%('$1','[$ARGS]',d['ms_per_step'],d['value'],d['stage_ms_per_frame']['traverse']))"
}
for rep in 1 2; do
run shipped
for V in "$@"; do
  cp prismarine-core_amd/csrc/variants/$V prismarine-core_amd/libpsm_hip.so
  run $V
  cp /tmp/psm_orig.so prismarine-core_amd/libpsm_hip.so
done
done
