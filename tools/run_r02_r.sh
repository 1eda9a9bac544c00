#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
run() {
  cfg="$1"; tag=$(echo $cfg | tr ' ,-' '___')
  timeout -k 10 200 python bench.py --steps 24 --warmup 4 --no-cpu-baseline $cfg > $OUT/r02r_$tag.json 2> $OUT/r02r_$tag.err || { echo "FAILED $cfg"; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02r_$tag.json").read().strip().splitlines()[-1])
print("%-62s %.3f ms/frame %.0f Mrays/s" % ("$cfg", d["ms_per_step"], d["value"]))
PY
}
run "--traverse whole"
for a in "8,8,16384,3,1048576" "12,8,16384,3,1048576" "16,8,16384,3,1048576" "12,8,16384,2,1048576" "8,8,16384,2,1048576" "16,8,16384,2,1048576" "12,8,65536,3,1048576" "12,8,4096,3,1048576" "12,16,16384,3,1048576" "12,8,16384,3,262144" "12,8,16384,3,0" "10,8,16384,3,1048576" "12,8,16384,4,262144" "20,8,32768,3,1048576"; do
  run "--traverse adaptive --trav-adaptive $a"
done
run "--traverse phased --trav-caps 64"
run "--traverse phased --trav-caps 80"
run "--traverse phased --trav-caps 128"
run "--traverse phased --trav-caps 64,64"
run "--traverse adaptive --trav-adaptive 12,8,16384,3,1048576 --lanes 3"
run "--traverse adaptive --trav-adaptive 12,8,16384,3,1048576 --lanes 6"
run "--traverse adaptive --trav-adaptive 12,8,16384,3,1048576 --lanes 2"
run "--traverse whole"
