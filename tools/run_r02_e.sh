#!/bin/bash
# Record of round 2 GPU session e: s_setprio experiments (by launch / by wave age) on an EXPERIMENT build of trace.hip. The
# PSM_EXP_* switches it sets were removed with that build (results in DESIGN.md 5.2; diffs in the history); kept as a record.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline $ARGS > $OUT/r02e_$name.json 2> $OUT/r02e_$name.err || { echo "$name failed"; tail -3 $OUT/r02e_$name.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/r02e_$name.json").read().strip().splitlines()[-1])
print("%-40s %.3f ms/frame %.0f Mrays/s  (serial traverse %.3f ms)" % ("$name", d["ms_per_step"], d["value"], d["stage_ms_per_frame"]["traverse"]))
PY
}
ARGS="--traverse whole" run whole_base X=1
ARGS="--traverse whole" run whole_age64 PSM_EXP_AGE=64
ARGS="--traverse whole" run whole_age128 PSM_EXP_AGE=128
ARGS="--traverse adaptive" run adapt_base X=1
ARGS="--traverse adaptive" run adapt_prio1 PSM_EXP_PRIO=1
ARGS="--traverse adaptive" run adapt_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse adaptive" run adapt_prio3 PSM_EXP_PRIO=3
ARGS="--traverse adaptive" run adapt_age32 PSM_EXP_AGE=32
ARGS="--traverse adaptive --trav-adaptive 24,8,4096,8,1048576" run adapt24_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse adaptive --trav-adaptive 16,8,4096,8,262144" run adapt16_min256k_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse adaptive --lanes 8" run adapt_l8_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse adaptive --lanes 6" run adapt_l6_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse adaptive --lanes 1" run adapt_l1_prio1_age64 PSM_EXP_PRIO=1 PSM_EXP_AGE=64
ARGS="--traverse whole --lanes 1" run whole_l1 X=1
