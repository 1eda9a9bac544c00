#!/bin/bash
# One parametrised GPU session script (run through gpurun from the repo root); replaces the per-session scripts of
# round 2. Every step writes under gpurun_out/; steps are joined with && so that nothing runs after a failure.
#   tools/gpu_session.sh tests [pytest args]     the -m gpu suite
#   tools/gpu_session.sh bench <tag> [bench args] one bench line -> gpurun_out/<tag>.json / .err
#   tools/gpu_session.sh tiles <tag>              tile emulation sweep (native sharded path, one GPU)
#   tools/gpu_session.sh collect <tag> [args]     profiles/collect.sh
#   tools/gpu_session.sh rehearsal <tag>          bench.py --gpus 2 and 4 with the ranks SHARING this GPU (host-staged transport):
#                                                 the sharded path end to end incl. its self-check -- a rehearsal, value = null
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
what=$1; shift
case $what in
  tests)
    timeout -k 10 1100 python -m pytest tests -q -m gpu "$@" > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
    tail -3 gpurun_out/gputests.log ;;
  bench)
    tag=$1; shift
    timeout -k 10 900 python bench.py "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -20 gpurun_out/$tag.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/$tag.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$tag: %.1f Mrays/s, %.3f ms/step; %s x%d avg %.3f ms, %.3f ms/step, frac %.3f, hbm_frac %s" % (
    d["value"], d["ms_per_step"], r["kernel"], r["launches"], r["avg_launch_ms"], r["ms_per_step"], r["frac"], r.get("hbm_frac")))
PY
    ;;
  tiles)
    tag=$1; shift
    : > gpurun_out/$tag.txt
    CFGS=${TILE_CFGS:-"0/1 4 none;1/2 8 none;0/2 8 default;1/2 8 default;1/4 8 none;0/4 8 default;1/4 8 default;1/8 8 none;0/8 8 none;0/8 8 default;1/8 8 default"}
    IFS=';' read -ra LIST <<< "$CFGS"
    for cfg in "${LIST[@]}"; do
      set -- $cfg
      line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip --force-dist --emulate-tile $1 --lanes $2 --band-weights $3 --steps ${TILE_STEPS:-32} --warmup ${TILE_WARMUP:-4} $TILE_EXTRA 2>> gpurun_out/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/frame  %.1f Mrays/s' % (d['ms_per_step'], d['value']))")
      echo "tile $1 lanes $2 weights $3 steps ${TILE_STEPS:-32} $TILE_EXTRA: $line" | tee -a gpurun_out/$tag.txt
    done ;;
  sweep)   # tools/gpu_session.sh sweep <tag> "<args 1>;<args 2>;..."   one short bench line per argument set
    tag=$1; shift
    : > gpurun_out/$tag.txt
    IFS=';' read -ra LIST <<< "$1"
    for a in "${LIST[@]}"; do
      line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-obj-roundtrip $a 2>> gpurun_out/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  %.1f Mrays/s' % (d['ms_per_step'], d['value']))")
      echo "$a: $line" | tee -a gpurun_out/$tag.txt
    done ;;
  collect)
    bash profiles/collect.sh "$@" ;;
  rehearsal)
    tag=$1; shift
    for n in 2 4; do
      PSM_DIST_TRANSPORT=hoststaged PSM_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --steps 16 --warmup 4 --no-cpu-baseline --repeats 2 "$@" > gpurun_out/${tag}_gpus${n}_hoststaged.json 2> gpurun_out/${tag}_gpus${n}.err || { tail -20 gpurun_out/${tag}_gpus${n}.err; exit 1; }
      python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${tag}_gpus${n}_hoststaged.json") if l.startswith("{")][-1])
print("rehearsal $n ranks on one GPU: rehearsal=%s value=%s (%.0f Mrays/s, %.3f ms/step) sharded_check=%s comm_ranks=%s rays per rank %s" % (
    d.get("rehearsal"), d["value"], d["rehearsal_value_mrays_s"], d["ms_per_step"], d["sharded_check"], d["ranks"]["comm_ranks"], d["ranks"]["rays_traced_per_rank"]))
PY
    done ;;
  *) echo "unknown step $what"; exit 2 ;;
esac
