#!/bin/bash
# the sharded path end to end on ONE shared GPU with bench.py's defaults for a sharded run (12 lanes per rank): a rehearsal, value = null
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 2 4; do
  PSM_DIST_TRANSPORT=hoststaged PSM_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --steps 16 --warmup 4 --no-cpu-baseline --repeats 2 > gpurun_out/r04_rehearsal_gpus${n}_hoststaged.json 2> gpurun_out/r04_rehearsal_gpus${n}.err || { tail -20 gpurun_out/r04_rehearsal_gpus${n}.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04_rehearsal_gpus${n}_hoststaged.json") if l.startswith("{")][-1])
print("rehearsal $n ranks on one GPU: rehearsal=%s value=%s rehearsal_value=%.0f Mrays/s %.3f ms/step, %s lanes per rank, image mean %.5f" % (
    d.get("rehearsal"), d["value"], d["rehearsal_value_mrays_s"], d["ms_per_step"], d["config"]["frames_in_flight"], d["image_mean"]))
PY
done
