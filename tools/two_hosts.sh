#!/bin/bash
# Is the tile-sharded frame loop bound by the ONE host thread that drives its lanes? Two bench processes, each with half the
# lanes, on the same GPU at the same time: if their combined rate beats one process with all the lanes, it is.
cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--no-cpu-baseline --no-obj-roundtrip --force-dist --emulate-tile 1/8 --band-weights none --steps 192 --warmup 8"
one=$(timeout -k 10 300 python bench.py $A --lanes 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
echo "one process, 8 lanes: $one ms/frame"
(MASTER_PORT=29611 timeout -k 10 300 python bench.py $A --lanes 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('process A, 4 lanes:', d['ms_per_step'], 'ms/frame')") &
(MASTER_PORT=29612 timeout -k 10 300 python bench.py $A --lanes 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('process B, 4 lanes:', d['ms_per_step'], 'ms/frame')") &
wait
