#!/bin/bash
# round 2, GPU session ag: whole GPU suite + smoke + the three bench lines with the final library
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/r02ag_all.log 2>&1; tail -3 $OUT/r02ag_all.log
grep -q " failed\|rror" $OUT/r02ag_all.log && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > $OUT/r02_final_c3.json 2> $OUT/r02_final_c3.err
timeout -k 10 300 python bench.py --lanes 1 --no-cpu-baseline > $OUT/r02_final_c3_l1.json 2>/dev/null
timeout -k 10 600 python bench.py --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 > $OUT/r02_final_c5.json 2> $OUT/r02_final_c5.err
timeout -k 10 600 python bench.py --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 --lanes 1 --no-cpu-baseline > $OUT/r02_final_c5_l1.json 2>/dev/null
for f in c3 c3_l1 c5 c5_l1; do python3 -c "import json;d=json.loads(open('$OUT/r02_final_$f.json').read().strip().splitlines()[-1]);print('$f %.3f ms/frame %.0f Mrays/s traverse serial %.3f frac %.3f'%(d['ms_per_step'],d['value'],d['stage_ms_per_frame']['traverse'],d['roofline']['frac']))"; done
