#!/bin/bash
# round 2, GPU session x: the round's evidence with the final kernels -- rocprofv3 kernel traces + PMC passes of C3 and C5,
# traffic file, final bench lines (with cpu_baseline), tile emulation
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
echo "== collect C3"; STEPS=8 bash profiles/collect.sh r02_c3 "" 2>&1 | tail -2
echo "== collect C5"; STEPS=4 bash profiles/collect.sh r02_c5 "--scene stress --width 3840 --height 2160" 2>&1 | tail -2
cd $REPO
python3 profiles/make_traffic.py $OUT/traffic_r02.json r02_c3 sponza_like 1920 1080 r02_c5 stress 3840 2160
cp $OUT/traffic_r02.json profiles/traffic_r02.json
timeout -k 10 300 python bench.py > $OUT/r02_final_c3.json 2> $OUT/r02_final_c3.err; tail -c 300 $OUT/r02_final_c3.json; echo
timeout -k 10 300 python bench.py --lanes 1 --no-cpu-baseline > $OUT/r02_final_c3_l1.json 2>/dev/null
timeout -k 10 600 python bench.py --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 > $OUT/r02_final_c5.json 2> $OUT/r02_final_c5.err
timeout -k 10 600 python bench.py --scene stress --width 3840 --height 2160 --steps 4 --warmup 1 --lanes 1 --no-cpu-baseline > $OUT/r02_final_c5_l1.json 2>/dev/null
for f in c3 c3_l1 c5 c5_l1; do python3 -c "import json;d=json.loads(open('$OUT/r02_final_$f.json').read().strip().splitlines()[-1]);print('$f %.3f ms/frame %.0f Mrays/s'%(d['ms_per_step'],d['value']))"; done
{
echo "bench.py --force-dist --emulate-tile R/W (one MI355X; the rank's tile through the native psm_dist_* path, frames in flight as given)"
for cfg in "--steps 24" "--force-dist --steps 24" "--force-dist --emulate-tile 1/2 --lanes 8 --steps 32 --warmup 8" "--force-dist --emulate-tile 1/4 --lanes 8 --steps 48 --warmup 8" "--force-dist --emulate-tile 1/4 --lanes 12 --steps 48 --warmup 12" "--force-dist --emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16" "--force-dist --emulate-tile 1/8 --lanes 16 --steps 64 --warmup 16" "--force-dist --emulate-tile 0/8 --lanes 8 --steps 64 --warmup 16" "--emulate-tile 1/8 --lanes 8 --steps 64 --warmup 16"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $cfg > $OUT/r02x_tile.json 2> $OUT/r02x_tile.err || { echo "FAILED $cfg"; continue; }
  python3 -c "import json;d=json.loads(open('$OUT/r02x_tile.json').read().strip().splitlines()[-1]);print('%-80s %.3f ms/frame %.0f Mrays/s'%('[$cfg]',d['ms_per_step'],d['value']))"
done
} > $OUT/r02_tile_emulation.txt 2>&1
cat $OUT/r02_tile_emulation.txt
