#!/bin/bash
# round 4, closing session: the -m gpu suite against the product library, the fenced schedules' tests against the experimental one, smoke,
# the wave log, and the committed bench lines
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout -k 10 1500 python -m pytest tests -q -m gpu -x --durations=6 ) > gpurun_out/r04_z_tests.log 2>&1 || { tail -60 gpurun_out/r04_z_tests.log; exit 1; }
tail -12 gpurun_out/r04_z_tests.log
PSM_HIP_LIB=$PWD/prismarine-core_amd/csrc/variants/libpsm_experimental.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "refill or split or several_pipelines or grouped or schedule" > gpurun_out/r04_z_tests_exp.log 2>&1 || { tail -40 gpurun_out/r04_z_tests_exp.log; exit 1; }
tail -2 gpurun_out/r04_z_tests_exp.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
PSM_HIP_LIB=$PWD/prismarine-core_amd/csrc/variants/libpsm_wavelog.so timeout -k 10 300 python tests/studies/wave_log.py > gpurun_out/r04_wave_log.txt 2>&1 || { tail -20 gpurun_out/r04_wave_log.txt; exit 1; }
grep -v "   wave" gpurun_out/r04_wave_log.txt
# the sharded path end to end on ONE shared GPU (host-staged transport: a rehearsal, the line says so and carries value = null)
for n in 2 4; do
  PSM_DIST_TRANSPORT=hoststaged PSM_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --steps 16 --warmup 4 --no-cpu-baseline --repeats 2 > gpurun_out/r04_rehearsal_gpus${n}_hoststaged.json 2> gpurun_out/r04_rehearsal_gpus${n}.err || { tail -20 gpurun_out/r04_rehearsal_gpus${n}.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_rehearsal_gpus${n}_hoststaged.json') if l.startswith('{')][-1])
print('rehearsal $n ranks on one GPU: rehearsal=%s value=%s rehearsal_value=%.0f Mrays/s %.3f ms/step image mean %.5f' % (d.get('rehearsal'), d['value'], d['rehearsal_value_mrays_s'], d['ms_per_step'], d['image_mean']))"
done
tools/gpu_session.sh bench r04_final_bench_c3 --steps 20 --warmup 5
tools/gpu_session.sh bench r04_final_bench_c3_32 --steps 32 --warmup 8 --no-cpu-baseline
tools/gpu_session.sh bench r04_final_bench_c3_lanes1 --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline
tools/gpu_session.sh bench r04_final_bench_c5 --scene stress --width 3840 --height 2160 --steps 20 --warmup 5
tools/gpu_session.sh bench r04_final_bench_c5_lanes1 --scene stress --width 3840 --height 2160 --steps 8 --warmup 2 --lanes 1 --no-cpu-baseline
