#!/bin/bash
# radix_local's shape (PSM_SORT_TUNE = S_small,S_large,threads[,cap_small,cap_large[,match]]) against the Morton codes of the two
# bench scenes; GPU box:  bash tools/sort_sweep.sh > gpurun_out/r05_sort_sweep.txt
for tune in 1024,3072,1024,4096,5120,0 1024,3072,1024,4096,5120,1 1024,2048,1024,4096,4096,1 1024,2048,512,4096,4096,1 1024,2048,512,4096,4096,0 1024,3072,512,4096,5120,1 1024,4096,512,4096,6144,1 1024,4096,512,4096,6144,0 512,3072,512,4096,5120,1; do
  PSM_SORT_TUNE=$tune timeout -k 10 200 python3 tools/sort_bench.py --no-uniform --algos 0,2 2>&1 | grep -v "three-kernel" || exit 1
done
