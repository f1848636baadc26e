#!/bin/bash
# LoFTR bench at several backbone chunk sizes (pairs per backbone pass): tools/chunk_run.sh 16 32 64 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -f $R/gpurun_out/chunk.txt
for c in "$@"; do
  MSF_LOFTR_CHUNK=$c timeout -k 10 120 python3 $R/bench.py --matcher loftr --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/chunk_$c.log 2>&1 || exit 1
  echo "CHUNK=$c $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/chunk_$c.log) $(grep -o '"backbone_convs": [0-9.]*, "tr' $R/gpurun_out/chunk_$c.log)" >> $R/gpurun_out/chunk.txt
done
cat $R/gpurun_out/chunk.txt
