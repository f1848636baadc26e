#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do
  MSF_LOFTR_CHUNK=$c timeout -k 10 120 tools/prof_quick.sh loftr chunk$c > /dev/null 2>&1 || exit 1
  echo "== CHUNK=$c" >> $R/gpurun_out/chunk.txt
  head -12 $R/gpurun_out/profq_chunk$c/summary.txt | sed 's/(float const.*n=/ n=/; s/(void const.*n=/ n=/' >> $R/gpurun_out/chunk.txt
  grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/profq_chunk$c/bench.log >> $R/gpurun_out/chunk.txt
  grep -o '"backbone_convs": [0-9.]*, "tr' $R/gpurun_out/profq_chunk$c/bench.log >> $R/gpurun_out/chunk.txt
done
cat $R/gpurun_out/chunk.txt
