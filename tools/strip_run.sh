#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -f $R/gpurun_out/strip.txt
for m in "$@"; do
  export MSF_LOFTR_STRIP=$m
  timeout -k 10 200 python3 $R/tools/dbg_split.py > $R/gpurun_out/dbg_$m.log 2>&1 || { tail -5 $R/gpurun_out/dbg_$m.log; exit 1; }
  timeout -k 10 120 $R/tools/prof_quick.sh loftr strip$m > /dev/null 2>&1 || exit 1
  echo "== STRIP=$m" >> $R/gpurun_out/strip.txt
  head -5 $R/gpurun_out/dbg_$m.log >> $R/gpurun_out/strip.txt
  grep -E "k_stem|k_strip8" $R/gpurun_out/profq_strip$m/summary.txt | cut -c1-30,100-170 >> $R/gpurun_out/strip.txt
  grep -o '"ms_per_step": [0-9.]*\|"backbone_convs": [0-9.]*, "tr' $R/gpurun_out/profq_strip$m/bench.log >> $R/gpurun_out/strip.txt
done
cat $R/gpurun_out/strip.txt
