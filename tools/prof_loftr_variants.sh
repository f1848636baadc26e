#!/bin/bash
# per-kernel durations (rocprofv3 --stats) of a bench leg (default: LoFTR; PROF_ARGS="..." for another) for a list of A/B
# libraries: name:path ...   (PROF_FILTER: regular expression of the kernel names to print)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}
  out=$R/gpurun_out/pv_$name
  rm -rf $out
  if [ "$lib" != "-" ]; then export MSF_LIB_PATH=$lib; else unset MSF_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py ${PROF_ARGS:---matcher loftr --steps 5 --warmup 1 --no-cpu-baseline --no-two-handles} > $out.log 2>&1 || { echo "$name FAILED"; tail -3 $out.log; continue; }
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  echo "== $name"
  python3 - "$f" "${PROF_FILTER:-strip|down|convx|k_conv}" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("msf::", "")
    if re.search(sys.argv[2], n):
        print("   %-28s %8.1f us" % (n[:28], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out
done
