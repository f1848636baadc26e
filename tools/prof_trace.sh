#!/bin/bash
# Per-dispatch kernel durations of the ORB bench leg, grouped by (kernel, grid size): tools/prof_trace.sh tag [env...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-t}; shift
OUT=$R/gpurun_out/proft_$TAG
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/bench.py ${PT_ARGS:---steps 4 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary} > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
F=$(find $OUT/tr -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY' > $OUT/summary.txt
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msf::", "")
    key = (name[:40], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for (name, grid), v in acc.items():
    if sum(v) / tot > 0.002:
        print("%-40s grid %9d  n=%3d  avg %8.1f us  min %8.1f  share %5.1f%%" % (name, grid, len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
PY
tail -1 $OUT/bench.log | cut -c1-160 >> $OUT/summary.txt
rm -rf $OUT/tr
cat $OUT/summary.txt
