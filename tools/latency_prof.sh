#!/bin/bash
# Kernel-time share of the single-pair latency: tools/latency.py under rocprofv3 --kernel-trace --stats (GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/lat_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/latency.py --no-cache > $OUT/lat.log 2>&1 || exit 1
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = {"orb": 0.0, "loftr": 0.0}
ORB = ("k_resize", "k_fast", "k_thr_harris", "k_select", "k_describe", "k_match")
for r in rows:
    n = r["Name"]
    k = "orb" if any(o in n for o in ORB) else ("loftr" if "msf" in n else None)
    if k: tot[k] += float(r["TotalDurationNs"])
for k, v in tot.items():
    print(k, "kernel time per call %.3f ms (220 calls)" % (v / 220 / 1e6))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} n={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
grep latency $OUT/lat.log | cut -c1-120
find $OUT/stats -type f ! -name '*kernel_stats.csv' -delete
