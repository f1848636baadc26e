#!/bin/bash
# Quick per-kernel timing of one bench leg on the GPU box: tools/prof_quick.sh loftr|orb [tag]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LEG=${1:-loftr}
TAG=${2:-q}
OUT=$R/gpurun_out/profq_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp
if [ "$LEG" = loftr ]; then ARGS="--matcher loftr --steps 5 --warmup 1 --no-cpu-baseline --no-two-handles"; else ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench.log 2>&1 || exit 1
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY' > $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
PY
tail -1 $OUT/bench.log | cut -c1-200 >> $OUT/summary.txt
find $OUT/stats -type f ! -name '*kernel_stats.csv' -delete
cat $OUT/summary.txt
