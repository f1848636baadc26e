#!/bin/bash
# HBM traffic per LoFTR kernel (separate --pmc passes, kernel-trace only).  Run on the GPU box through gpurun.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_loftr
rm -rf $OUT; mkdir -p $OUT
ARGS="--matcher loftr --steps 2 --warmup 1 --no-cpu-baseline --no-two-handles"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $OUT/fetch > $OUT/fetch.txt
python3 $R/tools/pmc_summary.py $OUT/write > $OUT/write.txt
find $OUT -name '*.csv' -delete
cat $OUT/fetch.txt $OUT/write.txt
