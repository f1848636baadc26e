#!/bin/bash
# SQ counters of the ORB kernels (two --pmc passes, kernel-trace only): where the walker's wave-cycles go.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_walk
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
ARGS=${PMC_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary}
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/a > $OUT/a.txt
python3 $R/tools/pmc_summary.py $OUT/b > $OUT/b.txt
find $OUT -name '*.csv' -delete
cat $OUT/a.txt $OUT/b.txt
