#!/bin/bash
# SQ counters of the LoFTR head kernels (kernel-trace + pmc only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sim
rm -rf $OUT; mkdir -p $OUT
ARGS="--matcher loftr --steps 2 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/a > $OUT/a.txt
python3 $R/tools/pmc_summary.py $OUT/b > $OUT/b.txt
find $OUT -name '*.csv' -delete
grep -E "k_sim|k_attn|k_stem|k_strip8x" $OUT/a.txt $OUT/b.txt
