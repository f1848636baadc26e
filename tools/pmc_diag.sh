#!/bin/bash
# Diagnostic SQ counters (instruction fetch, LDS FIFOs, VMEM issue cycles, MFMA / VALU co-execution) of a bench leg.
# usage (GPU box): tools/pmc_diag.sh "<bench args>"; two --pmc passes, kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_diag
rm -rf $OUT; mkdir -p $OUT
ARGS=${1:---matcher loftr --steps 2 --warmup 1 --no-cpu-baseline --no-two-handles}
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/a > $OUT/a.txt
python3 $R/tools/pmc_summary.py $OUT/b > $OUT/b.txt
find $OUT -name '*.csv' -delete
grep -E "strip|down|convx" $OUT/a.txt $OUT/b.txt
