#!/bin/bash
# Collects HBM-traffic and memory-stall counters for the ORB kernels (separate --pmc passes; kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Run on the GPU box through gpurun.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_orb
ARGS="--steps 2 --warmup 1 --pairs ${PAIRS:-256} --no-cpu-baseline --no-two-handles"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT.fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT.write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_TAG_STALL_sum TCC_BUSY_avr SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/stall -- python3 $R/bench.py $ARGS > $OUT.stall.log 2>&1 || exit 1
echo done
