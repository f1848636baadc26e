#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the default ORB bench and of the LoFTR bench leg
#   2. HBM traffic counters of the same commands, in separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass)
# Outputs land under gpurun_out/prof_<tag>/ ; tools/pmc_summary.py + tools/make_traffic_json.py digest them.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp
ORB="--steps 5 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary"
VGA="--steps 5 --warmup 1 --no-cpu-baseline --no-two-handles --no-secondary --width 640 --height 480 --pairs 4096"
LOF="--matcher loftr --steps 5 --warmup 1 --no-cpu-baseline --no-two-handles"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/orb_stats -- python3 $R/bench.py $ORB > $OUT/orb_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loftr_stats -- python3 $R/bench.py $LOF > $OUT/loftr_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/orb_fetch -- python3 $R/bench.py $ORB > $OUT/orb_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/orb_write -- python3 $R/bench.py $ORB > $OUT/orb_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/orb_vga_stats -- python3 $R/bench.py $VGA > $OUT/orb_vga_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/orb_vga_fetch -- python3 $R/bench.py $VGA > $OUT/orb_vga_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/orb_vga_write -- python3 $R/bench.py $VGA > $OUT/orb_vga_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/loftr_fetch -- python3 $R/bench.py $LOF > $OUT/loftr_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/loftr_write -- python3 $R/bench.py $LOF > $OUT/loftr_write.log 2>&1 || exit 1
if [ "$2" = "f32" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/loftr_f32_fetch -- python3 $R/bench.py $LOF --loftr-f32 > $OUT/loftr_f32_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/loftr_f32_write -- python3 $R/bench.py $LOF --loftr-f32 > $OUT/loftr_f32_write.log 2>&1 || exit 1
fi
# per-kernel table of the stats passes (name, calls, total / average duration) next to the raw CSVs
for w in orb orb_vga loftr; do
  f=$(find $OUT/${w}_stats -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv
done
echo collected
