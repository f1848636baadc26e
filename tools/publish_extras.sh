#!/bin/bash
# Run HERE after `gpurun -- bash tools/collect_extras.sh <tag>`: copies what the box wrote under gpurun_out/extras_<tag>/ into
# profiles/ (the bench lines without the log lines gloo prints into stdout; the SQ counter tables under one header).
set -e
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/extras_$TAG
grep '^{"metric"' $E/bench_default.json | tail -1 > $R/profiles/${TAG}_bench_default.json
grep '^{"metric"' $E/selflaunch_2rank_gloo.json | tail -1 > $R/profiles/${TAG}_selflaunch_2rank_gloo_pipelined.json
grep '^{"metric"' $E/selflaunch_2rank_stub_product.json | tail -1 > $R/profiles/${TAG}_selflaunch_2rank_stub_product.json
cp $E/kfdb.json $R/profiles/${TAG}_kfdb.json
cp $E/latency.txt $R/profiles/${TAG}_latency.txt
cp $E/parity_sweep_loftr.log $R/profiles/${TAG}_parity_sweep_loftr.log
cp $E/replay_rot_zoom.json $R/profiles/${TAG}_replay_rot_zoom.json
H=$(git -C $R rev-parse --short HEAD)
{ echo "# SQ counters per launch (tools/pmc_walk.sh, two --pmc passes each; $TAG HEAD $H)"
  echo "## ORB, bench.py --steps 2 --warmup 1 (1024 pairs of 1280x720)"; grep "msf::\|^sf::" $E/sq_orb.txt
  echo "## LoFTR, bench.py --matcher loftr --steps 2 --warmup 1 (256 pairs)"; grep "msf::\|^sf::" $E/sq_loftr.txt; } > $R/profiles/${TAG}_sq_counters.txt
echo published
