#!/bin/bash
# The rest of a round's evidence (GPU box, after tools/collect_profiles.sh): SQ counters of both matchers, the default
# bench line, the two-rank rehearsals on one card (gloo + torch gather; gloo + the product gather over the test-only RCCL
# stand-in), the LoFTR parity sweep, single-pair latency, KeyFrameMatchDatabase query, 1000-frame replay.
# Everything lands under gpurun_out/extras_<tag>/ ; copy what is to be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
O=$R/gpurun_out/extras_$TAG
rm -rf $O; mkdir -p $O
cd $R
bash tools/pmc_walk.sh > $O/sq_orb.txt 2>&1 || exit 1
bash tools/pmc_walk.sh "PMC_ARGS=--matcher loftr --steps 2 --warmup 1 --no-cpu-baseline --no-two-handles" > $O/sq_loftr.txt 2>&1 || exit 1
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-two-handles > $O/selflaunch_2rank_gloo.json 2> $O/selflaunch_2rank_gloo.err || exit 1
MSF_RCCL_LIBRARY=$R/tests/stub_rccl/libstub_rccl.so python3 bench.py --gpus 2 --backend gloo --gather product --steps 10 --warmup 3 --no-cpu-baseline --no-two-handles > $O/selflaunch_2rank_stub_product.json 2> $O/selflaunch_2rank_stub_product.err || exit 1
python3 tools/parity_sweep_loftr.py > $O/parity_sweep_loftr.log 2>&1 || exit 1
python3 tools/latency.py > $O/latency.txt 2>&1 && python3 tools/latency.py --no-cache >> $O/latency.txt 2>&1 || exit 1
python3 tools/bench_kfdb.py > $O/kfdb.json 2> $O/kfdb.err || exit 1
python3 tools/replay_sequence.py --frames 1000 --rotate-deg 30 --zoom 1.3 > $O/replay_rot_zoom.json 2> $O/replay.err || exit 1
echo extras collected
