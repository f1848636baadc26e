#!/bin/bash
# A/B of ORB bench variants on the GPU box: tools/ab_orb.sh "NAME1:ENV=VAL ENV2=VAL" "NAME2:..." ...
# prints value, ms/step and the stage times of each variant (ORB 720p headline only, no CPU leg)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${AB_ARGS:---steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-two-handles}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  out=$R/gpurun_out/ab_$name.json
  env $envs timeout -k 10 180 python3 $R/bench.py $ARGS > $out 2> $R/gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -3 $R/gpurun_out/ab_$name.err; continue; }
  python3 - "$name" "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
print("%-14s %9.0f pairs/s %7.3f ms  %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"]))
PY
done
