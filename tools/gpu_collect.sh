#!/bin/bash
# Run HERE (build container): clears the LOCAL pull directory of the tag, collects the rocprofv3 evidence on a GPU box
# (tools/collect_profiles.sh) and digests it.  gpurun merges what the box wrote into the local gpurun_out/ and never
# deletes: without the rm a second collection leaves two <pid>_counter_collection.csv per pass directory, and a sum over
# them is a multiple of one pass (r04's traffic records were 3x).  tools/make_traffic_json.py refuses such a directory.
set -e
TAG=${1:-r05}
EXTRA=${2:-f32}
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$R/gpurun_out/prof_$TAG"
/usr/local/graft/bin/gpurun --timeout 1100 -- "bash tools/collect_profiles.sh $TAG $EXTRA"
python3 "$R/tools/make_traffic_json.py" "$TAG" 6 | tee "$R/profiles/${TAG}_pmc_traffic_summary.txt"
for w in orb_720p:orb orb_vga:orb_vga loftr_vga:loftr; do
  cp "$R/gpurun_out/prof_$TAG/${w#*:}_kernel_stats.csv" "$R/profiles/${TAG}_${w%%:*}_kernel_stats.csv"
done
