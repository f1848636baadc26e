export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat -- python3 $R/tools/_lat.py > $R/gpurun_out/prof_lat.log 2>&1
echo done
