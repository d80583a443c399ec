#!/bin/bash
# Sweep of the path kernel's run-time knobs on the benchmark workloads (medians of 2 runs each):
#   tools/sweep_params.sh > gpurun_out/sweep.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in dragon1m sponza stress10m demo; do
 for env in "X=0" "RT_AMD_REFILL=20" "RT_AMD_REFILL=24" "RT_AMD_REFILL=28" "RT_AMD_REFILL=40" "RT_AMD_PATHS_LEAF_BATCH=4" "RT_AMD_PATHS_LEAF_BATCH=6" "RT_AMD_PATHS_LEAF_BATCH=12" "RT_AMD_PATHS_LEAF_BATCH=16"; do
  echo "== $w $env"
  timeout -k 10 200 python3 tools/ab.py --libs cur=raytracing_amd/librt_amd.so --workloads $w --reps 2 --env "$env" | tail -1 || exit 1
 done
done
