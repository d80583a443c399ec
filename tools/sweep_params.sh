cd $GRAFT_REPO_ROOT
for w in dragon1m sponza stress10m; do
 for env in "X=0" "RT_AMD_REFILL=24" "RT_AMD_REFILL=40" "RT_AMD_REFILL=48" "RT_AMD_PATHS_LEAF_BATCH=4" "RT_AMD_PATHS_LEAF_BATCH=12" "RT_AMD_PATHS_LEAF_BATCH=16" "RT_AMD_PATHS_OCC=6" "RT_AMD_ACCEL_LDS_ENTRIES=12"; do
  echo "== $w $env"
  python3 tools/ab.py --libs cur=raytracing_amd/librt_amd.so --workloads $w --reps 2 --env "$env" | tail -1
 done
done
