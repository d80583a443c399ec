cd ${GRAFT_REPO_ROOT:-/root/repo}
for env in "RT_AMD_POOL=1,RT_AMD_POOL_OCC=4,RT_AMD_POOL_LDS_ENTRIES=16" "RT_AMD_POOL=1,RT_AMD_POOL_OCC=5,RT_AMD_POOL_LDS_ENTRIES=12" "RT_AMD_POOL=1,RT_AMD_POOL_OCC=4,RT_AMD_POOL_LDS_ENTRIES=8" "RT_AMD_POOL=1,RT_AMD_POOL_OCC=6,RT_AMD_POOL_LDS_ENTRIES=6"; do
  echo "== $env"
  python3 tools/ab.py --libs pool=raytracing_amd/librt_amd.so --workloads dragon1m --reps 1 --env "$env" | tail -1
done
