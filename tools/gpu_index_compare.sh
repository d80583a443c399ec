# host-built vs GPU-built search index: upload time and tracing rate on the benchmark scenes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for args in "--kind 0 --level 8 --copies 1 --frames 64 --warmup 8" "--kind 1 --level 64 --width 3840 --height 2160 --frames 21 --warmup 4" "--kind 0 --level 8 --copies 8 --bounces 8 --frames 64 --warmup 8"; do
  for g in "" "--gpu-index"; do
    echo "== $args $g"
    RT_AMD_VERBOSE=1 timeout -k 10 200 python3 tools/prof_render.py $args $g --count --checksum 2>&1 | tail -4
  done
done
