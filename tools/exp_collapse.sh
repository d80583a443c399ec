#!/bin/bash
# 4-wide collapse of the host-built index: largest child first (default) against least summed area
# (RT_AMD_COLLAPSE_DP=1), rays per second and boxes per ray on the benchmark scenes.
for W in dragon1m sponza stress10m demo; do
  for G in 0 1; do
    if [ $G = 0 ]; then E="RT_AMD_COLLAPSE_DP=1"; else E="RT_AMD_NOTHING=1"; fi
    echo "== $W greedy=$G"
    python3 tools/ab.py --libs x=raytracing_amd/librt_amd.so --workloads $W --reps 2 --env $E | tail -1
    env $E python3 tools/prof_render.py --count $(python3 - <<PY
import sys; sys.path.insert(0,'tools'); import ab; print(' '.join(ab.WORK['$W']))
PY
) | grep -o "nodes/ray [0-9.]* tris/ray [0-9.]*"
  done
done
