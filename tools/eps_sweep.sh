#!/bin/bash
# Experiment: how far can the growth of the index's triangle boxes shrink before a ray is missed?
# Runs the GPU parity tests that compare the index search with the oracle for several growths.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in -15 -17 -19 -21 -23 -25 -27; do
  echo "== RT_AMD_INDEX_EPS_LOG2=$e"
  RT_AMD_INDEX_EPS_LOG2=$e timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q --timeout 300 --timeout-method thread 2>&1 | tail -3
  RT_AMD_INDEX_EPS_LOG2=$e timeout -k 10 200 python3 tools/ab.py --libs cur=raytracing_amd/librt_amd.so --workloads dragon1m,stress10m_hi --reps 1 | tail -2
done
