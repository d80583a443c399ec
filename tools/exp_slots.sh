#!/bin/bash
# One frame per call, pipelined: launch slots (RT_AMD_SLOTS; the grid divisor of the round-3 experiment,
# RT_AMD_GRID_DIV, exists only in profiles/r03_exp_tail_handover.patch).
#   tools/exp_slots.sh <workload of tools/ab.py> [library]
W=${1:-dragon1m_f1}
LIB=${2:-raytracing_amd/librt_amd.so}
for cfg in "0 1" "2 1" "3 1"; do
  set -- $cfg
  if [ "$1" = 0 ]; then E="RT_AMD_GRID_DIV=$2"; else E="RT_AMD_SLOTS=$1,RT_AMD_GRID_DIV=$2"; fi
  echo "== slots $1 (0 = default) grid/$2"
  python3 tools/ab.py --libs x=$LIB --workloads $W --reps 2 --env $E | tail -1
done
