#!/bin/bash
# Registers, spills and occupancy of the path and extend kernel variants as the compiler reports them:
#   tools/kernel_resources.sh [source file] [extra compiler flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-$ROOT/raytracing_amd/csrc/rt_device.hip}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
  -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" --cuda-device-only -I"$ROOT/raytracing_amd/csrc" \
  -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /TotalSGPRs:/ {s=$4} /VGPRs Spill:/ {vs=$5} /SGPRs Spill:/ {ss=$5} /ScratchSize/ {sc=$5} /Occupancy/ {oc=$5} /LDS Size/ {print name, "vgpr", v, "sgpr", s, "vgpr_spill", vs, "sgpr_spill", ss, "scratch_bytes", sc, "occ", oc}' |
  grep -E "k_paths|k_extend_fast" | sed -E 's/_ZN12_GLOBAL__N_1[0-9]+//; s/EvN3rtd.*WorkCountersE//; s/EEvN3rtd.*$//'
