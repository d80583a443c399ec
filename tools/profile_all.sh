#!/bin/bash
# PMC passes + kernel-trace statistics for the benchmark workloads; results under gpurun_out/prof/<workload>.
# usage: tools/profile_all.sh [workload ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
declare -A ARGS FR
ARGS[dragon1m]="--kind 0 --level 8 --copies 1 --frames 64 --warmup 64"; FR[dragon1m]=64
ARGS[sponza]="--kind 1 --level 64 --width 3840 --height 2160 --frames 16 --warmup 16"; FR[sponza]=16
ARGS[stress10m]="--kind 0 --level 8 --copies 8 --bounces 8 --frames 64 --warmup 64"; FR[stress10m]=64
ARGS[demo]="--demo --frames 64 --warmup 64"; FR[demo]=64
for w in "${@:-dragon1m sponza stress10m demo}"; do
  for ww in $w; do
    out=$ROOT/gpurun_out/prof/$ww
    rm -rf "$out"; mkdir -p "$out"
    bash tools/pmc_passes.sh "$out" ${ARGS[$ww]} || exit 1
    cd "$ROOT"
    python3 tools/pmc_summary.py "$out" k_paths > "$out/summary.txt"
    python3 tools/pmc_to_json.py "$out" $ww ${FR[$ww]} "$out/pmc.json" > /dev/null
    echo "== $ww"; cat "$out/summary.txt" | head -60
  done
done
