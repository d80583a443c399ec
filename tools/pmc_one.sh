#!/bin/bash
# One rocprofv3 counter pass of tools/prof_render.py: tools/pmc_one.sh <outdir> <name> "<counters>" [prof_render args...]
OUT=$1; NAME=$2; CTRS=$3; shift 3
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/$NAME" -o p -- python3 "$ROOT/tools/prof_render.py" "$@" > "$OUT/$NAME.log" 2>&1 || { echo "pass $NAME failed"; tail -3 "$OUT/$NAME.log"; exit 1; }
