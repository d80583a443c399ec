#!/bin/bash
# L2 hit rate and traffic beyond L2 of k_paths for two builds of the library (RT_AMD_LIBRARY).
# usage: tools/l2_ab.sh <outdir> <libA> <libB>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $2 $3; do for w in "dragon1m --copies 1" "stress10m --copies 8 --bounces 8"; do set -- $w; name=$1; shift
tag=$(basename $lib .so)_$name
RT_AMD_LIBRARY=$ROOT/$lib timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/$tag -o p -- python3 $ROOT/tools/prof_render.py --frames 8 --warmup 1 "$@" > $OUT/$tag.log 2>&1 || exit 1
python3 - $OUT/$tag $tag <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_paths" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
h, m = tot.get("TCC_HIT_sum", 0), tot.get("TCC_MISS_sum", 0)
print(sys.argv[2], "L2 hit rate %.3f" % (h / max(h + m, 1)), "requests %.3e" % tot.get("TCC_REQ_sum", 0), "reads beyond L2 %.3e" % tot.get("TCC_EA0_RDREQ_sum", 0))
PY
done; done
