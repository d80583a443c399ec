#!/bin/bash
# Runs tools/micro/gather_bytes plain and under the two counter passes that show the bytes per record:
#   tools/micro/gather_bytes.sh <outdir>
OUT=$1; mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 "$ROOT/tools/micro/gather_bytes" 4 256 > "$OUT/plain.log" 2>&1 || { echo "plain run failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- "$ROOT/tools/micro/gather_bytes" 4 64 > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/rdreq" -o p -- "$ROOT/tools/micro/gather_bytes" 4 64 > "$OUT/rdreq.log" 2>&1 || { echo "rdreq pass failed"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_gather" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:40]})[r["Counter_Name"]] = float(r["Counter_Value"])
print(open(out + "/plain.log").read())
for d, v in rows.items():
    print(d, v)
PY
