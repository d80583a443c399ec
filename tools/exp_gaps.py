#!/usr/bin/env python3
"""Experiment: where the time of a call goes on the GPU's side.  Reads the kernel trace of
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/prof_render.py --per-call 1 --sync-each-call ...
and prints, for the last calls, every kernel with its duration and the idle gap before it."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tail = rows[-n:]
prev = None
for s, e, name in tail:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"gap {gap:8.1f} us   {name:40s} {(e - s) / 1e3:9.1f} us")
    prev = e
# per call: from one k_paths start to the next
starts = [s for s, e, nme in rows if nme.startswith("k_paths")]
ends = [e for s, e, nme in rows if nme.startswith("k_paths")]
if len(starts) > 20:
    per = [(starts[i + 1] - starts[i]) / 1e3 for i in range(len(starts) - 17, len(starts) - 1)]
    dur = [(ends[i] - starts[i]) / 1e3 for i in range(len(starts) - 17, len(starts) - 1)]
    print(f"last 16 calls: k_paths start to start {sum(per) / len(per):.1f} us, k_paths itself {sum(dur) / len(dur):.1f} us")
