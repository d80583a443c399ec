#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel and counter: tools/pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_extend"
    tot = defaultdict(float)
    cnt = defaultdict(int)
    dur = defaultdict(float)
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
        seen = set()
        for row in csv.DictReader(open(f)):
            if want not in row["Kernel_Name"]:
                continue
            key = (os.path.basename(os.path.dirname(f)), row["Counter_Name"])
            tot[key] += float(row["Counter_Value"])
            did = (f, row["Dispatch_Id"])
            if did not in seen:
                seen.add(did)
                cnt[key[0]] += 1
                dur[key[0]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    for p in sorted(cnt):
        print(f"[{p}] {cnt[p]} dispatches of *{want}*, {dur[p]:.3f} ms total")
        for (pp, name), v in sorted(tot.items()):
            if pp == p:
                print(f"   {name:36s} {v:18.0f}   per dispatch {v / cnt[p]:16.1f}")


if __name__ == "__main__":
    main()
