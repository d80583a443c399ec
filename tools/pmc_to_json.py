#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/pmc_passes.sh into the JSON record bench.py reads:
    tools/pmc_to_json.py <pass dir> <workload> <frames per dispatch> <out.json> [kernel substring]
Per-dispatch averages of the dominant kernel; FETCH_SIZE is doubled when turned into bytes (gfx950
tallies its 128-byte fabric reads at 64 bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

GUIDE_VALU_PEAK = 1024 * 2.4e9 / 2.0 / 1e9     # G wave-instructions/s: 1024 SIMDs, one wave64 v_fma_f32 per 2 cycles at 2.4 GHz


def main():
    d, workload, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    want = sys.argv[5] if len(sys.argv) > 5 else "k_paths"
    tot, disp, dur = defaultdict(float), defaultdict(set), defaultdict(float)
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
        p = os.path.basename(os.path.dirname(f))
        for row in csv.DictReader(open(f)):
            if want not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in disp[p]:
                disp[p].add(row["Dispatch_Id"])
                dur[p] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    per = {}
    pass_of = {"FETCH_SIZE": "fetch", "WRITE_SIZE": "write", "SQ_INSTS_VALU": "sq1", "SQ_THREAD_CYCLES_VALU": "sq1",
               "SQ_WAVE_CYCLES": "sq1", "SQ_WAIT_ANY": "sq1", "SQ_WAIT_INST_ANY": "sq1", "SQ_BUSY_CYCLES": "sq1",
               "TCC_HIT_sum": "tcc", "TCC_MISS_sum": "tcc", "TCC_REQ_sum": "tcc", "SQ_INSTS_VMEM_RD": "sq2", "SQ_INSTS_LDS": "sq2",
               "SQ_INSTS_SALU": "sq2", "TCP_TOTAL_CACHE_ACCESSES_sum": "tcp", "TCP_TCC_READ_REQ_sum": "tcp", "GRBM_GUI_ACTIVE": "fetch"}
    for name, p in pass_of.items():
        if name in tot and disp[p]:
            per[name] = tot[name] / len(disp[p])
    ms = {p: dur[p] / len(disp[p]) for p in disp if disp[p]}
    valu = per.get("SQ_INSTS_VALU", 0.0)
    rec = {
        "source": "rocprofv3 --pmc, one process per pass (tools/pmc_passes.sh), MI355X; per-dispatch averages of " + want,
        "workload": workload, "kernel": want, "pipeline": "persistent", "frames_per_dispatch": frames,
        "dispatches": {p: len(v) for p, v in disp.items()},
        "kernel_ms_per_dispatch": {p: round(v, 3) for p, v in ms.items()},
        "fetch_size_kb_per_dispatch": per.get("FETCH_SIZE"), "write_size_kb_per_dispatch": per.get("WRITE_SIZE"),
        "fetch_correction": 2.0,
        "note": "FETCH_SIZE counts 64 B per L2 fabric read request and gfx950 issues 128-B requests, so the figure is doubled; "
                "Infinity Cache hits are included, so this bounds HBM traffic from above",
        "counters_per_dispatch": {k: round(v, 1) for k, v in sorted(per.items())},
    }
    if valu and "sq1" in ms:
        rate = valu / (ms["sq1"] * 1e-3) / 1e9
        rec["valu_wave_instructions_per_s_G"] = round(rate, 1)
        rec["valu_issue_frac_of_guide_peak"] = round(rate / GUIDE_VALU_PEAK, 4)
        rec["valu_lane_utilisation"] = round(per.get("SQ_THREAD_CYCLES_VALU", 0.0) / (valu * 64.0), 4)
    if per.get("TCC_HIT_sum") is not None and per.get("TCC_MISS_sum") is not None:
        rec["l2_hit_rate"] = round(per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1.0), 4)
    if per.get("FETCH_SIZE") is not None and "fetch" in ms:
        rec["traffic_beyond_l2_tbps"] = round((per["FETCH_SIZE"] * 2.0 + per.get("WRITE_SIZE", 0.0)) * 1024.0 / (ms["fetch"] * 1e-3) / 1e12, 3)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
