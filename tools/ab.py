#!/usr/bin/env python3
"""A/B measurement of library builds on the benchmark workloads:
    python3 tools/ab.py --libs base=raytracing_amd/librt_amd.so,x=expl/librt_x.so --workloads dragon1m,sponza --reps 3
Runs tools/prof_render.py once per (library, workload, repetition), interleaved, one process each,
and prints the median Mrays/s, the kernel time and the checksum of the image (which must not
differ between builds)."""
import argparse
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = {
    "dragon1m": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "8"],
    "demo": ["--demo", "--frames", "64", "--warmup", "8"],
    "sponza": ["--kind", "1", "--level", "64", "--width", "3840", "--height", "2160", "--frames", "21", "--warmup", "4"],
    "stress10m": ["--kind", "0", "--level", "8", "--copies", "8", "--bounces", "8", "--frames", "64", "--warmup", "8"],
    "dragon1m_hi": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "8", "--host-index"],
    "stress10m_hi": ["--kind", "0", "--level", "8", "--copies", "8", "--bounces", "8", "--frames", "64", "--warmup", "8", "--host-index"],
    "dragon1m_gi": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "8", "--gpu-index"],
    "sponza_gi": ["--kind", "1", "--level", "64", "--width", "3840", "--height", "2160", "--frames", "21", "--warmup", "4", "--gpu-index"],
    "stress10m_gi": ["--kind", "0", "--level", "8", "--copies", "8", "--bounces", "8", "--frames", "64", "--warmup", "8", "--gpu-index"],
    "dragon1m_f8": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "32", "--per-call", "8"],
    "dragon1m_f1": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "8", "--per-call", "1"],
    "dragon1m_f2": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "8", "--per-call", "2"],
    "demo_f1": ["--demo", "--frames", "64", "--warmup", "8", "--per-call", "1"],
    "stress10m_f1": ["--kind", "0", "--level", "8", "--copies", "8", "--bounces", "8", "--frames", "64", "--warmup", "8", "--per-call", "1"],
    "small_wf": ["--kind", "0", "--level", "4", "--width", "160", "--height", "90", "--frames", "24", "--warmup", "8", "--pipeline", "0"],
    "small": ["--kind", "0", "--level", "4", "--width", "160", "--height", "90", "--frames", "24", "--warmup", "8"],
    "dragon1m_wf": ["--kind", "0", "--level", "8", "--copies", "1", "--frames", "64", "--warmup", "64", "--pipeline", "0"],
    "stress10m_wf": ["--kind", "0", "--level", "8", "--copies", "8", "--bounces", "8", "--frames", "64", "--warmup", "64", "--pipeline", "0"],
    "sponza_wf": ["--kind", "1", "--level", "64", "--width", "3840", "--height", "2160", "--frames", "16", "--warmup", "16", "--pipeline", "0"],
    "demo_wf": ["--demo", "--frames", "64", "--warmup", "64", "--pipeline", "0"],
    "sponza_f1": ["--kind", "1", "--level", "64", "--width", "3840", "--height", "2160", "--frames", "24", "--warmup", "4", "--per-call", "1"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--workloads", default="dragon1m")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--env", default="", help="extra VAR=value,... for every run")
    a = ap.parse_args()
    libs = [kv.split("=", 1) for kv in a.libs.split(",")]
    res = {}
    for w in a.workloads.split(","):
        for rep in range(a.reps):
            for name, path in libs:
                env = dict(os.environ)
                env["RT_AMD_LIBRARY"] = os.path.join(ROOT, path)
                for kv in filter(None, a.env.split(",")):
                    k, v = kv.split("=", 1)
                    env[k] = v
                try:
                    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_render.py"), "--checksum"] + WORK[w],
                                         env=env, capture_output=True, text=True, timeout=240)
                except subprocess.TimeoutExpired:
                    print(f"{w} {name}: TIMEOUT", flush=True)
                    return 1
                if out.returncode != 0:
                    print(f"{w} {name}: FAILED rc={out.returncode}\n{out.stderr[-800:]}", flush=True)
                    return 1
                m = re.search(r"-> ([0-9.]+) Mrays/s; dominant kernel ([0-9.]+) ms over (\d+) launches.*checksum (\w+)", out.stdout)
                res.setdefault((w, name), []).append((float(m.group(1)), float(m.group(2)) / int(m.group(3)), m.group(4)))
                print(f"{w:12s} {name:10s} rep {rep}: {m.group(1)} Mrays/s, {float(m.group(2)) / int(m.group(3)):.3f} ms/launch, image {m.group(4)}", flush=True)
    print("---- medians")
    for (w, name), v in res.items():
        sums = {x[2] for x in v}
        print(f"{w:12s} {name:10s} {statistics.median(x[0] for x in v):9.1f} Mrays/s  {statistics.median(x[1] for x in v):8.3f} ms/launch  image {','.join(sorted(sums))}")
    for w in a.workloads.split(","):
        if len({x[2] for (ww, _), v in res.items() if ww == w for x in v}) != 1:
            print(f"IMAGES DIFFER on {w}")
            return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
