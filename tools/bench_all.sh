#!/bin/bash
# Round results: bench.py on every named workload (+ variants), kernel-trace statistics of the default run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/bench; mkdir -p $O
python bench.py > $O/dragon1m.json 2> $O/err.log || exit 1
python bench.py --workload demo --no-cpu-baseline > $O/demo.json 2>> $O/err.log || exit 1
python bench.py --workload sponza --no-cpu-baseline > $O/sponza.json 2>> $O/err.log || exit 1
python bench.py --workload stress10m --no-cpu-baseline > $O/stress10m.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --gpu-index > $O/dragon1m_gpu_index.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --workload stress10m --gpu-index > $O/stress10m_gpu_index.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --workload sponza --gpu-index > $O/sponza_gpu_index.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --no-overlap > $O/dragon1m_no_overlap.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --no-accel > $O/dragon1m_reference_order.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --pipeline wavefront --spp-per-step 8 > $O/dragon1m_wavefront.json 2>> $O/err.log || exit 1
python bench.py --no-cpu-baseline --no-call-granularity --workload sponza --total-spp 4096 --steps 8 --warmup 1 > $O/sponza_4096spp_strong.json 2>> $O/err.log || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-call-granularity > $GRAFT_REPO_ROOT/$O/dragon1m_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_noov -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-call-granularity --no-overlap > $GRAFT_REPO_ROOT/$O/dragon1m_no_overlap_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
for f in $O/*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], "Mrays/s", d["ms_per_step"], "ms/step; kernel", r["kernel_ms_per_launch"], "ms; valu frac", r["frac"], "hbm_frac", r.get("hbm_frac"), "lane util", r.get("valu_lane_utilisation"), d.get("call_granularity"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
