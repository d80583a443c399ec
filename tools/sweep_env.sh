# source me: run <name> VAR=value... runs bench.py (pipeline from $PIPE, default persistent) and prints Mrays/s and kernel ms
run() {
  name=$1; shift
  out=$(env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline ${PIPE:-persistent} --steps 6 --warmup 1 $EXTRA 2>gpurun_out/sweep_$name.err) || { echo "$name FAILED"; return 1; }
  python - "$name" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[2].strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['roofline']['kernel_ms_per_launch'])
PY
}
