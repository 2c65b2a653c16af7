# run <name> <bench args...>: one bench.py run (headline only), one line of result; env in front of the call
mkdir -p gpurun_out/ab
run() {
  name=$1; shift
  timeout -k 10 240 python bench.py "$@" --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || echo "$name FAILED"
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/ab/%s.json'%n).read().strip().splitlines()[-1])
    print("%-22s step %8.1f us  scoring_only %8.1f us  kernel %8.1f us  frac %.3f" % (n, d['ms_per_step']*1e3, d['scoring_only']['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))
except Exception as e:
    print(n, "no result", e)
PY
}
C1="--views 50000 --sensor 64 --headings 16 --steps 300 --warmup 30 --event-every 4"
C2="--steps 40 --warmup 5"
