cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_f2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/trace_f2.err
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_f2/*/*kernel_stats.csv"), key=os.path.getsize)
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
