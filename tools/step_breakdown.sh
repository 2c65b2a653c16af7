# usage (through gpurun, from the repo root): bash tools/step_breakdown.sh
# rocprofv3 kernel trace of a short bench run; prints the average duration of the step's kernels.
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/breakdown
rm -rf $OUT; mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 --batch-agents 0 "$@" > /dev/null 2> $OUT/err.log
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/**/*_kernel_stats.csv", recursive=True), key=os.path.getsize)
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-60s calls %5s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
