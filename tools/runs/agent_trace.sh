OUT=${GRAFT_REPO_ROOT}/gpurun_out/agent_trace
rm -rf $OUT; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 400 > /dev/null 2> $OUT/err.log
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/**/*_kernel_stats.csv", recursive=True), key=os.path.getsize)
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 300: print("%-70s calls %5s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
t = max(glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fold" in r["Kernel_Name"]]
i = idx[-50]
for a, b in zip(rows[i-6:i+8], rows[i-5:i+9]):
    print(a["Kernel_Name"][:36], "dur %.1f" % ((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3), "gap %.1f" % ((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3))
PY
