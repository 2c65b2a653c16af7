OUT=${GRAFT_REPO_ROOT}/gpurun_out/fake_trace
rm -rf $OUT; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/exp/fake_walk.py > $OUT/out.log 2>&1
tail -3 $OUT/out.log
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/**/*_kernel_stats.csv", recursive=True), key=os.path.getsize)
for r in csv.DictReader(open(f)):
    print("%-80s calls %6s avg %10.1f us max %10.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
