cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/fin
rm -rf $R/gpurun_out/fin/trace_ens
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin/trace_ens -- python3 $R/bench.py --views 20000 --sensor 64 --headings 16 --steps 20 --warmup 5 --cpu-views 0 --secondary 0 --agent-steps 200 --batch-agents 32 > $R/gpurun_out/fin/trace_ens.json 2> $R/gpurun_out/fin/trace_ens.err
echo rc=$?
f=$(ls -S $R/gpurun_out/fin/trace_ens/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"])>=10: print("%-90s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
