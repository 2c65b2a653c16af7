# kernel timeline of the ensemble block (two chains of passes on two streams): do the scoring kernels of consecutive passes overlap?
OUT=${GRAFT_REPO_ROOT}/gpurun_out/ens_trace
rm -rf $OUT; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --agent-steps 0 > /dev/null 2> $OUT/err.log
python3 - <<PY
import csv, glob, os
t = max(glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_sad_lc22" in r["Kernel_Name"] or ("k_sad_mfma_dual" in r["Kernel_Name"] and "true, 4, 3, false, 2" in r["Kernel_Name"])]
i = idx[-12]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[i-2:i+40]:
    print("%-28s q%-3s s%-3s start %8.1f end %8.1f dur %6.1f" % (r["Kernel_Name"][9:37], r["Queue_Id"], r.get("Stream_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
