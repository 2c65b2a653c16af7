# What bounds k_sad_lc22's stage loop: timing-only builds of the library (tools/exp/ab/lc22_<bits>.so, -DDEJAVU_EXP22=<bits>: bit 0 no
# MFMA, 1 no library rows, 2 no coefficient rows, 3 no masks, 4 no LDS operand reads after a stage's first unit, 5 the finishing's barriers only), one chain of passes
# (DEJAVU_CHAINS=1: the kernels run alone), the kernel's duration from the rocprofv3 trace.
ROOT=$GRAFT_REPO_ROOT
L=$ROOT/navigation-by-deja-vu_amd/csrc/libdejavu_hip.so
cp $L /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
export DEJAVU_CHAINS=1
for v in base 32 33 34 36 38 40 48 63; do
  if [ $v = base ]; then cp /tmp/keep.so $L; else cp $ROOT/tools/exp/ab/lc22_$v.so $L; fi
  OUT=$ROOT/gpurun_out/lc22_exp/$v; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/exp/ens_time.py > $OUT/out.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*_kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
for r in rows:
    if "k_sad_lc22" in r["Name"]: print("variant %-5s k_sad_lc22 calls %4s avg %8.1f us min %8.1f" % ("$v", r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
cp /tmp/keep.so $L
