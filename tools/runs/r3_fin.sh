# Build first: STAMPS_SO=libdejavu_stamps_fin.so python3 tools/exp/stamps.py build -DDEJAVU_EXP_FIN
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
timeout -k 5 700 python -m pytest tests -m gpu -q -x > gpurun_out/fin/pytest.log 2>&1
rc=$?
tail -4 gpurun_out/fin/pytest.log
[ $rc -ne 0 ] && exit $rc
for shape in "50000 64 16" "500000 128 32" "100000 64 64"; do
  echo "=== $shape"
  STAMPS_FIN=1 STAMPS_SO=libdejavu_stamps_fin.so timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase 1->2\|phase 3->4\|finishing\|^exit"
done
source tools/runs/r3_ab_fn.sh
run c1_fin $C1
run c2_fin $C2
timeout -k 5 200 python tools/bench_ensemble.py 2>&1 | tail -3
