cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/code
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp4 or matrix_core or full_size or large_library or ragged" > gpurun_out/code/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/code/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in "DEJAVU_VCODE=0" "DEJAVU_VCODE=1"; do
  for shape in "50000 64 16" "500000 128 32"; do
    echo "=== $cfg $shape"
    env $cfg timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase\|^exit"
  done
done
source tools/runs/r3_ab_fn.sh
DEJAVU_VCODE=0 run c1_thermo $C1
DEJAVU_VCODE=1 run c1_code $C1
DEJAVU_VCODE=0 run c2_thermo $C2
DEJAVU_VCODE=1 run c2_code $C2
