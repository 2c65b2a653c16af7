cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lc
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp4 or matrix_core or full_size or ragged or large_library or golden" > gpurun_out/lc/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/lc/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in "DEJAVU_LC=1" "DEJAVU_LC=1 DEJAVU_VCODE=1" "DEJAVU_LC=1 DEJAVU_BALANCE=1" "DEJAVU_LC=1 DEJAVU_VCODE=1 DEJAVU_BALANCE=1"; do
  echo "=== $cfg C1"
  env $cfg timeout -k 5 60 python tools/exp/stamps.py run 50000 64 16 2>/dev/null | grep "phase\|workgroups\|exit"
done
source tools/runs/r3_ab_fn.sh
DEJAVU_LC=1 run c1_lc1 $C1
DEJAVU_LC=1 DEJAVU_VCODE=1 run c1_lc1_code $C1
DEJAVU_LC=1 DEJAVU_VCODE=1 DEJAVU_BALANCE=1 run c1_lc1_code_bal $C1
DEJAVU_LC=0 run c2_lc0 $C2
DEJAVU_LC=1 run c2_lc1 $C2
DEJAVU_LC=1 DEJAVU_VCODE=1 run c2_lc1_code $C2
DEJAVU_LC=1 DEJAVU_VCODE=1 DEJAVU_BALANCE=1 run c2_lc1_code_bal $C2
DEJAVU_LC=0 DEJAVU_VCODE=1 run c2_lc0_code $C2
