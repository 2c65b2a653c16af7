cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/split
timeout -k 5 150 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp4_form or off_level" > gpurun_out/split/pytest0.log 2>&1
rc=$?
tail -3 gpurun_out/split/pytest0.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 700 python -m pytest tests -m gpu -q -x > gpurun_out/split/pytest.log 2>&1
rc=$?
tail -3 gpurun_out/split/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 200 python tools/bench_ensemble.py 2>&1 | tail -1 | cut -c1-160
source tools/runs/r3_ab_fn.sh
run c1_split $C1
run c2_split $C2
