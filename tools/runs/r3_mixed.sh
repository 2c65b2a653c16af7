cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mixed
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ragged or fp4 or matrix_core or golden or batched" > gpurun_out/mixed/pytest.log 2>&1
rc=$?
tail -8 gpurun_out/mixed/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/runs/r3_mixed2.sh
