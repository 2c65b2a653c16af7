cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -q -x -k "shared_accumulators" > gpurun_out/r4_lc22_new.log 2>&1; rc=$?; tail -25 gpurun_out/r4_lc22_new.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/r4_lc22_suite.log 2>&1; rc=$?; tail -3 gpurun_out/r4_lc22_suite.log
exit $rc
