cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x -k "two_group_body" > gpurun_out/r4_lc22c.log 2>&1; rc=$?; tail -15 gpurun_out/r4_lc22c.log
exit $rc
