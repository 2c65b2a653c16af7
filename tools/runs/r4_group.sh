cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_group.py -q -x > gpurun_out/r4_group_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r4_group_tests.log
exit $rc
