cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_gpu_round4.py -q -x -k "worked_out_when_read" > gpurun_out/r4_lazy2.log 2>&1; rc=$?; tail -12 gpurun_out/r4_lazy2.log
exit $rc
