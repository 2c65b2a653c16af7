cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_gpu_round4.py -q -x -k "pipelined or superseded" > gpurun_out/r4_pipe4.log 2>&1; rc=$?; tail -15 gpurun_out/r4_pipe4.log
exit $rc
