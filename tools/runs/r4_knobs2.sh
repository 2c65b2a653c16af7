cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x -k "chain_knobs" > gpurun_out/r4_knobs2.log 2>&1; rc=$?; tail -15 gpurun_out/r4_knobs2.log
exit $rc
