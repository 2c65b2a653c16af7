cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -x -m gpu -k "deferred_error or trajector or error_metrics or pipelined or experiment or agent" > gpurun_out/r4_defer.log 2>&1; rc=$?; tail -15 gpurun_out/r4_defer.log
exit $rc
