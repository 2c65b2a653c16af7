cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -x -m gpu -k "ensemble or experiment or error_metrics" > gpurun_out/r4_ensmetrics.log 2>&1; rc=$?; tail -25 gpurun_out/r4_ensmetrics.log
exit $rc
