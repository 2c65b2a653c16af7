cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -x -m gpu -k "error_metrics or trajector or agent or pipelined or experiment" > gpurun_out/r4_prep_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4_prep_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/runs/agent_trace.sh > gpurun_out/r4_agent_trace.txt 2>&1; head -4 gpurun_out/r4_agent_trace.txt
bash tools/runs/r4_pipe3.sh 2>&1 | grep -E "nav_steps|median_step_us"
