# The agent's begun / ended steps: the new tests, the whole GPU suite, the agent's rate with and without (bench.py's agent block).
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_gpu_round4.py -q -x -k "pipelined or superseded" > gpurun_out/r4_pipe_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4_pipe_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 900 python -m pytest tests -q -x -m gpu > gpurun_out/r4_pipe_suite.log 2>&1; rc=$?; tail -3 gpurun_out/r4_pipe_suite.log
[ $rc -ne 0 ] && exit $rc
for p in 1 0 1 0; do
  DEJAVU_AGENT_PIPELINE=$p python bench.py --steps 3 --warmup 1 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('agent',{}); print('pipeline=$p', {k:a.get(k) for k in ('nav_steps_per_s','nav_steps_per_s_fake','median_step_us','ensemble_of_32_nav_steps_per_s')})"
done
python bench.py > gpurun_out/r4_bench_pipe.json 2> gpurun_out/r4_bench_pipe.err; echo bench rc=$?
