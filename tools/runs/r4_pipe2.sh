cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -x -m gpu -k "pipelined or superseded or trajector or agent or experiment" > gpurun_out/r4_pipe2_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4_pipe2_tests.log
[ $rc -ne 0 ] && exit $rc
for p in 1 0 1 0 1; do
  DEJAVU_AGENT_PIPELINE=$p python bench.py --steps 3 --warmup 1 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('agent',{}); print('pipeline=$p', {k:a.get(k) for k in ('nav_steps_per_s','nav_steps_per_s_fake')})"
done
