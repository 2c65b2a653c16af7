cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite gpurun_out/tune
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -8 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 30 --warmup 5 --cpu-views 0 --secondary 0 --batch-agents 32 > gpurun_out/tune/agent.json 2> gpurun_out/tune/agent.err
python -c "
import json
d=json.loads(open('gpurun_out/tune/agent.json').read().strip().splitlines()[-1])
print('agent', d['agent']['nav_steps_per_s'], d['agent']['nav_steps_per_s_fake'], d['agent']['ensemble_of_32_nav_steps_per_s'], d['agent']['median_step_us_and_steps_over_4x_median'])
print('ensemble', d.get('ensemble'))
"
