# The round's last build: the whole GPU suite, bench.py with its defaults, the agent trace.
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/r4_suite3.log 2>&1; rc=$?; tail -3 gpurun_out/r4_suite3.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r4_bench4.json 2> gpurun_out/r4_bench4.err; echo bench rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r4_bench4.json')); s=d['roofline']['secondary']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_measured_ceiling'))
for k in sorted(s): print(k, s[k])"
bash tools/runs/agent_trace.sh > gpurun_out/r4_agent_trace.txt 2>&1; head -4 gpurun_out/r4_agent_trace.txt
