# Evidence of the round's last build: the ensemble profile (k_sad_lc22), the agent trace, bench.py with its defaults.
cd $GRAFT_REPO_ROOT
PROFILE_CMD="tools/bench_ensemble.py" timeout -k 10 400 bash tools/profile_bench.sh r04_ens > gpurun_out/profile_r04_ens.log 2>&1; echo ens rc=$?; tail -12 gpurun_out/profile_r04_ens.log
bash tools/runs/ens_trace.sh > gpurun_out/r04_ens_trace.txt 2>&1; tail -3 gpurun_out/r04_ens_trace.txt
bash tools/runs/agent_trace.sh > gpurun_out/r4_agent_trace.txt 2>&1; head -4 gpurun_out/r4_agent_trace.txt
python bench.py > gpurun_out/r4_bench3.json 2> gpurun_out/r4_bench3.err; echo bench rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r4_bench3.json')); s=d['roofline']['secondary']
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('c1_step_us','c1_kernel_us','ens_ms','ens_uploaded_ms','ens_mfma_frac','agent_steps_per_s','agent_steps_per_s_fake','agent_ensemble_of_32_steps_per_s'): print(k, s.get(k))"
