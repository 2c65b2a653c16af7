# Final round-4 evidence: the GPU suite, bench.py with its defaults, the two ssd_f32 profiles of the final tail, the agent trace.
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r4_suite_final.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r4_suite_final.log
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench rc=$?"
PROFILE_CMD="tools/bench_ssd_f32.py" timeout -k 10 300 bash tools/profile_bench.sh r04_ssd_f32 > gpurun_out/profile_r04_ssd.log 2>&1; echo ssd rc=$?; tail -6 gpurun_out/profile_r04_ssd.log
PROFILE_CMD="tools/bench_ssd_f32.py big 20" timeout -k 10 400 bash tools/profile_bench.sh r04_ssd_f32_c2 > gpurun_out/profile_r04_ssd_c2.log 2>&1; echo ssd_c2 rc=$?; tail -6 gpurun_out/profile_r04_ssd_c2.log
bash tools/runs/agent_trace.sh > gpurun_out/r4_agent_trace.txt 2>&1; head -4 gpurun_out/r4_agent_trace.txt
