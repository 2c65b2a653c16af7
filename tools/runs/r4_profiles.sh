# Round-4 profiles (tools/profile_bench.sh: rocprofv3 kernel stats + PMC passes each): the headline, configs[1], ssd_f32 at configs[1]'s
# and configs[2]'s literal size (matrix-core form), the ensemble share of configs[4].  Copy with tools/copy_profile.py r04_<tag>.
cd $GRAFT_REPO_ROOT
PROFILE_STEPS=100 timeout -k 10 500 bash tools/profile_bench.sh r04_c2 > gpurun_out/profile_r04_c2.log 2>&1; echo c2 rc=$?; tail -9 gpurun_out/profile_r04_c2.log
BENCH_ARGS="--views 50000 --sensor 64 --headings 16 --event-every 4" PROFILE_STEPS=300 timeout -k 10 400 bash tools/profile_bench.sh r04_c1 > gpurun_out/profile_r04_c1.log 2>&1; echo c1 rc=$?; tail -9 gpurun_out/profile_r04_c1.log
PROFILE_CMD="tools/bench_ssd_f32.py" timeout -k 10 300 bash tools/profile_bench.sh r04_ssd_f32 > gpurun_out/profile_r04_ssd.log 2>&1; echo ssd rc=$?; tail -9 gpurun_out/profile_r04_ssd.log
PROFILE_CMD="tools/bench_ssd_f32.py big 20" timeout -k 10 400 bash tools/profile_bench.sh r04_ssd_f32_c2 > gpurun_out/profile_r04_ssd_c2.log 2>&1; echo ssd_c2 rc=$?; tail -9 gpurun_out/profile_r04_ssd_c2.log
PROFILE_CMD="tools/bench_ensemble.py" timeout -k 10 400 bash tools/profile_bench.sh r04_ens > gpurun_out/profile_r04_ens.log 2>&1; echo ens rc=$?; tail -10 gpurun_out/profile_r04_ens.log
