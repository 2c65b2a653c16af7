cd $GRAFT_REPO_ROOT
BENCH_ARGS="--full-range-s" PROFILE_STEPS=40 timeout -k 10 500 bash tools/profile_bench.sh r03_c2_full_s > gpurun_out/profile_r03_fulls.log 2>&1; echo fulls rc=$?; tail -12 gpurun_out/profile_r03_fulls.log
PROFILE_CMD="tools/bench_ssd_f32.py" timeout -k 10 300 bash tools/profile_bench.sh r03_ssd > gpurun_out/profile_r03_ssd.log 2>&1; echo ssd rc=$?; tail -10 gpurun_out/profile_r03_ssd.log
PROFILE_CMD="tools/bench_ensemble.py" timeout -k 10 400 bash tools/profile_bench.sh r03_ens > gpurun_out/profile_r03_ens.log 2>&1; echo ens rc=$?; tail -12 gpurun_out/profile_r03_ens.log
