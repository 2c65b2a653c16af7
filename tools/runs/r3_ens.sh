cd $GRAFT_REPO_ROOT
PROFILE_CMD="tools/bench_ensemble.py" timeout -k 10 400 bash tools/profile_bench.sh r03_ens > gpurun_out/profile_r03_ens.log 2>&1; echo ens rc=$?; tail -12 gpurun_out/profile_r03_ens.log
