cd $GRAFT_REPO_ROOT
PROFILE_STEPS=100 timeout -k 10 500 bash tools/profile_bench.sh r03_c2 > gpurun_out/profile_r03_c2.log 2>&1; echo c2 rc=$?; tail -12 gpurun_out/profile_r03_c2.log
BENCH_ARGS="--views 50000 --sensor 64 --headings 16 --event-every 4" PROFILE_STEPS=300 timeout -k 10 400 bash tools/profile_bench.sh r03_c1 > gpurun_out/profile_r03_c1.log 2>&1; echo c1 rc=$?; tail -12 gpurun_out/profile_r03_c1.log
