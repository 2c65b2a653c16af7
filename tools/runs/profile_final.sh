cd $GRAFT_REPO_ROOT
PROFILE_STEPS=100 timeout -k 10 500 bash tools/profile_bench.sh r02_c2 > gpurun_out/profile_c2.log 2>&1
echo c2 rc=$?
BENCH_ARGS="--views 50000 --sensor 64 --headings 16 --event-every 4" PROFILE_STEPS=300 timeout -k 10 500 bash tools/profile_bench.sh r02_c1 > gpurun_out/profile_c1.log 2>&1
echo c1 rc=$?
