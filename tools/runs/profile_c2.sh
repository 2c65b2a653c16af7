cd $GRAFT_REPO_ROOT
PROFILE_STEPS=100 timeout -k 10 1100 bash tools/profile_bench.sh r02_c2 > gpurun_out/profile_c2.log 2>&1
echo rc=$?
tail -5 gpurun_out/profile_c2.log
