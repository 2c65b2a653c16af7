cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/u8
timeout -k 5 700 python -m pytest tests -m gpu -q -x > gpurun_out/u8/pytest.log 2>&1
rc=$?
tail -4 gpurun_out/u8/pytest.log
[ $rc -ne 0 ] && exit $rc
PROFILE_CMD="tools/bench_ssd_u8.py" timeout -k 10 300 bash tools/profile_bench.sh r03_ssd_u8 > gpurun_out/profile_r03_ssd_u8.log 2>&1; echo u8 rc=$?; tail -10 gpurun_out/profile_r03_ssd_u8.log
