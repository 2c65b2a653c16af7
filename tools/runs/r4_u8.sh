# ssd_u8 after the first rows were moved in front of the LDS copy: its tests, its timing and profile; then the round's final evidence.
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests -m gpu -q -x -k "ssd or u8" > gpurun_out/r4_u8_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_u8_tests.log
[ $rc -ne 0 ] && exit $rc
PROFILE_CMD="tools/bench_ssd_u8.py" timeout -k 10 300 bash tools/profile_bench.sh r04_ssd_u8 > gpurun_out/profile_r04_ssd_u8.log 2>&1; echo u8 rc=$?; tail -8 gpurun_out/profile_r04_ssd_u8.log
PROFILE_CMD="tools/bench_ssd_u8.py 200000 128 16" timeout -k 10 300 bash tools/profile_bench.sh r04_ssd_u8_big > gpurun_out/profile_r04_ssd_u8_big.log 2>&1; echo u8big rc=$?; tail -8 gpurun_out/profile_r04_ssd_u8_big.log
