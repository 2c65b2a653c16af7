cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "ssd or u8 or exact or f32 or golden or tie or parity or errors" > gpurun_out/r4_tail_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4_tail_tests.log
[ $rc -ne 0 ] && exit $rc
PROFILE_CMD="tools/bench_ssd_u8.py" timeout -k 10 300 bash tools/profile_bench.sh r04_ssd_u8 > gpurun_out/profile_r04_ssd_u8.log 2>&1; grep -E "calls" gpurun_out/profile_r04_ssd_u8.log | head -6
python tools/bench_ssd_u8.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ssd_u8 step us', d['ms_per_step']*1e3, 'kernel', d['roofline']['kernel_ms']*1e3)"
