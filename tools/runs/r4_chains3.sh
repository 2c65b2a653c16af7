# Two against three chains of ensemble passes (DEJAVU_CHAINS), interleaved; the new 60-heading check of the dense test first.
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -q -x -k "shared_accumulators" > gpurun_out/r4_lc22_new.log 2>&1; rc=$?; tail -3 gpurun_out/r4_lc22_new.log
[ $rc -ne 0 ] && exit $rc
DEJAVU_CHAINS=3 timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_chains3_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_chains3_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for m in 2 3; do
    DEJAVU_CHAINS=$m timeout -k 5 120 python tools/bench_ensemble.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chains=$m sensed %.4f ms uploaded %.4f ms mfma %.3f' % (d['sensed']['ms_per_ensemble_step'], d['uploaded']['ms_per_ensemble_step'], d['mfma_frac_of_peak']))"
  done
done
