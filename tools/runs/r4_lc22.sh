# k_sad_lc22 (two view groups x two heading tiles per consumer): the ensemble / batch / wide tests with it and without, then the
# ensemble block's time both ways, interleaved.
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -m pytest tests -q -x -m gpu -k "batch or ensemble or wide or ninety" > gpurun_out/r4_lc22_tests1.log 2>&1; rc=$?; tail -12 gpurun_out/r4_lc22_tests1.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for m in 0 1; do
    DEJAVU_LC22=$m timeout -k 5 120 python tools/bench_ensemble.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lc22=$m sensed %.4f ms uploaded %.4f ms mfma %.3f' % (d['sensed']['ms_per_ensemble_step'], d['uploaded']['ms_per_ensemble_step'], d['mfma_frac_of_peak']))"
  done
done
