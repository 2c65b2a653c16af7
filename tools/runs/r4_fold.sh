# The folds of an ensemble group in one launch (k_fold_multi) against one launch per pass (DEJAVU_FOLD_MULTI=0): the ensemble tests
# both ways, then the ensemble block's time, interleaved.
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_fold_tests1.log 2>&1; rc=$?; tail -2 gpurun_out/r4_fold_tests1.log
[ $rc -ne 0 ] && exit $rc
DEJAVU_FOLD_MULTI=0 timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_fold_tests0.log 2>&1; rc=$?; tail -2 gpurun_out/r4_fold_tests0.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for m in 0 1; do
    DEJAVU_FOLD_MULTI=$m timeout -k 5 120 python tools/bench_ensemble.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fold_multi=$m sensed %.4f ms uploaded %.4f ms mfma %.3f' % (d['sensed']['ms_per_ensemble_step'], d['uploaded']['ms_per_ensemble_step'], d['mfma_frac_of_peak']))"
  done
done
