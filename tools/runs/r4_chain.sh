# How a chain of ensemble passes is laid out on its stream (DEJAVU_CHAIN_ORDER): 0 = [preparations][scoring kernels][folds],
# 1 = [preparation, scoring] per pass then the folds, 2 = [preparation, scoring, fold] per pass.  Interleaved repetitions.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for o in 0 1 2; do
    DEJAVU_CHAIN_ORDER=$o timeout -k 5 120 python tools/bench_ensemble.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('order=$o sensed %.4f ms uploaded %.4f ms mfma %.3f' % (d['sensed']['ms_per_ensemble_step'], d['uploaded']['ms_per_ensemble_step'], d['mfma_frac_of_peak']))"
  done
done
timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_chain_tests0.log 2>&1; tail -2 gpurun_out/r4_chain_tests0.log
DEJAVU_CHAIN_ORDER=1 timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_chain_tests1.log 2>&1; tail -2 gpurun_out/r4_chain_tests1.log
DEJAVU_CHAIN_ORDER=2 timeout -k 5 300 python -m pytest tests -q -x -m gpu -k "batch or ensemble" > gpurun_out/r4_chain_tests2.log 2>&1; tail -2 gpurun_out/r4_chain_tests2.log
