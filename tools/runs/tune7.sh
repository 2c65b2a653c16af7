cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune gpurun_out/suite
true
rc=$?
tail -25 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $BARGS --agent-steps 0 --batch-agents 0 --cpu-views 0 --secondary 0 > gpurun_out/tune/$name.json 2> gpurun_out/tune/$name.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/tune/$name.json').read().strip().splitlines()[-1])
print('%-28s step %.4f ms  kernel %.4f ms  rest %.1f us value %.3e  %s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], (d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['value'], d['roofline'].get('kernel')))
"
}
BARGS="--steps 80 --warmup 8"
run c2_a X=1
run c2_b X=1
run c2_c DEJAVU_FUSE=0
run c2_d X=1
