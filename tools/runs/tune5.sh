cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $BARGS --agent-steps 0 --batch-agents 0 --cpu-views 0 --secondary 0 > gpurun_out/tune/$name.json 2> gpurun_out/tune/$name.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/tune/$name.json').read().strip().splitlines()[-1])
print('%-28s step %.4f ms  kernel %.4f ms  rest %.1f us value %.3e' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], (d['ms_per_step']-d['roofline']['kernel_ms'])*1e3, d['value']))
"
}
BARGS="--steps 50 --warmup 5"
run c2_default X=1
run c2_vb2 DEJAVU_FINISH_VB=2
run c2_vb4 DEJAVU_FINISH_VB=4
run c2_finish0 DEJAVU_FINISH=0
BARGS="--views 200000 --sensor 64 --headings 16 --steps 200 --warmup 20 --event-every 4"
run f200k_a16_default X=1
run f200k_a16_vb2 DEJAVU_FINISH_VB=2
run f200k_a16_finish0 DEJAVU_FINISH=0
BARGS="--views 50000 --sensor 64 --headings 16 --steps 300 --warmup 30 --event-every 4"
run c1_default X=1
