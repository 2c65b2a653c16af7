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
print('%-28s step %.4f ms  kernel %.4f ms  value %.3e' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))
"
}
BARGS="--steps 50 --warmup 5"
run c2_finish2 DEJAVU_FINISH=2
run c2_finish0 DEJAVU_FINISH=0
BARGS="--views 50000 --sensor 64 --headings 32 --steps 300 --warmup 30 --event-every 4"
run c1_a32_finish2 DEJAVU_FINISH=2
run c1_a32_finish0 DEJAVU_FINISH=0
BARGS="--views 20000 --sensor 64 --headings 32 --steps 300 --warmup 30 --event-every 4"
run f20k_a32_finish2 DEJAVU_FINISH=2
run f20k_a32_finish0 DEJAVU_FINISH=0
BARGS="--views 50000 --sensor 64 --headings 16 --steps 300 --warmup 30 --event-every 4"
run c1_finish2 DEJAVU_FINISH=2
run c1_finish0 DEJAVU_FINISH=0
