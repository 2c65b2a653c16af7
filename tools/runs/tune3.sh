cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune
run() {
  name=$1; shift
  env "$@" DEJAVU_SHAPE=6 timeout -k 10 300 python bench.py $BARGS --agent-steps 0 --batch-agents 0 --cpu-views 0 --secondary 0 > gpurun_out/tune/$name.json 2> gpurun_out/tune/$name.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/tune/$name.json').read().strip().splitlines()[-1])
print('%-28s step %.4f ms  kernel %.4f ms  value %.3e  %s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['known_answer_step'][:2]))
" || tail -3 gpurun_out/tune/$name.err
}
BARGS="--steps 50 --warmup 5"
run c2_ring_125 DEJAVU_MFMA_VARIANT=3
run c2_ring_126 DEJAVU_MFMA_VARIANT=5
run c2_ring_124 DEJAVU_MFMA_VARIANT=6
run c2_ring_123 DEJAVU_MFMA_VARIANT=7
BARGS="--views 50000 --sensor 64 --headings 16 --steps 300 --warmup 30 --event-every 4"
run c1_ring_412 DEJAVU_MFMA_VARIANT=3
run c1_ring_215 DEJAVU_MFMA_VARIANT=5
run c1_ring_213 DEJAVU_MFMA_VARIANT=6
run c1_ring_119 DEJAVU_MFMA_VARIANT=7
BARGS="--views 100000 --sensor 64 --headings 64 --steps 100 --warmup 10"
run b64_v0 DEJAVU_MFMA_VARIANT=0
run b64_ring3 DEJAVU_MFMA_VARIANT=3
run b64_ring5 DEJAVU_MFMA_VARIANT=5
