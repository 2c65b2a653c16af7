cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" DEJAVU_SHAPE=6 timeout -k 10 300 python bench.py $BARGS --agent-steps 0 --batch-agents 0 --cpu-views 0 --secondary 0 > gpurun_out/tune/$name.json 2> gpurun_out/tune/$name.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/tune/$name.json').read().strip().splitlines()[-1])
print('%-28s step %.4f ms  kernel %.4f ms  value %.3e' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))
"
}
BARGS="--steps 50 --warmup 5"
run c2_t2sk2 DEJAVU_MFMA_TILES=2
run c2_t2sk4 DEJAVU_MFMA_TILES=2 DEJAVU_MFMA_VARIANT=1
run c2_t1sk4 DEJAVU_MFMA_TILES=1
run c2_t1sk8 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_VARIANT=1
run c2_t2sk2_chunk2 DEJAVU_MFMA_TILES=2 DEJAVU_MFMA_CHUNK=2
BARGS="--views 50000 --sensor 64 --headings 16 --steps 300 --warmup 30 --event-every 4"
run c1_t1sk4_auto DEJAVU_MFMA_TILES=1
run c1_t1sk4_c1 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_CHUNK=1
run c1_t1sk4_c2 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_CHUNK=2
run c1_t1sk4_c3 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_CHUNK=3
run c1_t1sk8_c1 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_CHUNK=1 DEJAVU_MFMA_VARIANT=1
run c1_t2sk2_c3 DEJAVU_MFMA_TILES=2 DEJAVU_MFMA_CHUNK=3
BARGS="--views 100000 --sensor 64 --headings 64 --steps 100 --warmup 10"
run b64_t1 DEJAVU_MFMA_TILES=1
run b64_t2 DEJAVU_MFMA_TILES=2
run b64_t1_v1 DEJAVU_MFMA_TILES=1 DEJAVU_MFMA_VARIANT=1
