cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ht
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x -k "fp4 or matrix_core or ragged or batched or ensemble or golden or shape or ties or ssd" > gpurun_out/ht/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/ht/pytest.log
[ $rc -ne 0 ] && exit $rc
for ht in 1 2; do
  DEJAVU_HT=$ht timeout -k 10 240 python bench.py --views 20000 --sensor 64 --headings 16 --steps 20 --warmup 5 --cpu-views 0 --secondary 0 --agent-steps 0 --batch-agents 32 > gpurun_out/ht/ens_$ht.json 2> gpurun_out/ht/ens_$ht.err
  python - $ht <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ht/ens_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("HT", sys.argv[1], d.get("ensemble"))
PY
done
# a single agent with 64 headings on 100 000 views of 64x64
for ht in 1 2; do
  DEJAVU_HT=$ht timeout -k 10 240 python bench.py --views 100000 --sensor 64 --headings 64 --steps 100 --warmup 10 --cpu-views 0 --secondary 0 --agent-steps 0 --batch-agents 0 > gpurun_out/ht/h64_$ht.json 2> gpurun_out/ht/h64_$ht.err
  python - $ht <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ht/h64_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("HT", sys.argv[1], "64 headings: step %.1f us scoring_only %.1f us kernel %.1f us" % (d['ms_per_step']*1e3, d['scoring_only']['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))
PY
done
