cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
DEJAVU_VERBOSE=1 timeout -k 10 300 python bench.py --views 500000 --sensor 128 --headings 32 --agent-steps 0 --batch-agents 0 --cpu-views 0 --steps 50 --warmup 5 > gpurun_out/suite/c2.json 2> gpurun_out/suite/c2.err && \
DEJAVU_VERBOSE=1 timeout -k 10 300 python bench.py --views 50000 --sensor 64 --headings 16 --agent-steps 0 --batch-agents 0 --cpu-views 0 --steps 200 --warmup 20 > gpurun_out/suite/c1.json 2> gpurun_out/suite/c1.err
grep dejavu gpurun_out/suite/c2.err gpurun_out/suite/c1.err
python -c "
import json
for f in ('c2','c1'):
    d=json.loads(open('gpurun_out/suite/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['config']['workgroup_shape'], d['roofline']['kernel_ms'])
"
