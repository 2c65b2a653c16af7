cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 50 --warmup 5 --cpu-views 0 > gpurun_out/suite/full.json 2> gpurun_out/suite/full.err
python -c "
import json
d=json.loads(open('gpurun_out/suite/full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])
print('c1', d['configs1']['value'], d['configs1']['ms_per_step'], d['configs1']['roofline']['kernel_ms'])
print('agent', d['agent'])
print('ens', d['ensemble'])
"
