cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
DEJAVU_VERBOSE=1 timeout -k 10 900 python bench.py > gpurun_out/bench/full.json 2> gpurun_out/bench/full.err
echo rc=$?
tail -5 gpurun_out/bench/full.err
python -c "
import json
d=json.loads(open('gpurun_out/bench/full.json').read().strip().splitlines()[-1])
print(json.dumps(d, indent=1)[:6000])
"
