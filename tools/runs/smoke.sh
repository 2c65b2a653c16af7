cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 600 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/forcedist.json 2> gpurun_out/forcedist.err; echo forcedist rc=$?
python -c "
import json
d=json.loads(open('gpurun_out/forcedist.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['exchange'], d['known_answer_step'])
"
