cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
for i in 1 2; do
timeout -k 10 600 python bench.py --cpu-views 0 --batch-agents 0 > gpurun_out/bench/a$i.json 2> gpurun_out/bench/a$i.err
python -c "
import json
d=json.loads(open('gpurun_out/bench/a$i.json').read().strip().splitlines()[-1])
print($i, d['value'], d['roofline']['kernel_ms'], {k:v for k,v in d['agent'].items() if k!='what'})
"
done
timeout -k 10 600 python bench.py --cpu-views 0 --batch-agents 0 --secondary 0 --steps 20 > gpurun_out/bench/a3.json 2> gpurun_out/bench/a3.err
python -c "
import json
d=json.loads(open('gpurun_out/bench/a3.json').read().strip().splitlines()[-1])
print(3, d['value'], d['roofline']['kernel_ms'], {k:v for k,v in d['agent'].items() if k!='what'})
"
