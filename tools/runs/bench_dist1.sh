cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 30 --warmup 5 --cpu-views 0 --agent-steps 0 --batch-agents 0 --secondary 0 > gpurun_out/bench/dist1.json 2> gpurun_out/bench/dist1.err
echo rc=$?
tail -3 gpurun_out/bench/dist1.err
python -c "
import json
d=json.loads(open('gpurun_out/bench/dist1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'], d['config'].get('exchange'), d['known_answer_step'])
"
