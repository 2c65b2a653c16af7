# After the group / pipelined-agent / chain-order changes: the whole GPU suite, bench.py under torch.distributed.run at world size 1
# (RCCL exchange on the step's stream), bench.py with its defaults.
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -q -x -m gpu > gpurun_out/r4_suite2.log 2>&1; rc=$?; tail -3 gpurun_out/r4_suite2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --secondary 0 --cpu-views 0 --batch-agents 0 --agent-steps 0 > gpurun_out/r4_dist1.json 2> gpurun_out/r4_dist1.err; echo dist rc=$?; python -c "
import json; d=json.loads(open('gpurun_out/r4_dist1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('rccl_ranks'), d['config'].get('exchange_us_per_step'))"
python bench.py > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err; echo bench rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r4_bench2.json')); s=d['roofline']['secondary']
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('c1_step_us','c1_kernel_us','ens_ms','ens_uploaded_ms','ens_mfma_frac','agent_steps_per_s','agent_steps_per_s_fake','agent_ensemble_of_32_steps_per_s','generic_hue_valu_frac','ssd_f32_c2_frac','ssd_u8_frac'): print(k, s.get(k))"
