# `python bench.py --gpus 2` without a launcher: bench.py starts its own ranks.  On a one-GPU box both ranks share GPU 0
# (gloo for the collectives; the RCCL exchange needs one GPU per rank) -- a rehearsal of the launch path, not a scaling number.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/selflaunch
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --views 20000 --sensor 64 --headings 16 --steps 20 --warmup 3 --cpu-views 0 > gpurun_out/selflaunch/out.json 2> gpurun_out/selflaunch/err.log
echo rc=$?
tail -3 gpurun_out/selflaunch/err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/selflaunch/out.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ("n_gpus","value","ms_per_step","known_answer_step")}, d["config"]["exchange"])
PY
