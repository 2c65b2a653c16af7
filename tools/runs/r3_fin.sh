cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
timeout -k 5 500 python -m pytest tests -m gpu -q -x > gpurun_out/fin/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/fin/pytest.log
[ $rc -ne 0 ] && exit $rc
echo "=== stamps C1"; timeout -k 5 60 python tools/exp/stamps.py run 50000 64 16 2>/dev/null | grep "phase\|exit"
echo "=== stamps 100k x 64 headings"; timeout -k 5 60 python tools/exp/stamps.py run 100000 64 64 2>/dev/null | grep "phase\|exit"
timeout -k 10 300 python bench.py --views 20000 --sensor 64 --headings 16 --steps 20 --warmup 5 --cpu-views 0 --secondary 0 --agent-steps 0 --batch-agents 32 > gpurun_out/fin/ens.json 2> gpurun_out/fin/ens.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/fin/ens.json').read().strip().splitlines()[-1])
print(json.dumps(d.get("ensemble"), indent=1))
PY
source tools/runs/r3_ab_fn.sh
run c1 $C1
run c2 $C2
