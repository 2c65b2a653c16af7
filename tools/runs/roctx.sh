cd /tmp && export TMPDIR=/tmp
DEJAVU_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/roctx -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dist --steps 30 --warmup 5 --views 50000 --sensor 64 --headings 16 --cpu-views 0 > $GRAFT_REPO_ROOT/gpurun_out/suite/roctx_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/suite/roctx.err
echo roctx rc=$?
ls $GRAFT_REPO_ROOT/gpurun_out/roctx/*/; cat $GRAFT_REPO_ROOT/gpurun_out/roctx/*/*domain_stats.csv; head -5 $GRAFT_REPO_ROOT/gpurun_out/roctx/*/*marker*stats*.csv 2>/dev/null
python3 -c "
import json
d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/suite/roctx_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['exchange'])
"
