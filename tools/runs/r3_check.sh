# round 3: suite, bench, kernel trace of the headline and configs[1] with fresh patches per step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r3/pytest.log
[ $rc -ne 0 ] && exit $rc
DEJAVU_VERBOSE=1 timeout -k 10 600 python bench.py > gpurun_out/r3/full.json 2> gpurun_out/r3/full.err
echo bench rc=$?
tail -3 gpurun_out/r3/full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/full.json').read().strip().splitlines()[-1])
keep={k:d[k] for k in ('value','ms_per_step','scoring_only') if k in d}
keep['kernel_ms']=d['roofline']['kernel_ms']; keep['frac']=d['roofline']['frac']
for k in ('configs1','agent','ssd_f32','ensemble'):
    v=d.get(k,{})
    keep[k]={kk:v.get(kk) for kk in ('value','ms_per_step','scoring_only','nav_steps_per_s','nav_steps_per_s_fake','view_comparisons_per_s','ms_per_ensemble_step','error')}
    if 'roofline' in v: keep[k]['kernel_ms']=v['roofline']['kernel_ms']; keep[k]['frac']=v['roofline']['frac']
print(json.dumps(keep,indent=1))
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c2:--steps 30 --warmup 5" "c1:--views 50000 --sensor 64 --headings 16 --steps 200 --warmup 20"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rm -rf $R/gpurun_out/r3/trace_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/trace_$tag -- python3 $R/bench.py $args --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 > $R/gpurun_out/r3/trace_$tag.json 2> $R/gpurun_out/r3/trace_$tag.err
  echo trace $tag rc=$?
  f=$(ls -S $R/gpurun_out/r3/trace_$tag/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"])>=10: print("%-70s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
