# A/B of two builds of the library (tools/exp/ab/{old,new}.so, not tracked) on configs[1]: scoring kernel, step, agent rate; interleaved.
cd $GRAFT_REPO_ROOT
L=navigation-by-deja-vu_amd/csrc/libdejavu_hip.so
cp $L /tmp/keep.so
for rep in 1 2 3; do
  for v in old new; do
    cp tools/exp/ab/$v.so $L
    python bench.py --steps 300 --warmup 30 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('agent',{}); print('$v kernel_us %.2f step_us %.2f agent %.0f fake %.0f' % (r['kernel_ms']*1e3, d['ms_per_step']*1e3, a.get('nav_steps_per_s') or 0, a.get('nav_steps_per_s_fake') or 0))"
  done
done
cp /tmp/keep.so $L
timeout -k 10 600 python -m pytest tests -q -x -m gpu -k "matrix_core or fp4 or parity or ragged or knobs or tie or golden" > gpurun_out/ab_c1_tests.log 2>&1; tail -2 gpurun_out/ab_c1_tests.log
