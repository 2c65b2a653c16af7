# usage: bash tools/sweep_grid.sh  -- sweeps experiment knobs of libdejavu_hip.so on the bench workload
for st in 0 1 2 3 4 6 8 0; do
  r=$(DEJAVU_STAGGER=$st timeout -k 10 120 python bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us frac %.3f step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['ms_per_step']*1e3, d['value']))")
  echo "STAGGER=$st : $r"
done
