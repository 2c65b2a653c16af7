# usage: bash tools/sweep_grid.sh  -- sweeps experiment knobs of libdejavu_hip.so on the bench workload
for it in 0 3900 4600 5400 6200 7000; do
  r=$(DEJAVU_TARGET_ITEMS=$it timeout -k 10 120 python bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us frac %.3f step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['ms_per_step']*1e3, d['value']))")
  echo "ITEMS=$it : $r"
done
