# usage: bash tools/sweep_grid.sh  -- sweeps the scoring-grid knobs of libdejavu_hip.so on the bench workload
for cfg in "0 4600" "0 5400" "0 6200" "0 7000" "0 7800" "24 5400" "24 6200"; do
  set -- $cfg
  r=$(DEJAVU_WPC=$1 DEJAVU_TARGET_ITEMS=$2 timeout -k 10 120 python bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us frac %.3f step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['ms_per_step']*1e3, d['value']))")
  echo "WPC=$1 ITEMS=$2 : $r"
done
