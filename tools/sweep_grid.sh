# usage: bash tools/sweep_grid.sh  -- sweeps the scoring-grid knobs of libdejavu_hip.so on the bench workload
for cfg in "24 5120 1" "28 5120 1" "28 5120 2" "28 7000 1" "28 7000 2" "28 6000 2" "24 5120 2" "28 6500 2" "28 4600 2" "28 3900 2" "28 8500 2" "28 10000 2"; do
  set -- $cfg
  r=$(DEJAVU_WPC=$1 DEJAVU_TARGET_ITEMS=$2 DEJAVU_PF=$3 timeout -k 10 120 python bench.py --steps 100 --warmup 10 --cpu-views 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us frac %.3f step %.1f us' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['ms_per_step']*1e3))")
  echo "WPC=$1 ITEMS=$2 PF=$3 : $r"
done
