# usage: bash tools/sweep_grid.sh  -- sweeps experiment knobs of libdejavu_hip.so on the bench workload
for signed in 1 0; do for qm in 0 1 0 1 0 1; do
  r=$(DEJAVU_SIGNED=$signed DEJAVU_QMAJOR=$qm timeout -k 10 120 python bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value']))")
  echo "SIGNED=$signed QMAJOR=$qm : $r"
done; done
