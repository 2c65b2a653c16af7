# A/B of alternative builds of libdejavu_hip.so on the bench workload: bash tools/ab_lib.sh tools/libA.so tools/libB.so ...
ORIG=navigation-by-deja-vu_amd/csrc/libdejavu_hip.so
cp $ORIG /tmp/orig.so
for round in 1 2 3; do
  for lib in /tmp/orig.so "$@"; do
    cp $lib $ORIG
    r=$(timeout -k 10 120 python bench.py --steps 200 --warmup 20 --cpu-views 0 --agent-steps 0 --batch-agents 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value']))")
    echo "round $round $(basename $lib): $r"
  done
done
cp /tmp/orig.so $ORIG
