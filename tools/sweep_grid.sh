# usage: bash tools/sweep_grid.sh  -- A/B of the k_sad_tiles workgroup shapes (DEJAVU_SHAPE 1..5, see launch_tiles_apad
# in csrc/dejavu_hip.hip; 0 = timed once per library, the default) over library sizes and heading counts
for views in 20000 50000 100000 200000; do for A in 8 16 32 64; do for shape in 1 2 3 4 5 0; do
  if [ $shape -ge 3 ] && [ $shape -le 4 ] && [ $A -eq 8 ]; then continue; fi
  if [ $shape -eq 4 ] && [ $A -ne 64 ]; then continue; fi
  r=$(DEJAVU_SHAPE=$shape timeout -k 10 120 python bench.py --views $views --headings $A --steps 100 --warmup 10 --cpu-views 0 --agent-steps 0 --batch-agents 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value']))")
  echo "views=$views A=$A shape=$shape : $r"
done; done; done
