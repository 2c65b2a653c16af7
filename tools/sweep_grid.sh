# usage: bash tools/sweep_grid.sh  -- A/B of the k_sad_tiles workgroup shape (DEJAVU_WPB: 1 or 4 waves per workgroup,
# 0 = the cost model in pick_waves_per_block) over library sizes and heading counts, on the bench workload
for views in 20000 50000 100000 200000; do for A in 8 16 32; do for wpb in 1 4 0; do
  r=$(DEJAVU_WPB=$wpb timeout -k 10 120 python bench.py --views $views --headings $A --steps 100 --warmup 10 --cpu-views 0 --agent-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kern %.1f us step %.1f us value %.3g' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value']))")
  echo "views=$views A=$A WPB=$wpb : $r"
done; done; done
