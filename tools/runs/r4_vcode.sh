# configs[1] with and without the 3-bit code rows of the value segment (DEJAVU_VCODE): kernel time, interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 0 1; do
    DEJAVU_VCODE=$v python bench.py --steps 200 --warmup 20 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('vcode=$v kernel_us %.2f step_us %.2f bytes %.1f MB' % (r['kernel_ms']*1e3, d['ms_per_step']*1e3, r.get('streamed_library_bytes_per_launch',0)/1e6))"
  done
done
