# configs[1] step / kernel under a knob (A/B): DEJAVU_NT = cache policy of the library rows
cd $GRAFT_REPO_ROOT
for nt in 0 1 0 1; do
  DEJAVU_NT=$nt python bench.py --views 50000 --sensor 64 --headings 16 --steps 400 --warmup 40 --event-every 4 --secondary 0 --batch-agents 0 --cpu-views 0 --agent-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NT=$nt step %.1f us scoring_only %.1f kernel %.1f us' % (d['ms_per_step']*1e3, d['scoring_only']['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
