cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --views 50000 --sensor 64 --headings 16 --cpu-views 0 --secondary 0 --batch-agents 0 --agent-steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d.get('agent',{}), indent=1)[:3000])"
