#!/usr/bin/env python3
"""The ensemble share of BASELINE.json configs[4] alone (32 agents x 16 headings, 100 000 views of 64x64, patches sensed on the
device): what tools/profile_bench.sh profiles as r03_ens.  Prints bench.py's `ensemble` block."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import bench
print(json.dumps(bench.ensemble_comparisons_per_s(64, 64, 16, 0.25, 20261004, 32, 100000, 10)))
