#!/usr/bin/env python3
"""Timing of the ssd_u8 metric (exact SSD of uint8 views on the int8 matrix cores): usage bench_ssd_u8.py [F h A]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import bench
F, h, A = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (50000, 64, 16)
print(json.dumps(bench.ssd_u8_block(0, F, h, h, A, 100)))
