#!/usr/bin/env python3
"""Developer tool: bench.py's edit leg by itself (the absorbing sphere into the headline body + its incremental remesh: the four calls in their
usual order, and with the sync enqueued while the edit is in flight). usage: time_edit_overlap.py"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from impact_amd.voxel import Context

ctx = Context(0)
print(json.dumps(bench.edit_benchmark(ctx, 2.05, None, reps=8), indent=1))
