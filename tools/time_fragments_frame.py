#!/usr/bin/env python3
"""Developer tool: bench.py's `fragments_frame` leg by itself (81 fragments on a ground plane: the frame looped and batched, the probe sync
and the mutual contacts of neighbouring fragments looped and batched). usage: time_fragments_frame.py [--cpu]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import bench
from impact_amd.voxel import Context

ctx = Context(0)
out = bench.fragments_frame_benchmark(ctx, "--cpu" in sys.argv)
print(json.dumps(out, indent=1))
