#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the kernel sources of libimpact_voxel_hip.so: what a committed PMC summary was measured on.
bench.py compares it with the tree it runs from before it quotes counter traffic (a summary of other kernels is not quoted)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "impact_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "impact_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "impact_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "impact_voxel_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
