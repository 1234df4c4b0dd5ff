#!/usr/bin/env python3
"""From a kernel-stats table of the step-only bench command (tools/rocpd_stats.py): the evaluator's share of the sample slot's time, stamped
with the hash of the kernel sources — what bench.py's `roofline.dominant_kernel_frac` divides by.
usage: kernel_share.py <kernel_stats.csv> <out.json>"""
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from source_sha import source_sha16  # noqa: E402

tot = {}
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    for k in ("k_sdf_eval", "k_sdf_prepass", "k_sdf_super"):
        if k + "<" in n or "::" + k + "(" in n:
            tot[k] = tot.get(k, 0.0) + float(row["TotalDurationNs"])
slot = sum(tot.values())
out = {"_source_sha16": source_sha16(), "from": sys.argv[1].split("/")[-1], "ns": tot,
       "k_sdf_eval_share_of_sdf_sample": (tot.get("k_sdf_eval", 0.0) / slot) if slot else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
