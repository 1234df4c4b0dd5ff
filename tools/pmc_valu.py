#!/usr/bin/env python3
"""VALU issue counters per kernel from one rocprofv3 PMC pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES): per-launch
averages and per-step totals. SQ_INSTS_VALU counts wave-instructions; SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md).
usage: pmc_valu.py results.db out.json <steps in the profiled run>"""
import json
import re
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_sha import source_sha16  # noqa: E402


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[3])
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
    res = {}
    for name, counter, avg, n in rows:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
        short = re.sub(r"^void ", "", short)
        r = res.setdefault(short, {"launches_per_step": n / steps})
        r[counter] = avg
        r[counter + "_per_step"] = avg * n / steps
    out = {"_note": "per-launch averages and per-step totals; SQ_INSTS_VALU = wave-instructions, SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES as rocprofv3 reports them"}
    out["_source_sha16"] = source_sha16()
    out.update(res)
    json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
