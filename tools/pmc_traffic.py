#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass). Both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so it is doubled (same guide, section HBM). Output: JSON kernel -> bytes per launch
and bytes per STEP (per-launch average x launches per step; the profiled command runs `steps` steps in all, warm-up included,
and nothing but steps).
usage: pmc_traffic.py fetch_results.db write_results.db out.json <steps in the profiled run> [note]"""
import json
import re
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_sha import source_sha16  # noqa: E402


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,))
    out = {}
    for name, avg, n in rows:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
        short = re.sub(r"^void ", "", short)
        out[short] = (avg, n)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[4])
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0))
        w = write.get(k, (0.0, 0))
        n = max(f[1], w[1])
        res[k] = {"fetch_bytes": 2.0 * f[0] * 1024.0, "write_bytes": w[0] * 1024.0, "launches_sampled": n, "launches_per_step": n / steps}
        res[k]["hbm_bytes"] = res[k]["fetch_bytes"] + res[k]["write_bytes"]
        res[k]["hbm_bytes_per_step"] = res[k]["hbm_bytes"] * n / steps
    meta = {"_note": "bytes per launch and per step; fetch = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; separate --pmc passes. "
                     + (sys.argv[5] if len(sys.argv) > 5 else ""),
            "_step_total_bytes": sum(r["hbm_bytes_per_step"] for r in res.values()), "_source_sha16": source_sha16()}
    meta.update(res)
    json.dump(meta, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
