#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats`
on ROCm 7.2) as the per-kernel CSV table rocprofv3 prints with `--output-format csv`.
usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"', file=out)
    for r in rows:
        print(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100 * r[2] / tot:.2f},{r[4]},{r[5]}', file=out)


if __name__ == "__main__":
    main()
