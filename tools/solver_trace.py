#!/usr/bin/env python3
"""Read a chain-stationary solve's round trace (developer tool): run with IVX_SOLVER_TRACE=<file> in the environment, e.g.
    IVX_SOLVER_TRACE=gpurun_out/trace.bin python3 tools/time_pile.py 16
then `python3 tools/solver_trace.py gpurun_out/trace.bin`. Per phase: the time a level takes, and how a round's time splits into
nap + poll (round begins -> operands there), the chain's arithmetic, and the stores."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n0, n1 = int(raw[0]), int(raw[1])
stamps = raw[2:2 + 4 * (n0 + n1)].astype(np.int64).reshape(-1, 4)
levels = np.frombuffer(raw[2 + 4 * (n0 + n1):].tobytes(), dtype=np.uint32)[: n0 + n1]
for name, lo, hi in (("velocity", 0, n0), ("positional", n0, n0 + n1)):
    if hi == lo:
        continue
    st, lv = stamps[lo:hi], levels[lo:hi]
    ok = (st > 0).all(axis=1)
    st, lv = st[ok], lv[ok]
    t0 = st[:, 0].min()
    us = (st - t0) * 0.01
    print(f"{name}: {len(st)} rounds, levels {lv.min()}..{lv.max()}, span {us[:, 3].max():.1f} us = {us[:, 3].max() / lv.max():.2f} us per level")
    print("  mean us: wait (nap + poll) %.2f | arithmetic %.2f | stores issued %.2f" % tuple((us[:, i + 1] - us[:, i]).mean() for i in range(3)))
    # per level: when its first round's operands were there, when its last round was done
    L = int(lv.max())
    first_ready = np.full(L + 1, np.inf)
    last_done = np.zeros(L + 1)
    np.minimum.at(first_ready, lv, us[:, 1])
    np.maximum.at(last_done, lv, us[:, 3])
    np.set_printoptions(precision=1, suppress=True, linewidth=200)
    ks = [k for k in (1, 10, 20, 40, 80, 120, 140, 160, 180) if k <= L]
    print("  level: first operands there / last round done (us):", [(k, round(float(first_ready[k]), 1), round(float(last_done[k]), 1)) for k in ks])
    d = np.diff(last_done[1:])
    print(f"  level-to-level (last done): median {np.median(d):.2f} us, p10 {np.percentile(d, 10):.2f}, p90 {np.percentile(d, 90):.2f}")
    arith = us[:, 2] - us[:, 1]
    print(f"  arithmetic per round: median {np.median(arith):.2f}, p10 {np.percentile(arith, 10):.2f}, p90 {np.percentile(arith, 90):.2f} us")
    # how long after a level's last producer finished did the next level's first consumer see its operands?
    gap = first_ready[2:] - last_done[1:-1]
    print(f"  next level's first operands after this level's last store: median {np.median(gap):.2f} us (negative: levels overlap)")
