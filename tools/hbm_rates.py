#!/usr/bin/env python3
"""Developer tool: what the box's HBM does for a pure write, a copy and a pure read (the practical ceilings beside the 8 TB/s figure):
torch's fill / copy / sum over 2 GiB, HIP events, best of 10."""
import torch

n = 1 << 29  # 2 GiB of f32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")


def best(fn, reps=10):
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return min(t)


gib = n * 4 / 1e9
tw = best(lambda: a.fill_(1.0))
tc = best(lambda: b.copy_(a))
tr = best(lambda: a.sum())
i32 = a.view(torch.int32)
tw2 = best(lambda: i32.zero_())
print(f"write (fill)   : {gib / tw * 1e3:.0f} GB/s")
print(f"write (zero)   : {gib / tw2 * 1e3:.0f} GB/s")
print(f"copy (r + w)   : {2 * gib / tc * 1e3:.0f} GB/s of traffic ({gib / tc * 1e3:.0f} GB/s copied)")
print(f"read (sum)     : {gib / tr * 1e3:.0f} GB/s")
