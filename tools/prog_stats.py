#!/usr/bin/env python3
"""Developer tool: what the sampler's pre-pass left for the per-voxel evaluator on the 512^3 bench workload — per evaluated chunk the number of
leaf evaluations, combinations (applied unconditionally / behind the 14-position test) and folded constants of its compact program.
Also: how many combinations take a bare leaf as their second operand (the evaluator could combine it from registers, without an LDS level of
its own) and how many LDS levels the programs would need then.
usage: prog_stats.py [scale | dense]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402


def main():
    dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
    scale = float(sys.argv[1]) if len(sys.argv) > 1 and not dense else 2.05
    ctx = Context(0)
    gen = SDFVoxelGenerator(1.0, scenes.plates_scene(32) if dense else scenes.asteroid_scene(scale), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.step(capi.STAGE_ALL)
    lib = capi.lib()
    lib.ivx_grid_device_ptr.restype = C.c_void_p
    n = obj.n_chunks
    hip = C.CDLL("libamdhip64.so")
    lens = np.zeros(4 * n + 16, dtype=np.uint32)
    ops = np.zeros((n, 128, 2), dtype=np.uint32)
    hip.hipDeviceSynchronize()
    assert hip.hipMemcpy(lens.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 7)), lens.nbytes, 2) == 0
    assert hip.hipMemcpy(ops.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 8)), ops.nbytes, 2) == 0
    cnt = lens[n + 8:n + 16]  # (the counters as the derive sweep rolled them over: [0..3) lists, [3] long, [4] short entries of the first)
    counts = cnt[:3]
    print("evaluation lists (one level, two, more):", counts.tolist())
    names = {}
    tot = np.zeros(16, dtype=np.int64)
    per_chunk = []
    fus = np.zeros(4, dtype=np.int64)  # combinations: second operand a bare leaf (applied / behind the test), other
    need_now, need_fused = [], []
    skips = [0]
    all_shapes = {}  # live steps: C constant; s / q / b sphere, capsule, box leaf; S scale; U D I union, subtraction, intersection applied; u d i behind the test
    shapes = {}  # op strings (C constant, L leaf, S scale, A combination applied, O combination behind the test) of the programs that still need two levels
    for c in range(3):
        seg = lens[n + 16 + c * n:n + 16 + (c + 1) * n]
        lst = np.concatenate([seg[:cnt[3]], seg[n - cnt[4]:]]) if c == 0 else seg[:counts[c]]
        for ch in lst:
            ln = lens[ch]
            if ln > 128:
                continue
            opc_all = ops[ch, :ln, 0] >> 28
            live, i = [], 0
            while i < ln:  # (OP_SKIP = 5: the evaluator jumps over a dropped first operand's steps)
                if opc_all[i] == 5:
                    i += int(ops[ch, i, 1])
                    skips[0] += 1
                    continue
                live.append(int(opc_all[i]))
                i += 1
            opc = np.array(live, dtype=np.int64)
            kinds = (ops[ch, :ln, 0] >> 24) & 15
            sh, i = "", 0
            while i < ln:
                if opc_all[i] == 5:
                    i += int(ops[ch, i, 1])
                    continue
                o = int(opc_all[i])
                sh += {0: "C", 1: "sqb"[min(int(kinds[i]), 2)], 2: "S", 3: {7: "U", 8: "D", 9: "I"}.get(int(kinds[i]), "?"), 4: {7: "u", 8: "d", 9: "i"}.get(int(kinds[i]), "?")}[o]
                i += 1
            all_shapes[sh] = all_shapes.get(sh, 0) + 1
            h = np.bincount(opc, minlength=16)
            tot += h
            per_chunk.append(h)
            st, st_f = [], []  # (levels needed, constant, bare leaf)
            for o in opc:
                if o == 0:
                    st.append((0, True, False)); st_f.append((0, True, False))
                elif o == 1:
                    st.append((1, False, True)); st_f.append((1, False, True))
                elif o == 2:
                    a = st.pop(); st.append((a[0], a[1], False))
                    a = st_f.pop(); st_f.append((a[0], a[1], False))
                else:
                    b, a = st.pop(), st.pop()
                    st.append((max(a[0], (0 if a[1] else 1) + b[0], 1), False, False))
                    b, a = st_f.pop(), st_f.pop()
                    if b[2]:
                        fus[0 if o == 3 else 1] += 1
                        st_f.append((max(a[0], 1), False, False))
                    else:
                        fus[2] += 1
                        st_f.append((max(a[0], (0 if a[1] else 1) + b[0], 1), False, False))
            need_now.append(st[0][0] if st else 0)
            need_fused.append(st_f[0][0] if st_f else 0)
            if need_fused[-1] >= 2:
                shapes["".join("CLSAO"[min(int(o), 4)] for o in opc)] = shapes.get("".join("CLSAO"[min(int(o), 4)] for o in opc), 0) + 1
    per_chunk = np.array(per_chunk)
    print("chunks", len(per_chunk), "ops per chunk by opcode (mean):", {i: round(float(per_chunk[:, i].mean()), 2) for i in range(16) if tot[i]})
    print("combinations with a bare leaf as second operand: applied", int(fus[0]), "behind the test", int(fus[1]), "| other", int(fus[2]))
    print("LDS levels without the leaf + combination fusion:", np.bincount(need_now).tolist(), "with it (what the pre-pass counts):", np.bincount(need_fused).tolist())
    print("programs that would still need >= 2 levels:", sorted(shapes.items(), key=lambda kv: -kv[1])[:12])
    print("dropped first operands (OP_SKIP):", skips[0])
    print("most common programs:", sorted(all_shapes.items(), key=lambda kv: -kv[1])[:16])
    print("program length percentiles:", [int(np.percentile(per_chunk.sum(1), q)) for q in (10, 50, 90, 100)])


if __name__ == "__main__":
    main()
