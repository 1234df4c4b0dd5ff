"""What the sampler's interval pre-pass leaves the per-voxel evaluator (a property of the pruning, not of the bytes — those are held to the
oracle everywhere else): on the bench's two bodies the compact per-chunk programs must stay as short as round 6 made them — an operand that
cannot reach a chunk's bytes is dropped whether it is a constant or not, on either side of its combination, by distance (identity / mirror)
or by saturation class. A regression here costs the evaluator its speed without failing a single parity test."""
import ctypes as C

import numpy as np
import pytest

from impact_amd import capi, scenes
from impact_amd.voxel import SDFVoxelGenerator, VoxelObject

pytestmark = pytest.mark.gpu

OP_CONST, OP_LEAF, OP_SCALE, OP_COMBINE, OP_COMBINE_OUTSIDE, OP_SKIP = 0, 1, 2, 3, 4, 5
OP_OVERFLOW = 0xFFFFFFFF


def live_programs(ctx, graph):
    """(op-code histogram per evaluated chunk, number of chunks on the full program) after one sample stage"""
    gen = SDFVoxelGenerator(1.0, graph, 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.step(capi.STAGE_ALL)  # (the derive sweep rolls the list counters over: their statistics copy is what is read below)
    lib = capi.lib()
    lib.ivx_grid_device_ptr.restype = C.c_void_p
    hip = C.CDLL("libamdhip64.so")
    n = obj.n_chunks
    lens = np.zeros(4 * n + 16, dtype=np.uint32)
    ops = np.zeros((n, 128, 2), dtype=np.uint32)
    hip.hipDeviceSynchronize()
    assert hip.hipMemcpy(lens.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 7)), lens.nbytes, 2) == 0
    assert hip.hipMemcpy(ops.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 8)), ops.nbytes, 2) == 0
    cnt = lens[n + 8:n + 16]  # rolled counters: [0..3) the three lists, [3] long and [4] short entries of the first
    hist, full = [], 0
    for c in range(3):
        seg = lens[n + 16 + c * n:n + 16 + (c + 1) * n]
        lst = np.concatenate([seg[:cnt[3]], seg[n - cnt[4]:]]) if c == 0 else seg[:cnt[c]]
        for ch in lst:
            ln = lens[ch]
            if ln == OP_OVERFLOW:
                full += 1
                continue
            h, i = np.zeros(6, dtype=np.int64), 0
            while i < ln:
                o = int(ops[ch, i, 0] >> 28)
                if o == OP_SKIP:  # (a dropped first operand's steps: the evaluator jumps over them)
                    i += int(ops[ch, i, 1])
                    continue
                h[o] += 1
                i += 1
            hist.append(h)
    evaluated = obj.stage_counters()["evaluated_chunks"]
    obj.close()
    return np.array(hist), full, (int(cnt[0]), int(cnt[1]), int(cnt[2])), evaluated


def test_asteroid_programs_are_short(ctx):
    """config 2's asteroid (core + six bumps under smooth unions, minus eight craters, 33 nodes): a surface chunk sees the core or a bump, often
    both, rarely a crater — 4.1 leaves + 4.1 combinations per chunk before round 6"""
    hist, full, lists, evaluated = live_programs(ctx, scenes.asteroid_scene(1.0))
    assert len(hist) + full == evaluated == sum(lists) and evaluated > 500
    assert full == 0  # (every combination of this body is applied: nothing falls back to the full program)
    mean = hist.mean(axis=0)
    assert mean[OP_LEAF] <= 2.3, mean  # (2.10 at this scale; the smoothing distances are wider against the chunk than on the x2.05 body: 1.77)
    assert mean[OP_COMBINE] + mean[OP_COMBINE_OUTSIDE] <= 1.3, mean
    assert mean[OP_CONST] <= 0.2, mean
    assert lists[1] == 0 and lists[2] == 0  # (every program runs in the one-level class: eight workgroups per CU)


def test_plates_programs_are_short(ctx):
    """the all-surface body (perforated plates one per chunk layer): a chunk keeps its own plate, the neighbour plate that comes within 2.54
    voxels of it, and the folded far holes — `bbUCD` where it was `CbUbUbUCD`"""
    hist, full, lists, evaluated = live_programs(ctx, scenes.plates_scene(8))
    assert len(hist) + full == evaluated == sum(lists) and evaluated == 8 ** 3
    assert full == 0
    mean = hist.mean(axis=0)
    assert mean[OP_LEAF] <= 2.3, mean
    assert mean[OP_COMBINE] + mean[OP_COMBINE_OUTSIDE] <= 2.3, mean
    assert lists[1] == 0 and lists[2] == 0
