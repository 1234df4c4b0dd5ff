"""CPU tests of the multi-rank host logic (no kernels): slab partitioning, the cross-rank region
resolve, and TorchComm's point-to-point / all-gather plumbing on the gloo backend with world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
from scipy import ndimage

from impact_amd.distributed import MAX_PAIRS, REC_WORDS, resolve_global_regions, slab_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_ranges_cover_and_balance():
    for n in (1, 5, 31, 32, 63, 255, 256):
        for w in (1, 2, 3, 4, 8):
            if n < w:
                continue
            r = slab_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def make_records(mask, world):
    """slab-local labelling with scipy, face pairs across slab boundaries -> the per-rank records"""
    nx = mask.shape[0]
    bounds = [(r * nx // world, (r + 1) * nx // world) for r in range(world)]
    labs, recs = [], np.zeros((world, REC_WORDS), dtype=np.int64)
    for r, (a, b) in enumerate(bounds):
        lab, n = ndimage.label(mask[a:b])
        labs.append(lab)
        recs[r, 0] = n
        recs[r, 18:28] = np.full(10, float(r + 1)).view(np.int64)
        recs[r, 2:14] = [r, r + 1, 0, 1, 0, 1, a, b, 0, 8, 0, 8]
        recs[r, 14:17] = [10 * (r + 1), 60 * (r + 1), r + 1]
    for r in range(world - 1):
        fa, fb = labs[r][-1], labs[r + 1][0]
        m = (fa > 0) & (fb > 0)
        pairs = np.unique(np.stack([fa[m] - 1, fb[m] - 1], axis=1), axis=0)
        recs[r, 1] = len(pairs)
        recs[r, 28:28 + 2 * len(pairs)] = pairs.reshape(-1)
    return labs, recs


@pytest.mark.parametrize("seed,world", [(0, 2), (1, 3), (2, 4), (3, 8)])
def test_resolve_global_regions_matches_whole_grid_labelling(seed, world):
    rng = np.random.default_rng(seed)
    mask = rng.random((32, 12, 12)) > 0.62
    labs, recs = make_records(mask, world)
    n, region_of, moments, occ, mesh = resolve_global_regions(recs)
    ref, n_ref = ndimage.label(mask)
    assert n == n_ref
    glob = np.zeros(mask.shape, dtype=np.int64)
    x = 0
    for r, lab in enumerate(labs):
        m = lab > 0
        out = np.zeros(lab.shape, dtype=np.int64)
        out[m] = region_of[r][lab[m] - 1] + 1
        glob[x:x + lab.shape[0]] = out
        x += lab.shape[0]
    # same partition: a bijection between the two labellings
    pairs = np.unique(np.stack([glob[mask], ref[mask]], axis=1), axis=0)
    assert len(pairs) == n and len(np.unique(pairs[:, 0])) == n and len(np.unique(pairs[:, 1])) == n
    np.testing.assert_allclose(moments, np.full(10, sum(range(1, world + 1))))
    assert occ[0] == 0 and occ[1] == world and occ[6] == 0 and occ[7] == 32
    assert mesh[world - 1] == (10 * world, 60 * world, world)


def test_resolve_rejects_overflowing_pair_lists():
    recs = np.zeros((2, REC_WORDS), dtype=np.int64)
    recs[0, 0] = recs[1, 0] = 1
    recs[0, 1] = MAX_PAIRS + 1
    with pytest.raises(Exception):
        resolve_global_regions(recs)


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["IVX_ROOT"])
from impact_amd.distributed import TorchComm, Exchange, REC_WORDS, resolve_global_regions

class HostBuf:  # stands in for DeviceBuffer on a machine without a GPU: TorchComm only touches `.t`
    def __init__(self, t): self.t = t

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
comm = TorchComm(dist, torch, rank, world)
assert not comm.on_device
# point-to-point halo exchange: every rank sends a buffer tagged with (rank, side)
n = 4096
lo = HostBuf(torch.full((n,), 10 * rank + 1, dtype=torch.uint8))
hi = HostBuf(torch.full((n,), 10 * rank + 2, dtype=torch.uint8))
rlo, rhi = HostBuf(torch.zeros(n, dtype=torch.uint8)), HostBuf(torch.zeros(n, dtype=torch.uint8))
class NoCuda:
    @staticmethod
    def synchronize(): pass
torch_cuda = torch.cuda
torch.cuda.synchronize = NoCuda.synchronize
# gloo path stages through .cpu(); host tensors make that a no-op copy
comm.exchange(Exchange(lo, hi, rlo, rhi))
if rank > 0:
    assert int(rlo.t[0]) == 10 * (rank - 1) + 2 and bool((rlo.t == rlo.t[0]).all())
    # a message may use only the head of the buffers (the first exchange of a step carries no face ids)
    rlo.t.zero_(); rhi.t.zero_()
    comm.exchange(Exchange(lo, hi, rlo, rhi, 1000))
    assert bool((rlo.t[:1000] == 10 * (rank - 1) + 2).all()) and int(rlo.t[1000:].sum()) == 0
    rlo.t.fill_(10 * (rank - 1) + 2)
else:
    comm.exchange(Exchange(lo, hi, rlo, rhi, 1000))
    assert int(rlo.t.sum()) == 0
if rank + 1 < world:
    assert int(rhi.t[0]) == 10 * (rank + 1) + 1 and bool((rhi.t == rhi.t[0]).all())
else:
    assert int(rhi.t.sum()) == 0
# all-gather of the per-rank record + identical resolve on every rank
rec = np.zeros(REC_WORDS, dtype=np.int64)
rec[0] = 2                      # two local components per rank
rec[18:28] = np.full(10, 1.5).view(np.int64)
if rank + 1 < world:
    rec[1] = 1
    rec[28:30] = [1, 0]         # my component 1 touches the next rank's component 0
records = comm.all_gather(rec)
assert records.shape == (world, REC_WORDS)
n_regions, region_of, moments, occ, mesh = resolve_global_regions(records)
assert n_regions == 2 * world - (world - 1)
assert np.allclose(moments, 1.5 * world)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_torchcomm_gloo_world_size_2(tmp_path):
    import subprocess

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IVX_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out
