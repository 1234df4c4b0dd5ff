"""The multi-GPU bench line on whatever the box has: `bench.py --gpus N` under torch.distributed.run with N = min(device count, 8), one rank per
GPU over RCCL (the library's own communicator: ivx_comm_init). Skipped on a one-GPU box — there the native protocol runs as separate processes
over the shared-device transport instead (tests/test_gpu_slabs_ipc.py). The first multi-GPU box therefore exercises the RCCL path — unique-id
hand-over, grouped send/recv of the face planes, the record all-gather, the doorbell — before anyone times it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_bench_line_on_every_gpu_of_the_box(overlap):
    """Both forms of the neighbour exchange (IVX_SLAB_OVERLAP: 0 = on the context's stream, the default; 1 = on the communicator's own stream
    beside the interior work) must give the same triangles and regions: the first box with two GPUs tests both."""
    import torch

    n = min(torch.cuda.device_count(), 8)  # (counting devices does not initialise the GPU)
    if n < 2:
        pytest.skip("one GPU: RCCL refuses two ranks on one device (the multi-process protocol test is tests/test_gpu_slabs_ipc.py)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", IVX_SLAB_OVERLAP=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-pile"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 5
    assert d["ranks"]["world_size"] == n and d["ranks"]["communicator_ranks"] == n and d["ranks"]["backend"] == "nccl"
    legs = d["scaling_legs"]
    assert set(legs) == {"strong", "weak"}
    # the decomposition does not change the mesh or the regions: the strong leg is the 512^3 headline body whatever N is
    assert legs["strong"]["regions"] == 1 and legs["weak"]["regions"] == 1
    assert legs["strong"]["triangles"] == 1813104  # (the 512^3 asteroid's triangle count: tests/test_gpu_parity.py::test_bench_workload_512 pins it to the oracle)
    assert legs["weak"]["triangles"] > 0 and legs["strong"]["ms_per_step"] > 0 and legs["weak"]["ms_per_step"] > 0
    assert d["value"] > 0 and d["unit"].startswith("voxels")
