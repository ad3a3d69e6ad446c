"""Multi-GPU self-check (runs with -m gpu; SKIPPED unless at least two GPUs are visible): bench.py launched as the
driver launches it for N>1 - one process per GPU, RCCL slot all-gather every step inside the library - on a small
collision-heavy case, asserting the correctness bit it computes itself: the state the ranks hold (collective
nbody_download over RCCL) equals a single-rank run of the same steps bit for bit.  On the 1-GPU box the same code path
is rehearsed with one rank by tests/test_gpu_parity.py::test_bench_distributed_control_path_one_rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()      # counting devices does not initialise the GPU on this image
    except Exception:
        return 0


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_two_ranks_self_check(ranks):
    if _gpu_count() < ranks:
        pytest.skip("needs %d visible GPUs (one rank per GPU over RCCL)" % ranks)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus",
           str(ranks), "--steps", "3", "--warmup", "1", "--bodies", "16384", "--stock-radii", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == ranks and d["config"]["rccl_ranks"] == ranks
    assert d["parity"]["bitwise_equal"] is True, d["parity"]
    assert d["parity"]["bodies_after"] < 16384            # deletions happened: ragged slots, re-drawn partition
