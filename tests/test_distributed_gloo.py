"""The N > 1 path on CPU: world_size-2 (and 4) gloo groups exchange the per-rank records exactly as the GPU ranks do over
RCCL and every rank reconstructs the unsharded u*.  See tests/dist_worker.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_exchange_reconstructs_unsharded_result(world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(world):
        assert "rank %d ok" % k in r.stdout


def test_shard_bounds_cover_everything_once():
    from mppi_generic_amd import distributed as md
    for K, G in ((16384, 8), (65536, 4), (8192, 1)):
        seen = np.zeros(K, int)
        for r in range(G):
            off, n = md.shard_bounds(K, r, G)
            seen[off:off + n] += 1
        assert (seen == 1).all()
    assert md.record_floats(100, 1) == 104 and md.record_floats(150, 2, 2) == 608
