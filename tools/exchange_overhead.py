#!/usr/bin/env python3
"""Per-iteration cost of the multi-GPU exchange path measured on ONE GPU: force_exchange routes every iteration through
local merge -> RCCL all-gather (world of one rank) -> global merge, i.e. everything but the inter-GPU hop."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import mppi_generic_amd as m  # noqa: E402
from common import cartpole_cfg, make_engine  # noqa: E402

cfg = cartpole_cfg(K=16384, T=100)
plain = make_engine(cfg)
plain.uploadState(cfg["x0"])
plain.optimize(50)
t, r = plain.timeIterations(300)
print("plain:            iteration %.1f us (rollout kernel %.1f us)" % (t / 300 * 1e3, r / 300 * 1e3))
lib = m.load_library()
buf = C.create_string_buffer(128)
nb = C.c_size_t()
assert lib.mppi_rccl_unique_id(buf, 128, C.byref(nb)) == 0
eng = make_engine(cfg, force_exchange=True)
eng.commInitRccl(bytes(buf.raw))
eng.uploadState(cfg["x0"])
eng.optimize(50)
t, r = eng.timeIterations(300)
print("exchange (1 rank): iteration %.1f us (rollout kernel %.1f us)" % (t / 300 * 1e3, r / 300 * 1e3))
# the P2P mailbox path with a world of one rank (the rank posts into its own mailbox): rollout, local merge + post, global merge
p2p = make_engine(cfg, force_exchange=True)
m.MPPIController.p2pConnectLocal([p2p])
p2p.uploadState(cfg["x0"])
p2p.optimize(50)
t, r = p2p.timeIterations(300)
print("p2p mailbox (1 rank): iteration %.1f us (rollout kernel %.1f us)" % (t / 300 * 1e3, r / 300 * 1e3))
# two ranks of a K = 16384 problem on ONE GPU (8192 rollouts each, both rollout kernels share the device): the strong-scaling
# shape of the exchange, minus the xGMI hop
import time  # noqa: E402
half = [make_engine(cfg, rank=r_, world_size=2) for r_ in range(2)]
plain.close(); eng.close(); p2p.close()
m.MPPIController.p2pConnectLocal(half)
for c in half:
    c.uploadState(cfg["x0"])
for c in half:
    c.optimize(50, synchronize=False)
for c in half:
    c.synchronize()
t0 = time.perf_counter()
for c in half:
    c.optimize(1000, synchronize=False)
for c in half:
    c.synchronize()
print("p2p mailbox, 2 ranks x 8192 rollouts on one GPU: %.1f us per iteration" % ((time.perf_counter() - t0) / 1000 * 1e6))
