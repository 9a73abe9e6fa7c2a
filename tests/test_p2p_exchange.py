"""P2P mailbox exchange (mppi_p2p_*; SURVEY.md §8e second stage): every rank writes its merged record straight into the peers'
mailboxes and the merge kernels spin on their own flags — no collective library.  On one GPU the ranks are several handles
of one process (mppi_p2p_connect_local) or several processes sharing the device through hipIpc (mppi_p2p_connect)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

import mppi_generic_amd as m
from common import cartpole_cfg, di_cfg, make_engine, make_oracle
import pyoracle as po

pytestmark = pytest.mark.gpu


def _run_ranks(cfg, world, iters, x0):
    import gc
    gc.collect()  # engines of earlier tests that were only dropped, not closed, still hold streams
    ranks = [make_engine(cfg, rank=r, world_size=world) for r in range(world)]
    m.MPPIController.p2pConnectLocal(ranks)
    for c in ranks:
        c.uploadState(x0)
    for c in ranks:  # all ranks' kernels are enqueued before anybody waits: each merge kernel spins for the others' posts
        c.optimize(iters, synchronize=False)
    for c in ranks:
        c.synchronize()
    return ranks


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_local_matches_unsharded(gpu, world):
    cfg = cartpole_cfg(K=4096, T=100, soft=True)
    full = make_engine(cfg)
    full.uploadState(cfg["x0"])
    full.optimize(3)
    u_full = full.getOptimalControlSeq()[0]
    rho_full = full.getStats().real_sys.baseline
    # ranks living in ONE process each need a hardware queue of their own (a merge kernel spins until the other ranks'
    # kernels have run; HIP multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues): no other stream stays alive here
    full.close()
    ranks = _run_ranks(cfg, world, 3, cfg["x0"])
    us = [c.getOptimalControlSeq()[0] for c in ranks]
    for u in us:
        assert np.abs(u - u_full).max() <= 5e-6, np.abs(u - u_full).max()
        assert np.array_equal(u, us[0])  # every rank merges the same records in the same order: same bits
    st = [c.getStats().real_sys for c in ranks]
    assert all(s.baseline == rho_full for s in st)
    for c in ranks:
        c.close()
    # and against the oracle's un-sharded single iteration on the same Philox stream
    ranks = _run_ranks(cfg, world, 1, cfg["x0"])
    orc = make_oracle(cfg)
    u_orc = orc.iterate(cfg["x0"], np.zeros((cfg["T"], 1), np.float32), po.philox_normal(42, 0, cfg["K"], cfg["T"], 1))[0]
    u0 = ranks[0].getOptimalControlSeq()[0]
    for c in ranks:  # leave no stream behind: the next test's ranks need the hardware queues
        c.close()
    assert np.abs(u0 - u_orc).max() <= 1e-5


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_tsallis_weights_on_sharded_handles(gpu, world):
    """ColoredMPPI's Tsallis weights (core/mppi_common.cu:968-985) need the GLOBAL baseline before any weight: on K-sharded
    handles the ranks exchange their minima first, then {sum w v | rho, sum w, sum w^2} under the common baseline (two mailbox
    exchanges per iteration, engine_iteration.hip: iterationShardedTsallis) — against the un-sharded engine on the same noise and
    against the oracle's un-sharded iteration"""
    from common import host_spectrum
    from test_colored_noise import _colored_cartpole
    import gc
    cfg = _colored_cartpole(K=2048, T=60)
    K, T = cfg["K"], cfg["T"]
    exps, decay, fmin = cfg["colored"]
    kw = dict(gamma=400.0, r_exp=1.7)
    z = host_spectrum(2, K, T, 1, seed=40)
    full = make_engine(cfg)
    full.setColoredMPPIParams(**kw)
    full.injectNoise(z)
    full.uploadState(cfg["x0"])
    full.optimize(2)
    u_full = full.getOptimalControlSeq()[0].copy()
    st_full = full.getStats().real_sys
    rho_full, eta_full = st_full.baseline, st_full.normalizer
    full.close()
    gc.collect()
    ranks = [make_engine(cfg, rank=r, world_size=world) for r in range(world)]
    m.MPPIController.p2pConnectLocal(ranks)
    kl = K // world
    for r, c in enumerate(ranks):
        c.setColoredMPPIParams(**kw)
        c.injectNoise(np.ascontiguousarray(z[:, r * kl:(r + 1) * kl]))
        c.uploadState(cfg["x0"])
    for c in ranks:
        c.optimize(2, synchronize=False)
    for c in ranks:
        c.synchronize()
    us = [c.getOptimalControlSeq()[0] for c in ranks]
    sts = [c.getStats().real_sys for c in ranks]
    for c in ranks:
        c.close()
    for u, st in zip(us, sts):
        assert np.isfinite(u).all()
        assert np.array_equal(u, us[0])  # every rank merges the same records in the same order
        assert np.abs(u - u_full).max() <= 5e-6, np.abs(u - u_full).max()
        assert st.baseline == rho_full
        assert abs(st.normalizer - eta_full) <= 1e-6 * eta_full
    # the oracle's un-sharded first iteration (time-domain noise from its own colored-noise pipeline)
    orc = make_oracle(cfg)
    orc.set_colored_mppi_params(kw["gamma"], kw["r_exp"], None, False, 1)
    eps = po.colored_noise(z[0], exps, decay, fmin)
    u_orc = orc.iterate(cfg["x0"], np.zeros((T, 1), np.float32), eps)[0]
    ranks = [make_engine(cfg, rank=r, world_size=world) for r in range(world)]
    m.MPPIController.p2pConnectLocal(ranks)
    for r, c in enumerate(ranks):
        c.setColoredMPPIParams(**kw)
        c.injectNoise(np.ascontiguousarray(z[:1, r * kl:(r + 1) * kl]))
        c.uploadState(cfg["x0"])
    for c in ranks:
        c.optimize(1, synchronize=False)
    for c in ranks:
        c.synchronize()
    u1 = ranks[0].getOptimalControlSeq()[0]
    for c in ranks:
        c.close()
    assert np.abs(u1 - u_orc).max() <= 1e-5, np.abs(u1 - u_orc).max()


def test_p2p_local_two_systems(gpu):
    """Tube-MPPI (two systems per record): the gathered records are [world][D][PS], merged with world-major strides"""
    cfg = di_cfg(K=2048, T=60, tube=True)
    x0 = np.tile(cfg["x0"], (2, 1))
    full = make_engine(cfg)
    full.uploadState(x0)
    full.optimize(2)
    u_full = full.getOptimalControlSeq()
    full.close()
    ranks = _run_ranks(cfg, 2, 2, x0)
    us = [c.getOptimalControlSeq() for c in ranks]
    for c in ranks:
        c.close()
    for u in us:
        assert np.abs(u - u_full).max() <= 5e-6


def test_p2p_missing_peer_times_out_instead_of_hanging(gpu):
    """a rank whose peer never posts: the merge kernel gives up after 2 s and the host is told (MPPI_ERR_COMM)"""
    cfg = cartpole_cfg(K=1024, T=20, soft=True)
    ranks = [make_engine(cfg, rank=r, world_size=2) for r in range(2)]
    m.MPPIController.p2pConnectLocal(ranks)
    ranks[0].uploadState(cfg["x0"])
    ranks[0].optimize(1, synchronize=True)  # rank 1 never runs
    with pytest.raises(m.MPPIError) as e:
        ranks[0].getStats()
    for c in ranks:
        c.close()
    assert e.value.status == 9


def _ipc_worker(rank, world, conn, repo):
    for p in (repo, os.path.join(repo, "oracle"), os.path.join(repo, "tests")):
        sys.path.insert(0, p)
    import numpy as np  # noqa: F811
    from common import cartpole_cfg, make_engine  # noqa: F811
    cfg = cartpole_cfg(K=4096, T=100, soft=True)
    eng = make_engine(cfg, rank=rank, world_size=world)
    conn.send(eng.p2pMailboxHandle())
    handles = conn.recv()
    eng.p2pConnect(handles)
    conn.send("connected")
    conn.recv()  # everybody is connected: go
    eng.uploadState(cfg["x0"])
    eng.optimize(3, synchronize=True)
    conn.send(eng.getOptimalControlSeq()[0].tobytes())
    conn.recv()
    # ---- a second session on the same handles (round-4 advisor finding): the mailbox still holds the flags 1..3 and the
    # records of the first one.  Connecting again without ending it is refused; reset -> export -> connect starts clean.
    import mppi_generic_amd as m  # noqa: F811
    try:
        eng.p2pConnect(handles)
        refused = "not refused"
    except m.MPPIError as e:
        refused = e.status
    eng.p2pReset()
    conn.send((refused, eng.p2pMailboxHandle()))
    handles2 = conn.recv()
    eng.p2pConnect(handles2)
    conn.send("connected")
    conn.recv()
    eng.optimize(2, synchronize=True)
    conn.send(eng.getOptimalControlSeq()[0].tobytes())
    conn.recv()
    eng.close()


def test_p2p_two_processes_share_the_gpu_through_hipipc(gpu):
    """one process per rank, as on a multi-GPU node — here both on the one device: mailbox handles travel as hipIpcMemHandle_t"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")
    world = 2
    pipes, procs = [], []
    for r in range(world):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_ipc_worker, args=(r, world, b, repo))
        p.start()
        pipes.append(a)
        procs.append(p)
    try:
        handles = []
        for a in pipes:
            assert a.poll(120), "worker did not produce its mailbox handle"
            handles.append(a.recv())
        for a in pipes:
            a.send(handles)
        for a in pipes:
            assert a.poll(120) and a.recv() == "connected"
        for a in pipes:
            a.send("go")
        us = []
        for a in pipes:
            assert a.poll(120), "worker did not finish"
            us.append(np.frombuffer(a.recv(), np.float32))
        for a in pipes:
            a.send("next")
        handles2 = []
        for a in pipes:
            assert a.poll(120), "worker did not reach the second session"
            refused, hnd = a.recv()
            assert refused == 7, refused  # MPPI_ERR_STATE: a live session must be ended with mppi_p2p_reset first
            handles2.append(hnd)
        for a in pipes:
            a.send(handles2)
        for a in pipes:
            assert a.poll(120) and a.recv() == "connected"
        for a in pipes:
            a.send("go")
        us2 = []
        for a in pipes:
            assert a.poll(120), "worker did not finish the second session"
            us2.append(np.frombuffer(a.recv(), np.float32))
        for a in pipes:
            a.send("bye")
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    cfg = cartpole_cfg(K=4096, T=100, soft=True)
    full = make_engine(cfg)
    full.uploadState(cfg["x0"])
    full.optimize(3)
    u_full = full.getOptimalControlSeq()[0].reshape(-1)
    assert np.array_equal(us[0], us[1])
    assert np.abs(us[0] - u_full).max() <= 5e-6
    full.optimize(2)  # the second session continues from the first one's mean (generations 3, 4)
    u_full2 = full.getOptimalControlSeq()[0].reshape(-1)
    full.close()
    assert np.array_equal(us2[0], us2[1])
    assert np.abs(us2[0] - u_full2).max() <= 1e-5


def test_bench_self_launches_two_ranks_from_plain_python(gpu):
    """`python3 bench.py --gpus 2` with NO launcher around it (the form the driver used for N = 1) spawns its own ranks,
    negotiates the exchange, and prints ONE JSON line whose headline is the strong-scaling BASELINE problem with the weak line
    and the other workload beside it.  Both ranks share this box's one GPU (MPPI_BENCH_DEVICE=0)."""
    import json
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPPI_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--min-time", "0.05"], capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 20
    assert out["config"]["global_rollouts"] == 16384 and out["config"]["rollouts_per_gpu"] == 8192
    assert out["finite"] and out["value"] > 0
    neg = out["config"]["exchange_negotiation"]
    assert neg["mode"] in ("p2p", "rccl", "host") and neg["bit_equal_u_across_ranks"] is True
    assert out["weak"]["global_rollouts"] == 32768 and out["weak"]["value"] > 0
    for key in ("autorally_strong", "autorally_weak"):
        assert "error" not in out[key], out[key]
        assert out[key]["value"] > 0 and out[key]["finite"]
    assert out["autorally_strong"]["global_rollouts"] == 16384
    # round 5: BASELINE config 5 (K = 65536 in total, the one problem with more than one round of blocks per CU) as a strong leg
    c5 = out["lstm_colored_strong"]
    assert "error" not in c5, c5
    assert c5["global_rollouts"] == 65536 and c5["rollouts_per_gpu"] == 32768 and c5["value"] > 0 and c5["finite"]
    assert c5["predicted_ms_per_step"] > 0 and out["predicted_ms_per_step"] > 0
    # ... and BOTH exchange paths timed (or the reason a path cannot run here, as a string — never silence)
    xp = out["exchange_paths"]
    assert xp["p2p"]["ok"] and out["exchange_p2p_us"] > 0, xp
    assert xp["rccl"]["ok"] or (isinstance(xp["rccl"]["refused"], str) and len(xp["rccl"]["refused"]) > 0), xp
    if xp["rccl"]["ok"]:
        assert out["exchange_rccl_us"] > 0 and out["rccl_ranks"] == 2
