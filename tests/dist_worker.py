"""Worker of tests/test_distributed_gloo.py: one rank of a world_size-N gloo group (CPU, no GPU).

Each rank owns the rollout slice shard_bounds() assigns, builds its (U_g, rho_g, eta_g) record with the CPU oracle (the
oracle is the record PRODUCER stand-in here — on a GPU box the HIP engine produces the same record), exchanges it with
mppi_generic_amd.distributed.RecordExchange and merges; every rank must end with the unsharded u*."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    import mppi_generic_amd as m  # noqa: F401  (the package import must work without a GPU)
    from mppi_generic_amd import distributed as md
    import pyoracle as po
    from common import cartpole_cfg, di_cfg, make_oracle

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for cfg in (cartpole_cfg(K=512, T=30, soft=True), di_cfg(K=256, T=20, tube=True, lambda_=50.0)):
        K, T, D, lam = cfg["K"], cfg["T"], cfg["D"], cfg["lambda_"]
        orc = make_oracle(cfg)
        C = orc.C
        off, kl = md.shard_bounds(K, rank, world)
        # shard-invariant generator: the slice drawn on its own equals the slice of the full draw
        eps = po.philox_normal(42, 0, K, T, C)
        assert np.array_equal(po.philox_normal(42, 0, K, T, C, off, off + kl), eps[off:off + kl])
        mean = (0.1 * np.cos(np.arange(D * T * C, dtype=np.float32))).reshape(D, T, C)
        x0 = np.tile(cfg["x0"], (D, 1))
        v = orc.set_gaussian_controls(mean, eps, 1, 0)          # rules use the GLOBAL rollout index
        costs, vc = orc.rollout_costs(x0, mean, v)
        n = md.record_floats(T, C, D)
        rec = np.zeros((D, T * C + 4), np.float32)
        for d in range(D):
            s = costs[d, off:off + kl].astype(np.float64)
            rho = s.min()
            w = np.exp(-(s - rho) / lam)
            rec[d, :T * C] = (w[:, None] * vc[d, off:off + kl].reshape(kl, T * C)).sum(0)
            rec[d, T * C:T * C + 3] = [rho, w.sum(), (w * w).sum()]
        ex = md.RecordExchange(n)
        got = ex.all_gather(torch.from_numpy(rec.reshape(-1))).numpy().reshape(world, D, T * C + 4)
        assert np.array_equal(got[rank], rec)                      # own record at own slot: rank order preserved
        u_ref = orc.iterate(x0, mean, eps, 1, 0)                   # unsharded oracle iteration
        for d in range(D):
            u, rho, eta = md.merge_rule_float64(got[:, d, :T * C], got[:, d, T * C], got[:, d, T * C + 1], lam)
            assert abs(rho - costs[d].min()) == 0.0
            assert np.abs(u.reshape(T, C) - u_ref[d]).max() <= 2e-6, (rank, d, np.abs(u.reshape(T, C) - u_ref[d]).max())
            assert abs(eta - orc.stats()["normalizer"][d]) <= 1e-5 * eta
    # ---- Tsallis weights on K-sharded ranks (engine_iteration.hip: iterationShardedTsallis): the weights are not shift-invariant, so the
    # ranks exchange their MINIMA first and only then {sum w v | rho, sum w, sum w^2} under the common baseline — the merge of
    # the second exchange then rescales by exp(0) = 1.  Same two hops here over gloo, against the oracle's un-sharded iteration.
    cfg = cartpole_cfg(K=512, T=30, soft=True)
    K, T, lam = cfg["K"], cfg["T"], cfg["lambda_"]
    gamma, r_exp = 400.0, 1.7
    orc = make_oracle(cfg)
    orc.set_colored_mppi_params(gamma, r_exp, None, False, 1)
    off, kl = md.shard_bounds(K, rank, world)
    eps = po.philox_normal(7, 0, K, T, 1)
    mean = np.zeros((1, T, 1), np.float32)
    x0 = cfg["x0"][None]
    v = orc.set_gaussian_controls(mean, eps, 1, 0)
    costs, vc = orc.rollout_costs(x0, mean, v)
    n = md.record_floats(T, 1, 1)
    ex = md.RecordExchange(n)
    rec = np.zeros((1, T + 4), np.float32)
    rec[0, T] = costs[0, off:off + kl].min()                                      # exchange 1: the local minimum
    got = ex.all_gather(torch.from_numpy(rec.reshape(-1))).numpy().reshape(world, T + 4)
    rho = np.float32(got[:, T].min())
    d = costs[0, off:off + kl].astype(np.float64) - float(rho)
    w = np.where(d < gamma, np.exp(np.log(np.maximum(1.0 - d / gamma, 1e-300)) / (r_exp - 1.0)), 0.0)
    rec[0, :T] = (w[:, None] * vc[0, off:off + kl].reshape(kl, T)).sum(0)
    rec[0, T:T + 3] = [rho, w.sum(), (w * w).sum()]                              # exchange 2: the record under the common rho
    got = ex.all_gather(torch.from_numpy(rec.reshape(-1))).numpy().reshape(world, T + 4)
    assert (got[:, T] == rho).all()
    u, rho_m, eta = md.merge_rule_float64(got[:, :T], got[:, T], got[:, T + 1], lam)  # all rho equal: every scale factor is 1
    u_ref = orc.iterate(x0, mean, eps, 1, 0)[0]
    assert rho_m == orc.stats()["baseline"][0]
    assert np.abs(u.reshape(T, 1) - u_ref).max() <= 2e-6, np.abs(u.reshape(T, 1) - u_ref).max()
    assert abs(eta - orc.stats()["normalizer"][0]) <= 1e-5 * eta
    # ---- Robust MPPI's candidate evaluation sharded BY CANDIDATE (engine_controllers.hip: rmNominalStateAndStride): rank r owns the
    # candidates [r * ceil(nc / world), ...) and writes their costs at their position of the nc x ns array on every rank; the
    # slices must tile the array exactly once, for any world size, including ranks that own nothing
    for nc, ns in ((9, 32), (3, 64), (5, 8)):
        chunk = (nc + world - 1) // world
        lo, hi = min(nc, rank * chunk), min(nc, rank * chunk + chunk)
        mine = torch.zeros(nc * ns)
        mine[lo * ns:hi * ns] = 1.0 + rank
        dist.all_reduce(mine)  # what the aux mailbox assembles: every slot written by exactly one rank
        owner = np.repeat(np.minimum(np.arange(nc) // chunk, world - 1), ns)
        assert np.array_equal(mine.numpy(), 1.0 + owner), (nc, ns, world)
    # bad shardings are refused
    for bad in ((10, 0, 3), (8, 2, 2)):
        try:
            md.shard_bounds(*bad)
        except ValueError:
            pass
        else:
            raise AssertionError("shard_bounds accepted %r" % (bad,))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
