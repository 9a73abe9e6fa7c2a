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
