#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle (oracle/), whose pieces are pinned
against the reference's own known-answer tests in tests/test_oracle_kat.py.  Run from the repo root:
    python tests/golden/make_golden.py
The fixtures freeze oracle outputs for small seeded cases so that (a) a change of the oracle is visible in review and
(b) the GPU parity tests can also be checked against data that does not depend on the oracle build of the day.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

import pyoracle as po  # noqa: E402
from common import (autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, cartpole_cfg_lr, di_cfg, host_noise, host_spectrum,  # noqa: E402
                    make_oracle)


def vanilla_case(cfg, seed):
    eps = host_noise(cfg["num_iters"], cfg["K"], cfg["T"], len(cfg["std_dev"]), seed=seed)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    return dict(eps_seed=seed, costs=o.costs(), control=o.control(), state=o.state_traj(),
                baseline=o.stats()["baseline"][:1], normalizer=o.stats()["normalizer"][:1])


def main():
    out = {}
    for name, cfg in (("cartpole_example_K256_T40", cartpole_cfg(K=256, T=40)),
                      ("cartpole_soft_K256_T40_it2", cartpole_cfg(K=256, T=40, soft=True, num_iters=2)),
                      ("cartpole_lr_K256_T30", cartpole_cfg_lr(K=256, T=30))):
        for k, v in vanilla_case(cfg, 2024).items():
            out[name + "/" + k] = np.asarray(v)
    # tube double integrator, 3 closed-loop calls
    cfg = di_cfg(K=256, T=30, tube=True, num_iters=1)
    o = make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(3):
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=500 + i)
        o.tube_compute_control(x, 1, eps)
        out["di_tube_K256_T30/control_%d" % i] = o.control()
        out["di_tube_K256_T30/nominal_control_%d" % i] = o.nominal_control()
        out["di_tube_K256_T30/baseline_%d" % i] = o.stats()["baseline"].copy()
        x = x + np.array([0.05, -0.03, 0.2, -0.1], np.float32)
    # generator stream
    out["philox/seed42_gen3_K8_T10_C2"] = po.philox_normal(42, 3, 8, 10, 2)
    # det_math spot values (bit patterns)
    xs = np.linspace(-9, 9, 37).astype(np.float32)
    for f, nm in ((0, "sin"), (1, "cos"), (2, "exp"), (4, "tanh"), (5, "atan")):
        out["det/" + nm] = po.det_eval(f, xs)
    out["det/x"] = xs
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), len(out), "arrays")
    models_and_controllers()


def models_and_controllers():
    """second fixture file: the NN models, the colored-noise sampler and Robust MPPI"""
    out = {}
    for name, cfg in (("autorally_K256_T30", autorally_cfg(K=256, T=30)), ("bicycle_lstm_K256_T30", bicycle_lstm_cfg(K=256, T=30))):
        for k, v in vanilla_case(cfg, 77).items():
            out[name + "/" + k] = np.asarray(v)
    # colored noise: the sampler's output for a seeded spectrum, and one ColoredMPPI call
    z = host_spectrum(1, 256, 30, 2, seed=31)
    exps, decay, fmin = [1.0, 0.5], 0.97, 0.0
    out["colored/eps_gemm"] = po.colored_noise(z[0], exps, decay, fmin, 1, flavour="gemm")
    out["colored/eps_definition"] = po.colored_noise(z[0], exps, decay, fmin, 1, flavour="definition")
    cfg = bicycle_lstm_cfg(K=256, T=30)
    o = make_oracle(cfg)
    o.colored_compute_control(cfg["x0"], 1, z, exps, decay, fmin)
    out["colored/bicycle_lstm_control"] = o.control()
    out["colored/bicycle_lstm_costs"] = o.costs()
    # Robust MPPI on the double integrator: two steps (the second one evaluates the candidates)
    cfg = di_cfg(K=576, T=30, tube=True, num_iters=1)
    cfg["control_cost_coeff"] = [0.3, 0.2]
    o = make_oracle(cfg)
    rob = po.RobustOracle(o, 25.0, 9, 32)
    gains = np.random.default_rng(5).uniform(-0.3, 0.3, (30, 4, 2)).astype(np.float32)
    out["rmppi/gains"] = gains
    x = cfg["x0"].copy()
    for i in range(2):
        eps = host_noise(2, cfg["K"], cfg["T"], 2, seed=900 + i)
        rob.update_importance_sampling(x, 2, eps[0])
        rob.set_gains(gains)
        rob.compute_control(x, 1, eps[1:])
        ns, best, stride, fe = rob.state()
        out["rmppi/control_%d" % i] = o.control()
        out["rmppi/nominal_control_%d" % i] = o.nominal_control()
        out["rmppi/nominal_state_%d" % i] = ns
        out["rmppi/best_stride_%d" % i] = np.array([best, stride], np.int32)
        out["rmppi/free_energy_%d" % i] = fe
        x = x + np.array([0.1, -0.05, 0.2, -0.1], np.float32)
    np.savez_compressed(os.path.join(HERE, "oracle_golden_models.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden_models.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
