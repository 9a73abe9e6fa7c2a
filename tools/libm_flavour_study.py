#!/usr/bin/env python3
"""How far does the choice of math library move the MPPI result?  (oracle and engine share det_math.h, so
"0 ulp vs oracle" says nothing about the distance to an implementation on other transcendentals — the reference's CUDA path
uses __sinf / __cosf / tanhf / expf.)

The SAME oracle is built twice — on det_math.h's bit-reproducible functions (the checker) and on glibc's sinf / cosf / expf /
logf / tanhf / atanf (`make -C oracle libm`) — and both run one computeControl of each BASELINE configuration on the same
noise.  Reported: max relative difference of the trajectory costs and the L-inf difference of u*.  CPU only.

Usage: python tools/libm_flavour_study.py [--small]     (child mode: --dump <out.npz>)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)


def cases(small):
    from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg
    if small:
        return [("cartpole", cartpole_cfg(K=2048, T=100), "vanilla"), ("cartpole lambda=200", cartpole_cfg(K=2048, T=100, soft=True), "vanilla"),
                ("autorally", autorally_cfg(K=1024, T=60), "vanilla"), ("di tube", di_cfg(K=1024, T=60), "tube"),
                ("lstm", bicycle_lstm_cfg(K=1024, T=60), "vanilla")]
    return [("cartpole K=16384 T=100 lambda=0.25", cartpole_cfg(K=16384, T=100), "vanilla"),
            ("cartpole K=16384 T=100 lambda=200", cartpole_cfg(K=16384, T=100, soft=True), "vanilla"),
            ("autorally-nn K=16384 T=150", autorally_cfg(K=16384, T=150), "vanilla"),
            ("di-tube K=8192 T=150", di_cfg(K=8192, T=150), "tube"),
            ("lstm K=16384 T=200", bicycle_lstm_cfg(K=16384, T=200), "vanilla")]


def dump(path, small):
    import numpy as np
    from common import host_noise, make_oracle
    out = {}
    for i, (name, cfg, kind) in enumerate(cases(small)):
        orc = make_oracle(cfg)
        eps = host_noise(1, cfg["K"], cfg["T"], orc.C, seed=100 + i)
        if kind == "tube":
            orc.tube_compute_control(cfg["x0"], 1, eps)
        else:
            orc.vanilla_compute_control(cfg["x0"], 1, eps)
        out["u%d" % i] = orc.control()
        out["c%d" % i] = orc.costs()
    np.savez(path, **out)


def main():
    small = "--small" in sys.argv
    if "--dump" in sys.argv:
        dump(sys.argv[sys.argv.index("--dump") + 1], small)
        return 0
    import numpy as np
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "all", "libm"], check=True, capture_output=True)
    outs = []
    for tag, lib in (("det", ""), ("libm", os.path.join(REPO, "oracle", "_build", "libmppi_oracle_libm.so"))):
        env = dict(os.environ)
        env.pop("MPPI_ORACLE_LIB", None)
        if lib:
            env["MPPI_ORACLE_LIB"] = lib
        path = "/tmp/libm_study_%s.npz" % tag
        subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", path] + (["--small"] if small else []), check=True, env=env)
        outs.append(np.load(path))
    res = []
    for i, (name, cfg, kind) in enumerate(cases(small)):
        ua, ub = outs[0]["u%d" % i], outs[1]["u%d" % i]
        ca, cb = outs[0]["c%d" % i].astype(np.float64), outs[1]["c%d" % i].astype(np.float64)
        rel = float(np.max(np.abs(ca - cb) / np.maximum(np.abs(ca), 1e-30)))
        du = float(np.abs(ua - ub).max())
        res.append((name, rel, du))
        print("%-38s max rel cost difference %.2e   u* L-inf difference %.2e" % (name, rel, du))
    return res


if __name__ == "__main__":
    main()
