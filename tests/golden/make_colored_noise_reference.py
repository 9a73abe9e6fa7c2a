"""Golden vectors for the colored-noise sampler, produced by the REFERENCE's own implementation.

Imports /root/reference/scripts/colored_noise.py (felixpatzelt's powerlaw_psd_gaussian, the algorithm
sampling_distributions/colored_noise/colored_noise.cu:285-392 restates on the GPU), replaces the module's `normal`
with a recording generator so that the N(0,1) draws behind every Fourier coefficient are known, and stores
  z   [K][C][T+1][2] float32   the raw draws (real, imaginary) in the engine's / oracle's spectrum layout
  y   [K][T][C]      float64   the script's output, first T of its 2T samples (what the CUDA code keeps, :39-56)
per case.  Run in the build container (the reference tree is not on the GPU box); the .npz is committed:
    python tests/golden/make_colored_noise_reference.py
"""
import importlib.util
import os
import sys

import numpy as np

REF_SCRIPT = "/root/reference/scripts/colored_noise.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colored_noise_reference.npz")

# (name, K, T, exponents per control, fmin)
CASES = [("pink_T50", 24, 50, [1.0, 0.5], 0.0), ("white_brown_T33", 16, 33, [0.0, 2.0], 0.0),
         ("cutoff_T100", 8, 100, [1.0, 1.5], 0.05), ("config5_T200", 8, 200, [1.0, 1.0], 0.0)]


def load_reference_module():
    spec = importlib.util.spec_from_file_location("ref_colored_noise", REF_SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_case(mod, K, T, exponents, fmin, seed):
    """-> (z [K][C][T+1][2] float32, y [K][T][C] float64) from the reference script with recorded draws"""
    rng = np.random.Generator(np.random.Philox(seed))
    C = len(exponents)
    z = np.zeros((K, C, T + 1, 2), np.float32)
    y = np.zeros((K, T, C), np.float64)
    for c, beta in enumerate(exponents):
        draws = []

        def recording_normal(scale=1.0, size=None):
            d = rng.standard_normal(size, dtype=np.float32)  # float32 draws: exactly representable on both sides
            draws.append(d)
            return d.astype(np.float64) * scale

        mod.normal = recording_normal
        out = mod.powerlaw_psd_gaussian(beta, (K, 2 * T), fmin)
        assert len(draws) == 2 and out.shape == (K, 2 * T)
        z[:, c, :, 0] = draws[0]
        z[:, c, :, 1] = draws[1]
        y[:, :, c] = out[:, :T]
    return z, y


def main():
    mod = load_reference_module()
    blob = {}
    for i, (name, K, T, exps, fmin) in enumerate(CASES):
        z, y = run_case(mod, K, T, exps, fmin, seed=1000 + i)
        blob[name + "_z"] = z
        blob[name + "_y"] = y
        blob[name + "_exponents"] = np.asarray(exps, np.float64)
        blob[name + "_fmin"] = np.asarray([fmin], np.float64)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    sys.exit(main())
