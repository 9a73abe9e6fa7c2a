"""Shared test fixtures: the synthetic configurations of SURVEY.md §8d, built identically for the HIP engine (through
the C ABI) and for the CPU oracle."""
import numpy as np

import mppi_generic_amd as m
import pyoracle as po

SEED = 42


def cartpole_cfg(K=2048, T=100, lambda_=0.25, num_iters=1, soft=False):
    """examples/cartpole_example.cu:9-51 (reference).  soft=True raises lambda so that many rollouts carry weight —
    the weighted reduction is then a real average, not a copy of the best rollout."""
    cost = m.CartpoleQuadraticCostParams()
    cost.cart_position_coeff = 50
    cost.pole_angle_coeff = 200
    cost.cart_velocity_coeff = 10
    cost.pole_angular_velocity_coeff = 1
    cost.control_cost_coeff[0] = 0
    cost.terminal_cost_coeff = 0
    cost.desired_terminal_state[:] = [20, 0, np.float32(np.pi), 0]
    return dict(model="cartpole", K=K, T=T, D=1, dt=0.02, lambda_=200.0 if soft else lambda_, alpha=0.0,
                num_iters=num_iters, dyn=m.CartpoleDynamicsParams(1.0, 1.0, 1.0), cost=cost,
                ranges=[[-5.0, 5.0]], std_dev=[5.0], control_cost_coeff=[0.0], x0=np.zeros(4, np.float32))


def cartpole_cfg_lr(K=1024, T=50):
    """variant that exercises the likelihood-ratio term, the terminal cost, alpha and a non-zero mean"""
    c = cartpole_cfg(K, T, soft=True)
    c["control_cost_coeff"] = [0.7]
    c["alpha"] = 0.1
    c["cost"].terminal_cost_coeff = 3.0
    c["x0"] = np.array([0.3, -0.2, 0.5, 0.1], np.float32)
    return c


def di_cfg(K=1024, T=50, tube=True, lambda_=2.0, num_iters=1):
    """examples/double_integrator_CORL2020.cu:30-39, 316-352 (reference) at test size"""
    return dict(model="double_integrator", K=K, T=T, D=2 if tube else 1, dt=0.02, lambda_=lambda_, alpha=0.0,
                num_iters=num_iters, dyn=m.DoubleIntegratorParams(1.0), cost=m.DoubleIntegratorCircleCostParams(),
                ranges=None, std_dev=[1.0, 1.0], control_cost_coeff=[0.0, 0.0],
                x0=np.array([2.0, 0.0, 0.0, 1.0], np.float32))


def racer_cfg(K=1024, T=60, lambda_=0.2, num_iters=1):
    """RacerDubins (dynamics/racer_dubins/racer_dubins.cuh defaults) + QuadraticCost over its outputs: hold 1.6 m/s (the
    default parameters give dv/dt = 1.3 u - 3.7 v + 4.9, i.e. 1.32 m/s at u = 0 and 1.68 m/s at full throttle) towards a
    way-point; throttle/brake in [-1, 1], steering command in [-1, 1]"""
    cost = m.QuadraticCostParams28()
    coeffs = [0.0] * 28
    goal = [0.0] * 28
    # plain RacerDubins: outputs 0..6 are the states [VEL_X, YAW, POS_X, POS_Y, STEER_ANGLE, BRAKE_STATE, STEER_ANGLE_RATE]
    coeffs[0], goal[0] = 40.0, 1.6
    coeffs[2], goal[2] = 1.0, 5.0
    coeffs[3], goal[3] = 1.0, 2.0
    coeffs[4] = 0.5
    coeffs[6] = 0.05
    cost.s_coeffs[:] = coeffs
    cost.s_goal[:] = goal
    return dict(model="racer_dubins", K=K, T=T, D=1, dt=0.02, lambda_=lambda_, alpha=0.0, num_iters=num_iters,
                dyn=m.RacerDubinsParams(), cost=cost, ranges=[-1.0, 1.0, -1.0, 1.0], std_dev=[0.4, 0.5],
                control_cost_coeff=[0.0, 0.0], x0=np.array([0.5, 0.3, 0.0, 0.0, 0.05, 0.0, 0.0], np.float32))


def standard_track_map():
    """channel 0 of the reference's `track_map_standard.npz` (scripts/autorally/test/generateTestMaps.py:47-76):
    30 m x 30 m at 20 px/m, value = |15 - y| + x/30 in map coordinates; world bounds x in [-13, 17], y in [-10, 20]"""
    n = 600
    i = np.arange(n, dtype=np.float64)[:, None] / 20.0
    j = np.arange(n, dtype=np.float64)[None, :] / 20.0
    return (np.abs(15.0 - i) + j / 30.0).astype(np.float32), (-13.0, 17.0, -10.0, 20.0)


def autorally_cfg(K=1024, T=50, lambda_=20.0, num_iters=1):
    """SURVEY.md §8d config 4 at test size: NeuralNetModel<7,2,3> (FNN 6-32-32-4, synthetic weights U(-0.3, 0.3) in the
    reference's dynamics_W/b key layout — the real autorally_nnet_09_12_2018.npz is a git-LFS stub) + ARStandardCost on
    the reference's generated standard track map."""
    rng = np.random.default_rng(SEED)
    npz = {}
    layers = [6, 32, 32, 4]
    for i in range(1, 4):
        npz["dynamics_W%d" % i] = rng.uniform(-0.3, 0.3, (layers[i], layers[i - 1])).astype(np.float64)
        npz["dynamics_b%d" % i] = rng.uniform(-0.3, 0.3, layers[i]).astype(np.float64)
    cmap, (x0b, x1b, y0b, y1b) = standard_track_map()
    cost = m.ARStandardCostParams()
    cost.setTransformFromBounds(x0b, x1b, y0b, y1b)
    return dict(model="autorally_nn", K=K, T=T, D=1, dt=0.02, lambda_=lambda_, alpha=0.0, num_iters=num_iters,
                dyn=None, cost=cost, ranges=[[-0.99, 0.99], [-0.99, 0.65]], std_dev=[0.3, 0.3],
                control_cost_coeff=[0.0, 0.0], x0=np.array([-12.0, 5.0, 0.0, 0.0, 4.0, 0.0, 0.0], np.float32),
                blobs={"dynamics_weights": m.fnn_blob_from_npz_dict(npz), "costmap": cmap})


def lstm_npz(seed=SEED, scale=0.3, I=6, H=16, M=32, OUT=4):
    """synthetic network in the reference's LSTM .npz key layout (float64, PyTorch gate order; lstm_helper.cu:514-585)"""
    rng = np.random.default_rng(seed)
    d = {
        "lstm/weight_hh_l0": rng.uniform(-scale, scale, (4 * H, H)),
        "lstm/weight_ih_l0": rng.uniform(-scale, scale, (4 * H, I)),
        "lstm/bias_hh_l0": rng.uniform(-scale, scale, 4 * H),
        "lstm/bias_ih_l0": rng.uniform(-scale, scale, 4 * H),
        "lstm/h0": rng.uniform(-0.5, 0.5, H),
        "lstm/c0": rng.uniform(-0.5, 0.5, H),
        "output/dynamics_W1": rng.uniform(-scale, scale, (M, H + I)),
        "output/dynamics_b1": rng.uniform(-scale, scale, M),
        "output/dynamics_W2": rng.uniform(-scale, scale, (OUT, M)),
        "output/dynamics_b2": rng.uniform(-scale, scale, OUT),
    }
    return d


def bicycle_lstm_cfg(K=1024, T=50, lambda_=20.0, num_iters=1):
    """SURVEY.md §8d config 5 at test size: LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights in
    the reference's LSTM key layout) + ARStandardCost on the generated standard track map."""
    lstm_blob, fnn_blob = m.lstm_blob_from_npz_dict(lstm_npz())
    cmap, (x0b, x1b, y0b, y1b) = standard_track_map()
    cost = m.ARStandardCostParams()
    cost.setTransformFromBounds(x0b, x1b, y0b, y1b)
    return dict(model="bicycle_slip_lstm", K=K, T=T, D=1, dt=0.02, lambda_=lambda_, alpha=0.0, num_iters=num_iters,
                dyn=None, cost=cost, ranges=[[-0.99, 0.99], [-0.99, 0.65]], std_dev=[0.3, 0.3],
                control_cost_coeff=[0.0, 0.0], x0=np.array([-12.0, 5.0, 0.0, 0.0, 4.0, 0.0, 0.0], np.float32),
                blobs={"lstm_weights": lstm_blob, "lstm_output_weights": fnn_blob, "costmap": cmap})


def make_engine(cfg, tube=None, **kw):
    tube = (cfg["D"] == 2) if tube is None else tube
    cls = m.TubeMPPIController if tube else m.VanillaMPPIController
    if cfg.get("colored") is not None:
        cls = m.ColoredMPPIController
    c = cls(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], cfg["alpha"], cfg["num_iters"], seed=SEED, **kw)
    if cfg["dyn"] is not None:
        c.setDynamicsParams(cfg["dyn"])
    c.setCostParams(cfg["cost"])
    for name, arr in cfg.get("blobs", {}).items():
        c.setModelBlob(name, arr)
    if cfg["ranges"] is not None:
        c.setControlRanges(cfg["ranges"])
    c.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"], cfg.get("pure_pct", 0.01), cfg.get("decay", 1.0))
    if cfg.get("colored") is not None:
        c.setColoredNoiseParams(*cfg["colored"])
    return c


def host_spectrum(n_iters, K, T, C, seed=SEED):
    """z[n_iters][K][C][T+1][2] ~ N(0,1): the Gaussian spectrum the colored-noise sampler shapes (reference layout)"""
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n_iters, K, C, T + 1, 2), dtype=np.float32)


def make_oracle(cfg):
    o = po.Oracle(cfg["model"], cfg["K"], cfg["T"], cfg["D"], cfg["dt"], cfg["lambda_"], cfg["alpha"], cfg["num_iters"])
    if cfg["dyn"] is not None:
        o.set_dynamics_params(cfg["dyn"])
    o.set_cost_params(cfg["cost"])
    for name, arr in cfg.get("blobs", {}).items():
        o.set_blob(name, arr)
    if cfg["ranges"] is not None:
        o.set_control_ranges(cfg["ranges"])
    o.set_sampler(cfg["std_dev"], cfg["control_cost_coeff"], cfg.get("pure_pct", 0.01), cfg.get("decay", 1.0))
    return o


def host_noise(n_iters, K, T, C, seed=SEED):
    """eps[n_iters][K][T][C] ~ N(0,1) from a fixed-seed host generator (parity is defined downstream of eps)"""
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n_iters, K, T, C), dtype=np.float32)


def ulp_diff(a, b):
    """distance in units in the last place between two float32 arrays (same sign assumed where it matters)"""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def merge_records_numpy(U, rho, eta, lambda_):
    """float64 restatement of combineKernel's merge rule for host-logic tests"""
    rho_min = rho.min()
    s = np.exp(-(rho - rho_min) / lambda_)
    eta_tot = float((s * eta).sum())
    return (s[:, None] * U).sum(0) / eta_tot, rho_min, eta_tot
