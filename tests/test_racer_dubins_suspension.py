"""RacerDubinsElevationSuspension (reference: dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cu):
LSTM steering + spring / damper suspension over a height map and a normals map.

Pinning.  The reference's tests for this class (tests/dynamics/racer_dubins_elevation_suspension_test.cu) compare its GPU and
CPU paths with each other on random data — there is no known answer in the repository, and the network file is an LFS stub.
What is checked here instead: the suspension equations against an independent float64 restatement (wheel geometry, forces,
the three accelerations, the force outputs, the reference's quirks: front-wheel heading offset 4 / -9.1, rear wheels'
sides), physical sanity (rest on flat ground is an equilibrium; a pitched body is pushed back), and HIP against the oracle
bit for bit.  DESIGN.md lists the class as "parity by restatement, unpinned"."""
import math

import numpy as np
import pytest

import pyoracle as po
from common import host_noise, m, make_engine, make_oracle, ulp_diff
from test_racer_dubins_elevation import hills
from test_racer_dubins_lstm_steering import steering_blobs

(S_VEL, S_YAW, S_X, S_Y, S_STEER, S_BRAKE, S_ROLL, S_PITCH, S_CGZ, S_CGVZ, S_ROLL_RATE, S_PITCH_RATE, S_STEER_RATE) = range(13)
O_POS_Z, O_F_UP, O_F_FWD, O_F_SIDE = 4, 10, 11, 12
NS = 24
WHEELS = [(2.981, 0.737), (2.981, -0.737), (0.0, -0.737), (0.0, 0.737)]   # FL, FR, BL, BR as the reference places them


def st(*v):
    x = np.zeros(NS, np.float32)
    x[:len(v)] = v
    return x


def normals_of(z, res):
    """unit normals of a height field z[row = y][col = x] sampled every `res` metres: {h, w, 4}"""
    dzdy, dzdx = np.gradient(z.astype(np.float64), res)
    n = np.stack([-dzdx, -dzdy, np.ones_like(dzdx), np.zeros_like(dzdx)], axis=-1)
    n[..., :3] /= np.linalg.norm(n[..., :3], axis=-1, keepdims=True)
    return n.astype(np.float32)


def suspension_cfg(K=1024, T=60, lambda_=0.5, D=1, maps="both", zero_net=False):
    cost = m.QuadraticCostParams28()
    coeffs, goal = [0.0] * 28, [0.0] * 28
    coeffs[0], goal[0] = 20.0, 3.0   # BASELINK_VEL_B_X
    coeffs[2], goal[2] = 1.0, 8.0    # BASELINK_POS_I_X
    coeffs[3], goal[3] = 1.0, 3.0    # BASELINK_POS_I_Y
    coeffs[6] = 30.0                 # ROLL
    coeffs[7] = 10.0                 # PITCH
    coeffs[9] = 0.05                 # STEER_ANGLE_RATE
    coeffs[10] = 1e-7                # WHEEL_FORCE_UP_MAX
    coeffs[12] = 1e-7                # WHEEL_FORCE_SIDE_MAX
    coeffs[17] = coeffs[18] = 5.0    # UNCERTAINTY_POS_X / _Y
    cost.s_coeffs[:] = coeffs
    cost.s_goal[:] = goal
    dyn = m.RacerDubinsSuspensionParams()
    b = dyn.base
    b.c_0 = 0.0
    b.c_t[:] = [5.0, 5.0, 5.0]
    b.c_v[:] = [1.0, 1.0, 1.0]
    b.c_b[:] = [20.0, 20.0, 20.0]
    b.wheel_base = 2.981
    b.steer_angle_scale = -2.45
    x0 = np.zeros(NS, np.float32)
    x0[:8] = [1.0, 0.2, -4.0, -2.0, 0.03, 0.0, 0.0, 0.0]
    x0[13:17] = [0.01, 0.01, 0.001, 0.02]
    blobs = {}
    if maps in ("both", "elevation"):
        z, transform = hills()
        blobs["elevation_map"] = z
        blobs["elevation_map_transform"] = transform
        if maps == "both":
            blobs["normals_map"] = normals_of(z, 0.25)
        # the centre of gravity starts one wheel radius above the terrain under the car
        col, row = int((x0[S_X] + 1.49 + 30.0) / 0.25), int((x0[S_Y] + 30.0) / 0.25)
        x0[S_CGZ] = z[row, col] + 0.32
    else:
        x0[S_CGZ] = 0.32
    blobs.update(steering_blobs(zero=zero_net))
    return dict(model="racer_dubins_elevation_suspension", K=K, T=T, D=D, dt=0.02, lambda_=lambda_, alpha=0.0, num_iters=1,
                dyn=dyn, cost=cost, ranges=[-1.0, 1.0, -1.0, 1.0], std_dev=[0.4, 0.5], control_cost_coeff=[0.0, 0.0], x0=x0,
                blobs=blobs)


def suspension_f64(p, x, height_of, normal_of):
    """float64 restatement of computeSimpleSuspensionStep (…suspension_lstm.cu:199-340): (acc_z, acc_roll, acc_pitch, up_max,
    fwd_max, side_max)"""
    roll, pitch, yaw = float(x[S_ROLL]), float(x[S_PITCH]), float(x[S_YAW])
    cr, sr, cp, sp_, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    M = np.array([[cp * cy, sr * sp_ * cy - cr * sy, cr * sp_ * cy + sr * sy],
                  [cp * sy, sr * sp_ * sy + cr * cy, cr * sp_ * sy - sr * cy],
                  [-sp_, sr * cp, cr * cp]])
    az = aroll = apitch = 0.0
    up, fwd, side = [], [], []
    for i, (bx, by) in enumerate(WHEELS):
        wyaw = yaw + (4 / -9.1 if i < 2 else 0.0)
        c, s = math.cos(wyaw), math.sin(wyaw)
        world = M @ np.array([bx, by, 0.0]) + np.array([float(x[S_X]), float(x[S_Y]), 0.0])
        h = height_of(world)
        n = normal_of(world)
        cgx, cgy = bx - p.c_g[0], by - p.c_g[1]
        pos_z = float(x[S_CGZ]) + roll * cgy - pitch * cgx - p.wheel_radius
        vel_z = float(x[S_CGVZ]) + float(x[S_ROLL_RATE]) * cgy - float(x[S_PITCH_RATE]) * cgx
        h_dot = -(float(x[S_VEL]) * c * n[0] + float(x[S_VEL]) * s * n[1])
        f = -p.spring_k * (pos_z - h) - p.drag_c * (vel_z - h_dot)
        up.append(f)
        fwd.append(abs(f / n[2] * (n[0] * c + n[1] * s + n[2] * -pitch)))
        side.append(abs(f / n[2] * (-n[0] * s + n[1] * c + n[2] * roll)))
        az += f / p.mass
        aroll += f * cgy / p.I_xx
        apitch += -f * cgx / p.I_yy
    return az, aroll, apitch, max(up), max(fwd), max(side)


def test_oracle_rest_on_flat_ground_is_an_equilibrium():
    cfg = suspension_cfg(K=64, T=4, maps="none", zero_net=True)
    o = make_oracle(cfg)
    x = st(0.0, 0.3, 1.0, 2.0, 0.0, 0.0, 0.0, 0.0, 0.32)
    xn, xd, y = o.model_step_full(x, np.zeros(2, np.float32), 0.02)
    assert np.abs(xd[[S_CGZ, S_CGVZ, S_ROLL, S_PITCH, S_ROLL_RATE, S_PITCH_RATE]]).max() <= 1e-3   # 14000 N/m * fp32 ulp of 0.32
    assert abs(y[O_POS_Z] - 0.32) <= 1e-6 and abs(y[O_F_UP]) <= 1e-3
    # one centimetre too high: four springs pull down, no roll / pitch moment
    p = cfg["dyn"]
    xn, xd, y = o.model_step_full(st(0, 0, 0, 0, 0, 0, 0, 0, 0.33), np.zeros(2, np.float32), 0.02)
    assert abs(xd[S_CGVZ] + 4 * p.spring_k * 0.01 / p.mass) <= 2e-3 and abs(xd[S_ROLL_RATE]) <= 1e-3 and abs(xd[S_PITCH_RATE]) <= 1e-3
    # nose up by 0.01 rad: the pitch acceleration pushes it back, the body does not accelerate vertically
    xn, xd, y = o.model_step_full(st(0, 0, 0, 0, 0, 0, 0, 0.01, 0.32), np.zeros(2, np.float32), 0.02)
    cgx = 2.981 - p.c_g[0]
    assert abs(xd[S_PITCH_RATE] + 4 * p.spring_k * 0.01 * cgx * cgx / p.I_yy) <= 2e-3 and abs(xd[S_CGVZ]) <= 2e-3
    assert xd[S_PITCH_RATE] < 0


def test_oracle_suspension_against_float64():
    """constant slope and constant normal: lookups are exact, so the restatement isolates the force equations"""
    rng = np.random.default_rng(12)
    sx, sy = 0.06, -0.04
    centres = (np.arange(240) + 0.5) * 0.25 - 30.0
    X, Y = np.meshgrid(centres, centres)
    z = (sx * X + sy * Y).astype(np.float32)
    nvec = np.array([-sx, -sy, 1.0]) / math.sqrt(sx * sx + sy * sy + 1)
    cfg = suspension_cfg(K=64, T=4, maps="both", zero_net=True)
    cfg["blobs"]["elevation_map"] = z
    cfg["blobs"]["normals_map"] = np.broadcast_to(np.append(nvec, 0.0).astype(np.float32), (240, 240, 4)).copy()
    o = make_oracle(cfg)
    p = cfg["dyn"]
    n32 = np.append(nvec, 0.0).astype(np.float32).astype(np.float64)
    for trial in range(100):
        x = st(rng.uniform(-4, 4), rng.uniform(-3, 3), rng.uniform(-15, 15), rng.uniform(-15, 15), rng.uniform(-0.4, 0.4), 0.0,
               rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 0.0, rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3),
               rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5))
        x[S_CGZ] = sx * x[S_X] + sy * x[S_Y] + 0.32 + rng.uniform(-0.03, 0.03)
        xn, xd, y = o.model_step_full(x, np.array([0.3, 0.1], np.float32), 0.02)
        want = suspension_f64(p, x, lambda w: sx * w[0] + sy * w[1], lambda w: n32)
        got = (xd[S_CGVZ], xd[S_ROLL_RATE], xd[S_PITCH_RATE], y[O_F_UP], y[O_F_FWD], y[O_F_SIDE])
        for g_, w_ in zip(got, want):
            assert abs(g_ - w_) <= 2e-3 + 3e-4 * abs(w_), (trial, got, want)   # forces ~ 1e3 N from fp32 heights ~ 1e-7 apart
        assert xd[S_ROLL] == x[S_ROLL_RATE] and xd[S_PITCH] == x[S_PITCH_RATE] and xd[S_CGZ] == x[S_CGVZ]
        # explicit Euler over everything in front of the steering rate; the body height output comes from the c.g.
        assert abs(xn[S_CGVZ] - (x[S_CGVZ] + xd[S_CGVZ] * np.float32(0.02))) <= 1e-6 * max(1, abs(xn[S_CGVZ]))
        assert abs(y[O_POS_Z] - (xn[S_CGZ] + xn[S_PITCH] * p.c_g[0])) <= 1e-6
        assert xn[23] == x[23]


def test_oracle_closed_loop_over_the_hills():
    cfg = suspension_cfg(K=512, T=40)
    o = make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(100):
        o.vanilla_compute_control(x, 1, host_noise(1, cfg["K"], cfg["T"], 2, seed=200 + i))
        u = o.control()[0].copy()
        x, _ = o.model_step(x, u)
        o.vanilla_slide(1)
    assert np.isfinite(x).all() and 1.5 < x[S_VEL] < 3.6 and abs(x[S_ROLL]) < 0.5 and abs(x[S_PITCH]) < 0.5
    z, _ = hills()
    assert abs(x[S_CGZ] - 0.32 - z[int((x[S_Y] + 30) / 0.25), int((x[S_X] + 1.49 + 30) / 0.25)]) < 0.5   # rides on the terrain


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("maps,block_y,variant", [("both", 4, 1), ("both", 4, 2), ("elevation", 4, 1), ("none", 4, 2),
                                                  ("both", 1, 1), ("elevation", 1, 1), ("none", 1, 1)])
def test_suspension_rollout_costs_bit_exact(gpu, maps, block_y, variant):
    """block_y = 4 (the default shape): four replica lanes share out the wheels, the steering network and the covariance
    rows (RacerDubinsElevationSuspensionQuad), fused (1) and role-pipelined (2) kernel; block_y = 1: one lane per rollout"""
    cfg = suspension_cfg(K=1000, T=60, maps=maps)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=3)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    eng = make_engine(cfg, block_x=64, block_y=block_y, kernel_variant=variant)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert np.isfinite(o.costs()).all()
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5
    assert np.abs(eng.getTargetStateSeq() - o.state_traj()).max() <= 1e-4
    y = eng.getTargetOutputSeq()
    xs, ys = o.output_trajectory(cfg["x0"], o.control())
    finite = np.isfinite(ys)
    assert np.array_equal(np.isfinite(y), finite)
    assert np.abs(y[finite] - ys[finite]).max() <= 1e-4 * max(1.0, np.abs(ys[finite]).max())
    assert finite[1:].all() and not finite[0, 10:13].any()   # real wheel forces from the first step on, NaN at t = 0


@pytest.mark.gpu
def test_suspension_model_step_equals_oracle(gpu):
    cfg = suspension_cfg(K=256, T=20)
    o, eng = make_oracle(cfg), make_engine(cfg)
    rng = np.random.default_rng(17)
    for trial in range(100):
        x = st(rng.uniform(-5, 5), rng.uniform(-3, 3), rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-0.5, 0.5),
               rng.uniform(0, 1), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-1, 2), rng.uniform(-1, 1),
               rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-1, 1))
        x[13:23] = rng.uniform(-0.05, 0.05, 10)
        x[23] = rng.uniform(-1, 1)
        u = rng.uniform(-1, 1, 2).astype(np.float32)
        xe, ue = eng.modelStep(x, u)
        xo, uo = o.model_step(x, u)
        same = (xe.view(np.uint32) == xo.view(np.uint32)) | (np.isnan(xe) & np.isnan(xo))
        assert same.all(), (trial, x, u, xe, xo)


@pytest.mark.gpu
def test_suspension_tube_and_colored_closed_loop(gpu):
    cfg = suspension_cfg(K=1024, T=50, D=2)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=4)
    o = make_oracle(cfg)
    o.tube_compute_control(cfg["x0"], 1, eps)
    # no shape: the default (64, 4) with two systems, 512 threads — the instantiation that once returned NaN (see the model's
    # .hip); (32, 4, 2): 256 threads; (64, 1, 2): one lane per rollout
    for shape in ({}, dict(block_x=64, block_y=4), dict(block_x=32, block_y=4), dict(block_x=64, block_y=1)):
        eng = make_engine(cfg, **shape)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
        assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0, shape
        assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5
    cfg = suspension_cfg(K=2048, T=64)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng = make_engine(cfg)
    x = cfg["x0"].copy()
    for i in range(100):
        eng.computeControl(x, 1)
        u = eng.getControlSeq()[0].copy()
        x, _ = eng.modelStep(x, u)
        eng.slideControlSequence(1)
    assert np.isfinite(x).all() and x[S_VEL] > 1.5 and abs(x[S_ROLL]) < 0.5 and abs(x[S_PITCH]) < 0.5
