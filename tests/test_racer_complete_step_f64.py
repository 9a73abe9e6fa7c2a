"""RacerDubinsElevationLSTMUncertainty::step as ONE float64 restatement (numpy), written from the reference's source
(dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cu:496-605 and what it calls), independent of the C++ oracle: the
other tests of the class check its pieces; this one checks the WIRING — which derivative feeds which network, what is
integrated from what, which state the process noise and the static settling read — on random states over a sloped plane
(constant slope and normal: the map lookups are exact, as in tests/test_racer_dubins_suspension.py).

The oracle (fp32, the reference's operation order) must agree with the float64 step to fp32 accuracy in every one of the 26
next-state entries, the derivatives and the outputs the cost reads."""
import math

import numpy as np

from common import make_oracle
from test_racer_dubins_lstm_unc import (S_BRAKE, S_CGVZ, S_CGZ, S_OMEGA, S_PITCH, S_PITCH_RATE, S_ROLL, S_ROLL_RATE, S_STATIC_PITCH,
                                        S_STATIC_ROLL, S_STEER, S_STEER_RATE, S_VEL, S_X, S_Y, S_YAW, UNC, NS, st, uncertainty_cfg)
from test_racer_dubins_suspension import O_F_FWD, O_F_SIDE, O_F_UP, O_POS_Z, WHEELS, suspension_f64

PI = math.pi


def normalize_angle(a):
    """angle_utils::normalizeAngle: into [-pi, pi)"""
    return (a + PI) % (2 * PI) - PI if (a + PI) % (2 * PI) >= 0 else (a + PI) % (2 * PI) + PI


def lstm_step_f64(lstm, out, I, H, layers, x):
    """one LSTM step from the blob's (h0, c0) + output network on [h ; x] (utils/nn_helpers/lstm_helper.cu:407-470);
    lstm: [W_im W_fm W_om W_cm | W_ii W_fi W_oi W_ci | b_i b_f b_o b_c | h0 | c0]"""
    L = lstm.astype(np.float64)
    Wm = [L[g * H * H:(g + 1) * H * H].reshape(H, H) for g in range(4)]
    o = 4 * H * H
    Wi = [L[o + g * H * I:o + (g + 1) * H * I].reshape(H, I) for g in range(4)]
    o += 4 * H * I
    b = [L[o + g * H:o + (g + 1) * H] for g in range(4)]
    h, c = L[o + 4 * H:o + 5 * H], L[o + 5 * H:o + 6 * H]
    sig = lambda v: 1 / (1 + np.exp(-v))
    xin = np.zeros(I)
    xin[:len(x)] = x
    gi, gf, go = sig(Wm[0] @ h + Wi[0] @ xin + b[0]), sig(Wm[1] @ h + Wi[1] @ xin + b[1]), sig(Wm[2] @ h + Wi[2] @ xin + b[2])
    gc = np.tanh(Wm[3] @ h + Wi[3] @ xin + b[3])
    c = gi * gc + gf * c
    h = go * np.tanh(c)
    act = np.concatenate([h, xin])
    O = out.astype(np.float64)
    n0, n1, n2 = layers
    W1, b1 = O[:n1 * n0].reshape(n1, n0), O[n1 * n0:n1 * n0 + n1]
    o = n1 * n0 + n1
    W2, b2 = O[o:o + n2 * n1].reshape(n2, n1), O[o + n2 * n1:o + n2 * n1 + n2]
    return W2 @ np.tanh(W1 @ act + b1) + b2


def complete_step_f64(p, blobs, x, u, dt, height_of, normal_of, complete=True):
    """(next_state, state_der, {output index: value}) of the reference's device step(), in float64.
    complete: RacerDubinsElevationLSTMUncertainty (26 states); else its parent RacerDubinsElevationSuspension (24 states:
    linear brake lag, no mean network, the parametric process noise, no static-settling states —
    racer_dubins_elevation_suspension_lstm.cu:343-392)"""
    if complete:
        b, e, s = p.base, p.suspension.elevation, p.suspension
        unc0 = UNC
    else:
        b, e, s = p.base, p.elevation, p
        unc0 = 13
    x = x.astype(np.float64)
    u0, u1 = float(u[0]), float(u[1])
    xd, xn = np.zeros(len(x)), x.copy()
    v = x[S_VEL]
    err = (-u0 if u0 < 0 else 0.0) - x[S_BRAKE]
    if complete:  # quadratic brake lag (racer_dubins_elevation_lstm_unc.cu:510-520)
        xd[S_BRAKE] = min(max((err > 0) * (err * p.pos_quad_brake_c[0] + err * abs(err) * p.pos_quad_brake_c[1]) +
                              (err < 0) * (err * p.neg_quad_brake_c[0] + err * abs(err) * p.neg_quad_brake_c[1]),
                              -b.max_brake_rate_neg), b.max_brake_rate_pos)
    else:  # computeParametricDelayDeriv (racer_dubins.cu:281-293)
        xd[S_BRAKE] = min(max((err > 0) * err * b.brake_delay_constant + (err < 0) * err * b.brake_delay_constant_neg,
                              -b.max_brake_rate_neg), b.max_brake_rate_pos)
    # computeParametricAccelDeriv (racer_dubins_elevation.cu:759-798)
    idx = int(0.2 < abs(v) <= 3.0) + 2 * int(abs(v) > 3.0)
    bs = min(max(x[S_BRAKE], 0.0), 0.25)
    throttle = b.c_t[idx] * u0
    brake = b.c_b[idx] * bs * (-1.0 if v >= 0 else 1.0)
    if abs(v) <= 0.2:
        throttle = b.c_t[idx] * max(u0 - b.low_min_throttle, 0.0)
        brake = b.c_b[idx] * bs * -v
    ax = (0.0 if u0 < 0 else 1.0) * throttle * b.gear_sign + brake - b.c_v[idx] * v + b.c_0
    ax = min(max(ax, -e.clamp_ax), e.clamp_ax)
    if abs(x[S_PITCH]) < PI / 2:
        ax -= b.gravity * math.sin(normalize_angle(x[S_PITCH]))
    xd[S_VEL] = ax
    xd[S_YAW] = (v / b.wheel_base) * math.tan(normalize_angle(x[S_STEER] / b.steer_angle_scale))
    yaw_n = normalize_angle(x[S_YAW])
    xd[S_X], xd[S_Y] = v * math.cos(yaw_n), v * math.sin(yaw_n)
    # computeLSTMSteering (racer_dubins_elevation_lstm_steering.cu:132-168)
    pa = (u1 * b.steer_command_angle_scale - x[S_STEER]) * b.steering_constant
    sr = max(min((pa - x[S_STEER_RATE]) * b.steer_accel_constant - x[S_STEER_RATE] * b.steer_accel_drag_constant,
                 b.max_steer_rate), -b.max_steer_rate)
    nn = lstm_step_f64(blobs["lstm_weights"], blobs["lstm_output_weights"], 4, 4, (8, 20, 1),
                       [x[S_STEER] * 0.2, x[S_STEER_RATE] * 0.2, u1, sr * 0.2])
    xd[S_STEER_RATE] = sr + nn[0] * 5.0
    xd[S_STEER] = x[S_STEER_RATE]
    # computeSimpleSuspensionStep (racer_dubins_elevation_suspension_lstm.cu:199-340)
    az, aroll, apitch, f_up, f_fwd, f_side = suspension_f64(s, x.astype(np.float32), height_of, normal_of)
    xd[S_CGVZ], xd[S_ROLL_RATE], xd[S_PITCH_RATE] = az, aroll, apitch
    xd[S_ROLL], xd[S_PITCH], xd[S_CGZ] = x[S_ROLL_RATE], x[S_PITCH_RATE], x[S_CGVZ]
    # mean network, forward gear only (:526-585): corrects dv/dt and dyaw/dt
    thr, brk = (u0 if u0 >= 0 else 0.0), (-u0 if u0 <= 0 else 0.0)
    if complete and b.gear_sign == 1:
        mo = lstm_step_f64(blobs["mean_lstm_weights"], blobs["mean_lstm_output_weights"], 12, 4, (16, 20, 2),
                           [v, x[S_OMEGA], x[S_BRAKE], x[S_STEER], x[S_STEER_RATE], thr, brk, u1, math.sin(x[S_STATIC_PITCH]),
                            xd[S_VEL], xd[S_YAW]])
        xd[S_VEL] += mo[0]
        xd[S_YAW] += mo[1]
    if complete:
        xn[S_OMEGA] = xd[S_YAW]
    # updateState (racer_dubins_elevation_suspension_lstm.cu:394-418): Euler over everything in front of the steering rate
    for i in range(S_STEER_RATE):
        xn[i] = x[i] + xd[i] * dt
    xn[S_YAW] = normalize_angle(xn[S_YAW])
    xn[S_STEER] = max(min(xn[S_STEER], b.max_steer_angle), -b.max_steer_angle)
    xn[S_STEER_RATE] = x[S_STEER_RATE] + xd[S_STEER_RATE] * dt
    xn[S_BRAKE] = min(max(xn[S_BRAKE], 0.0), 1.0)
    # computeUncertaintyPropagation (racer_dubins_elevation.cu) with the network's process noise (lstm_unc.cu:300-494)
    delta = x[S_STEER] / b.steer_angle_scale
    sy, cy, t, c2 = math.sin(yaw_n), math.cos(yaw_n), math.tan(delta), math.cos(delta) ** 2
    A = np.zeros((4, 4))
    A[0] = [-b.c_v[idx] - e.K_vel_x - (idx == 0) * b.c_b[0] * bs, 0, -e.K_x * cy, -e.K_x * sy]
    A[1] = [t / b.wheel_base, -abs(v) * e.K_yaw / (b.wheel_base * c2), v * e.K_y * sy / (b.wheel_base * c2),
            -v * e.K_y * cy / (b.wheel_base * c2)]
    A[2] = [cy, -sy * v, 0, 0]
    A[3] = [sy, cy * v, 0, 0]
    Q = np.zeros((4, 4))
    # (in reverse the reference first calls the parent's computeQ and then — there is no return — overwrites all sixteen
    # entries with the network's, :305-309: the network's process noise applies in both gears)
    if complete:
        uo = lstm_step_f64(blobs["unc_lstm_weights"], blobs["unc_lstm_output_weights"], 13, 4, (17, 20, 5),
                           [v, x[S_OMEGA], x[S_BRAKE], x[S_STEER], x[S_STEER_RATE], thr, brk, u1, math.sin(x[S_STATIC_ROLL]),
                            math.sin(x[S_STATIC_PITCH]), xd[S_VEL], xd[S_YAW]])
        uo = np.abs(1 / (1 + np.exp(-uo)) * np.array(p.unc_scale[:5], np.float64))
        Q[0, 0] = uo[0] + (b.c_b[idx] * (v if idx == 0 else 1.0)) ** 2 * uo[4]
        Q[1, 1] = uo[1] + ((v / b.wheel_base) / (math.cos(delta) ** 2 * b.steer_angle_scale)) ** 2 * uo[3]
        Q[2, 2], Q[3, 3] = uo[2] * sy * sy, uo[2] * cy * cy
        Q[2, 3] = Q[3, 2] = -uo[2] * sy * cy
    else:  # the parametric process noise (racer_dubins_elevation.cu:421-506, device branch)
        side = v * v * t / b.wheel_base + b.gravity * math.sin(x[S_ROLL])
        q11 = abs(e.Q_y_f * abs(side) * max(abs(v) - 2, 0.0))
        Q[0, 0] = e.Q_x_acc * abs(xd[S_VEL]) + e.Q_x_v[idx] * abs(v)
        Q[1, 1] = abs(v) * (e.Q_omega_steering * abs(delta) + e.Q_omega_v)
        Q[2, 2], Q[3, 3] = q11 * sy * sy, q11 * cy * cy
        Q[2, 3] = Q[3, 2] = -q11 * sy * cy
    un = x[unc0:unc0 + 10]  # POS_X, POS_Y, YAW, VEL_X, POS_X_Y, POS_X_YAW, POS_X_VEL_X, POS_Y_YAW, POS_Y_VEL_X, YAW_VEL_X
    S = np.array([[un[3], un[9], un[6], un[8]], [un[9], un[2], un[5], un[7]], [un[6], un[5], un[0], un[4]],
                  [un[8], un[7], un[4], un[1]]])  # matrix order (v, yaw, x, y)
    F = np.eye(4) + A * dt
    Sn = F @ S @ F.T + Q * dt
    xn[unc0:unc0 + 10] = [Sn[2, 2], Sn[3, 3], Sn[1, 1], Sn[0, 0], Sn[3, 2], Sn[2, 1], Sn[2, 0], Sn[3, 1], Sn[3, 0], Sn[1, 0]]
    y = {0: xn[S_VEL], 2: xn[S_X], 3: xn[S_Y], O_POS_Z: xn[S_CGZ] - xn[S_PITCH] * (-s.c_g[0]), O_F_UP: f_up, O_F_FWD: f_fwd,
         O_F_SIDE: f_side}  # setOutputs (racer_dubins_elevation_suspension_lstm.cu:438-): the entries the cost reads
    if not complete:
        return xn, xd, y
    # static settling at the NEXT pose with the current static angles as the body attitude (racer_dubins.cu:359-433)
    roll, pitch, yaw = x[S_STATIC_ROLL], x[S_STATIC_PITCH], xn[S_YAW]
    cr, sr_, cp, sp_, cyw, syw = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    M = np.array([[cp * cyw, sr_ * sp_ * cyw - cr * syw, cr * sp_ * cyw + sr_ * syw],
                  [cp * syw, sr_ * sp_ * syw + cr * cyw, cr * sp_ * syw - sr_ * cyw],
                  [-sp_, sr_ * cp, cr * cp]])
    hh = {}
    for name, (bx, by) in zip(("fl", "fr", "rl", "rr"), ((2.981, 0.737), (2.981, -0.737), (0.0, 0.737), (0.0, -0.737))):
        hh[name] = height_of(M @ np.array([bx, by, 0.0]) + np.array([xn[S_X], xn[S_Y], 0.0]))
    clamp = lambda a, lim: max(min(a, lim), -lim)
    front, rear = clamp(hh["fl"] - hh["fr"], 0.736 * 2), clamp(hh["rl"] - hh["rr"], 0.736 * 2)
    xn[S_STATIC_ROLL] = (math.asin(front / (0.737 * 2)) + math.asin(rear / (0.737 * 2))) / 2
    left, right = clamp(hh["rl"] - hh["fl"], 2.98), clamp(hh["rr"] - hh["fr"], 2.98)
    xn[S_STATIC_PITCH] = (math.asin(left / 2.981) + math.asin(right / 2.981)) / 2
    return xn, xd, y


def test_oracle_whole_step_against_float64():
    rng = np.random.default_rng(77)
    sx, sy = 0.05, -0.03
    centres = (np.arange(240) + 0.5) * 0.25 - 30.0
    X, Y = np.meshgrid(centres, centres)
    z = (sx * X + sy * Y).astype(np.float32)
    nvec = np.array([-sx, -sy, 1.0]) / math.sqrt(sx * sx + sy * sy + 1)
    n32 = np.append(nvec, 0.0).astype(np.float32).astype(np.float64)
    worst = 0.0
    for gear in (1, -1):
        cfg = uncertainty_cfg(K=64, T=4, maps="both")
        cfg["blobs"]["elevation_map"] = z
        cfg["blobs"]["normals_map"] = np.broadcast_to(np.append(nvec, 0.0).astype(np.float32), (240, 240, 4)).copy()
        p = cfg["dyn"]
        p.base.gear_sign = gear
        p.unc_scale[:] = [0.3, 0.2, 0.1, 0.4, 0.05, 1.0, 1.0]
        o = make_oracle(cfg)
        for trial in range(60):
            x = st(rng.uniform(-4, 5), rng.uniform(-3, 3), rng.uniform(-12, 12), rng.uniform(-12, 12), rng.uniform(-0.4, 0.4),
                   rng.uniform(0, 0.6), rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 0.0, rng.uniform(-0.4, 0.4),
                   rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5),
                   rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1))
            x[S_CGZ] = sx * x[S_X] + sy * x[S_Y] + 0.32 + rng.uniform(-0.03, 0.03)
            cov = rng.uniform(-0.05, 0.05, (4, 4))
            cov = cov @ cov.T + 0.01 * np.eye(4)  # a valid covariance in matrix order (v, yaw, x, y)
            x[UNC:UNC + 10] = [cov[2, 2], cov[3, 3], cov[1, 1], cov[0, 0], cov[3, 2], cov[2, 1], cov[2, 0], cov[3, 1], cov[3, 0],
                               cov[1, 0]]
            u = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)], np.float32)
            xn, xd, y = o.model_step_full(x, u, 0.02)
            wn, wd, wy = complete_step_f64(p, cfg["blobs"], x, u, 0.02, lambda w: sx * w[0] + sy * w[1], lambda w: n32)
            for i in range(NS):
                # suspension accelerations are differences of ~1e4 N forces built from fp32 heights ~1e-7 apart
                loose = i in (S_CGVZ, S_ROLL_RATE, S_PITCH_RATE)
                tol = (2e-3 + 3e-4 * abs(wd[i])) if loose else (2e-5 + 2e-5 * abs(wd[i]))
                assert abs(xd[i] - wd[i]) <= tol, (gear, trial, "xd", i, xd[i], wd[i])
                tol = (1e-4 + 1e-5 * abs(wn[i])) if loose else (3e-6 + 3e-6 * abs(wn[i]))
                if i == S_YAW and abs(abs(wn[i]) - PI) < 1e-4:
                    continue  # the wrap point
                assert abs(xn[i] - wn[i]) <= tol, (gear, trial, "xn", i, xn[i], wn[i])
                worst = max(worst, abs(xn[i] - wn[i]) / max(1.0, abs(wn[i])))
            for k, wv in wy.items():
                tol = (2e-3 + 3e-4 * abs(wv)) if k in (O_F_UP, O_F_FWD, O_F_SIDE) else (3e-6 + 3e-6 * abs(wv))
                assert abs(y[k] - wv) <= tol, (gear, trial, "y", k, y[k], wv)
    assert worst < 1e-4


def test_oracle_suspension_whole_step_against_float64():
    """the parent class, RacerDubinsElevationSuspension (24 states), the same way — also listed as unpinned in DESIGN.md"""
    from test_racer_dubins_suspension import suspension_cfg
    from test_racer_dubins_suspension import st as st24
    rng = np.random.default_rng(78)
    sx, sy = -0.04, 0.06
    centres = (np.arange(240) + 0.5) * 0.25 - 30.0
    X, Y = np.meshgrid(centres, centres)
    z = (sx * X + sy * Y).astype(np.float32)
    nvec = np.array([-sx, -sy, 1.0]) / math.sqrt(sx * sx + sy * sy + 1)
    n32 = np.append(nvec, 0.0).astype(np.float32).astype(np.float64)
    for gear in (1, -1):
        cfg = suspension_cfg(K=64, T=4, maps="both")
        cfg["blobs"]["elevation_map"] = z
        cfg["blobs"]["normals_map"] = np.broadcast_to(np.append(nvec, 0.0).astype(np.float32), (240, 240, 4)).copy()
        p = cfg["dyn"]
        p.base.gear_sign = gear
        o = make_oracle(cfg)
        for trial in range(60):
            x = st24(rng.uniform(-4, 5), rng.uniform(-3, 3), rng.uniform(-12, 12), rng.uniform(-12, 12), rng.uniform(-0.4, 0.4),
                     rng.uniform(0, 0.6), rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 0.0, rng.uniform(-0.4, 0.4),
                     rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5))
            x[S_CGZ] = sx * x[S_X] + sy * x[S_Y] + 0.32 + rng.uniform(-0.03, 0.03)
            cov = rng.uniform(-0.05, 0.05, (4, 4))
            cov = cov @ cov.T + 0.01 * np.eye(4)
            x[13:23] = [cov[2, 2], cov[3, 3], cov[1, 1], cov[0, 0], cov[3, 2], cov[2, 1], cov[2, 0], cov[3, 1], cov[3, 0], cov[1, 0]]
            u = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)], np.float32)
            xn, xd, y = o.model_step_full(x, u, 0.02)
            wn, wd, wy = complete_step_f64(p, cfg["blobs"], x, u, 0.02, lambda w: sx * w[0] + sy * w[1], lambda w: n32,
                                           complete=False)
            for i in range(23):  # (entry 23 is the filler the model carries along)
                loose = i in (S_CGVZ, S_ROLL_RATE, S_PITCH_RATE)
                tol = (2e-3 + 3e-4 * abs(wd[i])) if loose else (2e-5 + 2e-5 * abs(wd[i]))
                assert abs(xd[i] - wd[i]) <= tol, (gear, trial, "xd", i, xd[i], wd[i])
                tol = (1e-4 + 1e-5 * abs(wn[i])) if loose else (3e-6 + 3e-6 * abs(wn[i]))
                if i == S_YAW and abs(abs(wn[i]) - PI) < 1e-4:
                    continue
                assert abs(xn[i] - wn[i]) <= tol, (gear, trial, "xn", i, xn[i], wn[i])
            for k, wv in wy.items():
                tol = (2e-3 + 3e-4 * abs(wv)) if k in (O_F_UP, O_F_FWD, O_F_SIDE) else (3e-6 + 3e-6 * abs(wv))
                assert abs(y[k] - wv) <= tol, (gear, trial, "y", k, y[k], wv)
