"""The reference's own closed-loop ACCEPTANCE tests, with their literal parameters — the only checks the reference holds that
include the random number generator (every other parity test here is defined downstream of eps):

  SwingUpTest                          tests/controllers/vanilla_mppi_test.cu:79-136   Cartpole, K = 2048, T = 100, dt = 0.01,
                                       lambda = 0.25, alpha = 0.01, sigma = 5, 1000 closed-loop steps => baseline < 1.0
  VanillaMPPINominalVariance           tests/controllers/tube_mppi_test.cu:150-204     double integrator, K = 1024, T = 50,
                                       dt = 0.02, 3 iterations, lambda = 4, plant noise variance 1: 500 steps inside the track
  TubeMPPILargeVariance                tests/controllers/tube_mppi_test.cu:352-497     the same with variance 100, Tube-MPPI,
                                       nominal threshold 100, DDP tracking gains Q = diag(500, 500, 100, 100), Qf = R = I
  RobustMPPILargeVariance              tests/controllers/rmppi_test.cu:561-698         Robust MPPI, lambda = 4, 1 iteration,
                                       value-function threshold 10, variance 100: 5000 steps inside the track
  RobustMPPILargeVarianceRobustCost    tests/controllers/rmppi_test.cu:700-908         DoubleIntegratorRobustCost (crash 100),
                                       lambda = 2, 3 iterations: 5000 steps inside the track

Each loop runs
  * on the ORACLE with the product's Philox stream evaluated on the host (CPU tests: the restatement itself passes the
    reference's thresholds — a pin on behaviour the reference holds, RNG included), and
  * on the HIP ENGINE with its in-kernel Philox draws (gpu tests): the default fused reduction must pass the reference's
    threshold for the literal step count, and a second handle in MPPI_REDUCTION_REFERENCE_ORDER must stay BIT-IDENTICAL to the
    oracle's loop (plant state, control, statistics) on every step of the compared stretch.

Not pinned, because the reference does not pin it either: the plant's disturbance (std::mt19937 seeded from
std::random_device, di_dynamics.cu:3-12, 60-66) — numpy's generator with a fixed seed here — and cuRAND's stream (the reference
never fixes a seed in these tests; any seed must pass; the gpu tests run three).

The DDP gain PRODUCER (include/mppi/ddp/ddp.h, host Eigen code) is outside the hot path: `ddp_gains_linear`
restates its backward pass (ddp.h:93-127) for the double integrator, whose linear dynamics make the gains independent of the
trajectory they are computed around.
"""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import make_engine, make_oracle

DT_DI = np.float32(0.02)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ddp_gains_linear(T, dt, Q, Qf, R):
    """ddp/ddp.h:93-127 (backward pass; Phi = I + A dt, B dt; Vxx symmetrised; Lk = -quu^-1 qux; the last step's gain stays
    zero) for DoubleIntegratorDynamics::computeGrad (di_dynamics.cu:24-34).  Returns (L[T][C][S] for the host feedback
    u_fb = L_t (x - x*), ddp.cuh:144-151, and the same gains in DDPFeedbackState::fb_gain_traj_ order [T][S][C])"""
    A = np.zeros((4, 4))
    A[0, 2] = A[1, 3] = 1
    B = np.zeros((4, 2))
    B[2, 0] = B[3, 1] = 1
    Phi, Bd = np.eye(4) + A * dt, B * dt
    Vxx = 0.5 * (Qf + Qf.T)
    L = np.zeros((T, 2, 4))
    for k in range(T - 2, -1, -1):
        qux = Bd.T @ Vxx @ Phi
        qxx = Q * dt + Phi.T @ Vxx @ Phi
        quu = R * dt + Bd.T @ Vxx @ Bd
        L[k] = np.linalg.solve(quu, -qux)
        Vxx = qxx + qux.T @ L[k]
        Vxx = 0.5 * (Vxx + Vxx.T)
    return L.astype(np.float32), np.ascontiguousarray(L.transpose(0, 2, 1)).astype(np.float32)


def tube_failure(s):
    """tube_mppi_test.cu:10-23"""
    r2 = float(s[0]) * float(s[0]) + float(s[1]) * float(s[1])
    return r2 < 1.675 * 1.675 or r2 > 2.325 * 2.325


def di_plant(x, u, rng, variance):
    """model.computeDynamics + updateState + computeStateDisturbance (tube_mppi_test.cu:191-196, di_dynamics.cu:14-22, 60-66)"""
    x = x + np.array([x[2], x[3], u[0], u[1]], np.float32) * DT_DI
    x[2:] += rng.normal(0.0, np.sqrt(variance), 2).astype(np.float32) * DT_DI
    return x.astype(np.float32)


def swingup_cfg():
    """vanilla_mppi_test.cu:8-31 (fixture) and :81-107 (the test's cost, sampler and controller parameters)"""
    cost = m.CartpoleQuadraticCostParams()
    cost.cart_position_coeff = 100
    cost.pole_angle_coeff = 200
    cost.cart_velocity_coeff = 10
    cost.pole_angular_velocity_coeff = 20
    cost.control_cost_coeff[0] = 1
    cost.terminal_cost_coeff = 0
    cost.desired_terminal_state[:] = [-20, 0, np.float32(np.pi), 0]
    return dict(model="cartpole", K=2048, T=100, D=1, dt=0.01, lambda_=0.25, alpha=0.01, num_iters=1,
                dyn=m.CartpoleDynamicsParams(1.0, 1.0, 1.0), cost=cost, ranges=None, std_dev=[5.0],
                control_cost_coeff=[1.0], pure_pct=0.01, x0=np.zeros(4, np.float32))


def di_acceptance_cfg(D, lambda_, num_iters, robust_cost=False, sampler_cost=1.0):
    """tube_mppi_test.cu:109-148 (DoubleIntegratorTracking fixture), rmppi_test.cu:561-583, 700-728"""
    cost = m.DoubleIntegratorCircleCostParams()
    cost.velocity_desired = 2
    if robust_cost:
        cost.crash_cost = 100
    return dict(model="double_integrator_robust" if robust_cost else "double_integrator", K=1024, T=50, D=D, dt=0.02,
                lambda_=lambda_, alpha=0.0, num_iters=num_iters, dyn=m.DoubleIntegratorParams(1.0), cost=cost, ranges=None,
                std_dev=[1.0, 1.0], control_cost_coeff=[sampler_cost, sampler_cost], x0=np.array([2, 0, 0, 1], np.float32))


class _Stream:
    """the engine's noise stream on the host: one Philox generation per rollout launch (engine_iteration.hip: h->generation)"""

    def __init__(self, seed, K, T, C):
        self.seed, self.K, self.T, self.C, self.g = seed, K, T, C, 0

    def take(self, n):
        e = np.stack([po.philox_normal(self.seed, self.g + i, self.K, self.T, self.C) for i in range(n)])
        self.g += n
        return e


# ---------------------------------------------------------------------------------------------------------------------
# the loops, written once for both sides: `side` is an adapter with the few calls a loop needs
# ---------------------------------------------------------------------------------------------------------------------
class _OracleSide:
    def __init__(self, cfg, seed, kind, thr=None):
        self.o, self.kind, self.cfg = make_oracle(cfg), kind, cfg
        self.s = _Stream(seed, cfg["K"], cfg["T"], len(cfg["std_dev"]))
        self.first = True
        if kind == "tube":
            self.o.set_controller_params(thr)
        if kind == "robust":
            self.r = po.RobustOracle(self.o, thr, 9, 32)

    def set_gains(self, g):
        self.r.set_gains(g)

    def update(self, x):
        self.r.update_importance_sampling(x, 1, None if self.first else self.s.take(1)[0])
        self.first = False

    def compute(self, x):
        eps = self.s.take(self.cfg["num_iters"])
        if self.kind == "vanilla":
            self.o.vanilla_compute_control(x, 1, eps)
        elif self.kind == "tube":
            self.o.tube_compute_control(x, 1, eps)
        else:
            self.r.compute_control(x, 1, eps)

    def control(self):
        return self.o.control()

    def nominal_state0(self):
        return self.o.nominal_state_traj()[0]

    def slide(self):
        (self.o.tube_slide if self.kind == "tube" else self.o.vanilla_slide)(1)

    def baseline(self):
        return float(self.o.stats()["baseline"][1 if self.kind == "robust" else 0])

    def nominal_used(self):
        return self.o.stats()["nominal_state_used"]

    def best_index(self):
        return self.r.state()[1]

    def step(self, x, u):
        return self.o.model_step(x, u)[0]

    def close(self):
        pass


class _EngineSide:
    def __init__(self, cfg, seed, kind, thr=None, reference_order=False):
        self.kind = kind
        if kind == "robust":
            e = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], cfg["alpha"],
                                       cfg["num_iters"], seed=seed)
            e.setDynamicsParams(cfg["dyn"])
            e.setCostParams(cfg["cost"])
            e.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
            e.setRMPPIParams(thr, 9, 32)
        else:
            e = make_engine(cfg, tube=(kind == "tube"))
            if kind == "tube":
                e.setNominalThreshold(thr)
        if reference_order:
            e.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
        e.setSeed(seed)
        self.e = e

    def set_gains(self, g):
        self.e.setFeedbackGains(g)

    def update(self, x):
        self.e.updateImportanceSamplingControl(x, 1)

    def compute(self, x):
        self.e.computeControl(x, 1)

    def control(self):
        return self.e.getControlSeq()

    def nominal_state0(self):
        return (self.e.getNominalStateSeq() if self.kind == "tube" else self.e.getTargetStateSeq())[0]

    def slide(self):
        self.e.slideControlSequence(1)

    def baseline(self):
        st = self.e.getStats()
        return float(st.real_sys.baseline)

    def nominal_used(self):
        return self.e.getStats().nominal_state_used

    def best_index(self):
        return self.e.getRMPPIState()[1]

    def step(self, x, u):
        return self.e.modelStep(x, u)[0]

    def close(self):
        self.e.close()


def swingup_loop(side, steps=1000, trace=None):
    """vanilla_mppi_test.cu:109-135"""
    x = np.zeros(4, np.float32)
    for i in range(steps):
        side.compute(x)
        u = side.control()
        x = side.step(x, u[0])  # model.computeStateDeriv + updateState (:126-129)
        side.slide()
        if trace is not None:
            trace.append((x.copy(), u[0].copy(), side.baseline()))
    return side.baseline(), x


def di_loop(side, kind, steps, variance, plant_seed, gains=None, trace=None):
    """tube_mppi_test.cu:166-204 (vanilla), :402-487 (tube); rmppi_test.cu:636-690, 800-900 (robust).
    Returns (step of the first tube failure or None, final state, set of nominal-state flags / best candidate indices)"""
    rng = np.random.default_rng(plant_seed)
    x = np.array([2, 0, 0, 1], np.float32)
    L, G = gains if gains is not None else (None, None)
    seen = set()
    if kind == "robust":
        side.set_gains(G)  # computeNominalFeedbackGains at the end of every update (robust_mppi_controller.cu:567): constant here
    for t in range(steps):
        if tube_failure(x):
            return t, x, seen
        if kind == "robust":
            side.update(x)
        side.compute(x)
        u = side.control()[0].copy()
        if kind == "tube":
            seen.add(side.nominal_used())
        if kind == "robust":
            seen.add(side.best_index())
        if L is not None:  # current_control += getFeedbackControl(x, getTargetStateSeq().col(0), 0)
            u = (u + L[0] @ (x - side.nominal_state0())).astype(np.float32)
        x = di_plant(x, u, rng, variance)
        if kind != "robust":  # RobustMPPIController::slideControlSequence is empty (robust_mppi_controller.cuh:190)
            side.slide()
        if trace is not None:
            trace.append((x.copy(), u.copy()))
    return None, x, seen


GAINS = ddp_gains_linear(50, 0.02, np.diag([500.0, 500.0, 100.0, 100.0]), np.eye(4), np.eye(2))


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle passes the reference's acceptance thresholds
# ---------------------------------------------------------------------------------------------------------------------
def test_ddp_gains_linear_is_the_riccati_recursion():
    """the restated backward pass against the textbook discrete-time LQR recursion on the same (Phi, B dt, Q dt, R dt)"""
    L, G = GAINS
    Phi = np.eye(4)
    Phi[0, 2] = Phi[1, 3] = 0.02
    Bd = np.zeros((4, 2))
    Bd[2, 0] = Bd[3, 1] = 0.02
    Q, R = np.diag([500.0, 500.0, 100.0, 100.0]) * 0.02, np.eye(2) * 0.02
    P = np.eye(4)
    for k in range(48, -1, -1):
        Kk = np.linalg.solve(R + Bd.T @ P @ Bd, Bd.T @ P @ Phi)
        P = Q + Phi.T @ P @ (Phi - Bd @ Kk)
        assert np.allclose(L[k], -Kk, rtol=1e-5, atol=1e-6), k
    assert not L[49].any()
    assert G.shape == (50, 4, 2) and G[3, 2, 1] == L[3, 1, 2]
    # a stabilising gain: position and velocity errors are pushed back
    assert L[0, 0, 0] < 0 and L[0, 0, 2] < 0 and abs(L[0, 0, 1]) < 1e-6


def test_swingup_oracle():
    """SwingUpTest on the restatement, the product's Philox stream: EXPECT_LT(getBaselineCost(), 1.0) after 1000 steps"""
    base, x = swingup_loop(_OracleSide(swingup_cfg(), 42, "vanilla"))
    assert base < 1.0, base
    assert abs(x[0] + 20.0) < 0.1 and abs(x[2] - np.pi) < 0.05, x  # at the goal, pole up


@pytest.mark.parametrize("seed", [1, 2])
def test_di_vanilla_nominal_variance_oracle(seed):
    fail, x, _ = di_loop(_OracleSide(di_acceptance_cfg(1, 4.0, 3), seed, "vanilla"), "vanilla", 500, 1.0, seed)
    assert fail is None, (fail, x)


@pytest.mark.parametrize("seed", [1])
def test_di_tube_large_variance_oracle(seed):
    side = _OracleSide(di_acceptance_cfg(2, 4.0, 3), seed, "tube", thr=100.0)
    fail, x, used = di_loop(side, "tube", 500, 100.0, seed, gains=GAINS)
    assert fail is None, (fail, x)
    assert used == {0, 1}, used  # the disturbance is large enough for the nominal system to take over now and then


def test_di_robust_large_variance_oracle_1000_steps():
    """the first 1000 of the reference's 5000 steps on the CPU (the gpu test runs all 5000 on the engine)"""
    side = _OracleSide(di_acceptance_cfg(2, 4.0, 1, sampler_cost=0.0), 1, "robust", thr=10.0)
    fail, x, best = di_loop(side, "robust", 1000, 100.0, 1, gains=GAINS)
    assert fail is None, (fail, x)
    assert len(best) > 2, best  # the line search moves the nominal state


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the engine passes them with its own in-kernel draws, and equals the oracle's loop bit for bit in reference order
# ---------------------------------------------------------------------------------------------------------------------
def _compare_traces(a, b, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for i, (ra, rb) in enumerate(zip(a, b)):
        for j, (va, vb) in enumerate(zip(ra, rb)):
            assert np.array_equal(_bits(va), _bits(vb)), "%s: step %d, item %d: %r != %r" % (what, i, j, va, vb)


@pytest.mark.gpu
def test_swingup_engine(gpu):
    cfg = swingup_cfg()
    eng = _EngineSide(cfg, 42, "vanilla")
    base, x = swingup_loop(eng)
    eng.close()
    assert base < 1.0, base
    assert abs(x[0] + 20.0) < 0.1 and abs(x[2] - np.pi) < 0.05, x
    # RNG included: the reference-order handle and the oracle on the same Philox stream, all 1000 steps, bit for bit
    te, to = [], []
    ex = _EngineSide(cfg, 42, "vanilla", reference_order=True)
    be, _ = swingup_loop(ex, trace=te)
    ex.close()
    bo, _ = swingup_loop(_OracleSide(cfg, 42, "vanilla"), trace=to)
    _compare_traces(te, to, "SwingUpTest")
    assert be == bo and be < 1.0


@pytest.mark.gpu
def test_di_vanilla_nominal_variance_engine(gpu):
    cfg = di_acceptance_cfg(1, 4.0, 3)
    for seed in (1, 2, 3):
        eng = _EngineSide(cfg, seed, "vanilla")
        fail, x, _ = di_loop(eng, "vanilla", 500, 1.0, seed)
        eng.close()
        assert fail is None, (seed, fail, x)
    te, to = [], []
    ex = _EngineSide(cfg, 1, "vanilla", reference_order=True)
    assert di_loop(ex, "vanilla", 500, 1.0, 1, trace=te)[0] is None
    ex.close()
    assert di_loop(_OracleSide(cfg, 1, "vanilla"), "vanilla", 500, 1.0, 1, trace=to)[0] is None
    _compare_traces(te, to, "VanillaMPPINominalVariance")


@pytest.mark.gpu
def test_di_tube_large_variance_engine(gpu):
    cfg = di_acceptance_cfg(2, 4.0, 3)
    for seed in (1, 2, 3):
        eng = _EngineSide(cfg, seed, "tube", thr=100.0)
        fail, x, used = di_loop(eng, "tube", 500, 100.0, seed, gains=GAINS)
        eng.close()
        assert fail is None, (seed, fail, x)
        assert used == {0, 1}, used
    te, to = [], []
    ex = _EngineSide(cfg, 1, "tube", thr=100.0, reference_order=True)
    fe, _, ue = di_loop(ex, "tube", 500, 100.0, 1, gains=GAINS, trace=te)
    ex.close()
    fo, _, uo = di_loop(_OracleSide(cfg, 1, "tube", thr=100.0), "tube", 500, 100.0, 1, gains=GAINS, trace=to)
    assert fe is None and fo is None and ue == uo
    _compare_traces(te, to, "TubeMPPILargeVariance")


@pytest.mark.gpu
@pytest.mark.parametrize("robust_cost", [False, True], ids=["circle_cost", "robust_cost"])
def test_di_robust_large_variance_engine(gpu, robust_cost):
    cfg = di_acceptance_cfg(2, 2.0 if robust_cost else 4.0, 3 if robust_cost else 1, robust_cost=robust_cost, sampler_cost=0.0)
    eng = _EngineSide(cfg, 1, "robust", thr=10.0)
    fail, x, best = di_loop(eng, "robust", 5000, 100.0, 1, gains=GAINS)  # the reference's total_time_horizon
    eng.close()
    assert fail is None, (fail, x)
    assert len(best) > 2, best
    te, to = [], []
    n = 600
    ex = _EngineSide(cfg, 1, "robust", thr=10.0, reference_order=True)
    fe, _, be = di_loop(ex, "robust", n, 100.0, 1, gains=GAINS, trace=te)
    ex.close()
    fo, _, bo = di_loop(_OracleSide(cfg, 1, "robust", thr=10.0), "robust", n, 100.0, 1, gains=GAINS, trace=to)
    assert fe is None and fo is None and be == bo
    _compare_traces(te, to, "RobustMPPILargeVariance")
