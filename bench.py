#!/usr/bin/env python3
"""bench.py — MPPI iterations/s of the HIP rollout-and-reduce engine (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the optimisation-loop body (sample -> K x T rollout -> baseline/normExp -> weighted reduction,
mean <- u*) with x0 and the control mean already resident in HBM.  Workload at every N: Cartpole
(CartpoleDynamics + CartpoleQuadraticCost), K = 16384 rollouts PER GPU, T = 100, fp32, Philox noise drawn in the kernel.
N > 1 defaults to weak scaling: rank r owns rollouts [r*K, (r+1)*K) of a K*N-rollout problem and the ranks exchange one
(T*C+4)-float record per iteration (RCCL all-gather) — value counts K-rollout iteration units over all ranks.
--scaling strong splits the SAME K = 16384 problem over the ranks (value = iterations/s of that one problem); --workload
autorally runs AutoRally-NN K = 16384, T = 150 as the primary workload instead (7x the work per rollout: the configuration
where sharding a fixed problem can pay — DESIGN.md §6).

The timed region is EXACTLY --steps iterations between barrier + synchronize; because 20 Cartpole steps are 0.7 ms, the region
is repeated (each repetition bracketed the same way) until at least --min-time seconds have been timed in total and
ms_per_step is the median repetition — the driver's --steps 20 and a 2000-step run then agree.

roofline: algorithmic bytes of the dominant kernel (rolloutKernel) per launch, B_alg = 4*(2*K*T*C + 2*K + 2*T*C)
(SURVEY.md §8d), divided by its average duration measured with HIP events on the engine's own stream.
cpu_baseline: the CPU oracle (a port of the reference's CPU path) timed on this box's host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

K_PER_GPU = 16384
T = 100
HBM_PEAK_GBS = 8000.0


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def _latest_pmc_file():
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_hbm_traffic.json")))
    return os.path.relpath(files[-1], here) if files else os.path.join("profiles", "none.json")


PMC_FILE = _latest_pmc_file()


def pmc_traffic(kernel_prefix, suffix=""):
    """HBM bytes per launch of the kernel whose (shortened) name starts with kernel_prefix (and ends with suffix: the tail of the
    template arguments tells instantiations of one kernel apart), from the committed PMC passes (tools/profile_bench.sh +
    tools/pmc_summary.py -> profiles/), or None"""
    try:
        d = json.load(open(os.path.join(REPO, PMC_FILE)))
        for name, e in d["kernels"].items():
            if name.startswith(kernel_prefix) and name.endswith(suffix) and e.get("hbm_traffic_bytes_per_launch"):
                return round(e["hbm_traffic_bytes_per_launch"], 1)
    except Exception:  # noqa: BLE001
        pass
    return None


def pmc_pipe_util(kernel_prefix, suffix=""):
    """{"mfma_util_percent", "valu_busy_percent"} of the kernel from the committed SQ counter pass
    (tools/profile_bench.sh + tools/pmc_mfma_summary.py -> profiles/r*_pmc_mfma_valu.json), or None"""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_mfma_valu.json")))
        d = json.load(open(files[-1]))
        for name, e in d["kernels"].items():
            if name.startswith(kernel_prefix) and name.endswith(suffix):
                return {"mfma_util_percent": round(e.get("MfmaUtil_percent", 0.0), 2),
                        "valu_busy_percent": round(e.get("VALUBusy_percent", 0.0), 2),
                        "source": os.path.relpath(files[-1], REPO) + " (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, "
                                  "GRBM_GUI_ACTIVE)"}
    except Exception:  # noqa: BLE001
        pass
    return None


def cpu_baseline(cfg, budget_s=12.0, label=None, min_iters=1, one_core_k=None, noise_note=""):
    """oracle iterations/s on the host cores, bounded sample (never the thing shipped or measured as `value`); any of the
    Vanilla / Tube configurations (cfg["D"] systems per iteration).

    SURVEY.md §8d's rule — at least 5 warm-up and 50 timed iterations, the MEDIAN — inside the bounded sample the contract
    asks for (budget_s of CPU work): where 55 iterations of the full problem do not fit the budget, the sample is a SLICE of
    K / 2^j rollouts (rollouts are independent and spread over the threads either way: the time of an iteration is
    proportional to K) and the rate is scaled back by the slice; the line says so.  The single-thread figure (the reference's
    CPU path is single-threaded) is taken the same way with a smaller count (>= 1 warm-up + >= 5 timed, median).
    min_iters / one_core_k: accepted for the callers of earlier rounds, superseded by the rule above."""
    import numpy as np
    import pyoracle as po
    from common import make_oracle
    K, Tn, D = cfg["K"], cfg["T"], cfg["D"]
    threads = max(1, min(po.max_threads(), usable_cores()))

    def sample(th, warm, timed, budget):
        k_used = K
        while True:
            c = dict(cfg, K=k_used)
            o = make_oracle(c)
            eps = po.philox_normal(42, 0, k_used, Tn, o.C)
            mean = np.zeros((D, Tn, o.C), np.float32)
            x0 = np.tile(np.asarray(cfg["x0"], np.float32), (D, 1))
            t1 = o.time_iterations(x0, mean, eps, 1, th)  # first warm-up iteration: also the estimate the slice comes from
            if (warm + timed) * t1 <= budget or k_used <= max(256, 64 * th) or k_used % 2:
                break
            del o
            k_used //= 2
        for _ in range(warm - 1):
            o.time_iterations(x0, mean, eps, 1, th)
        ts = [o.time_iterations(x0, mean, eps, 1, th) for _ in range(timed)]
        del o
        med = float(np.median(ts))
        return (1.0 / med) * (k_used / float(K)), k_used, med, float(min(ts)), float(max(ts))

    v, k_all, med, lo, hi = sample(threads, 5, 50, budget_s * 0.75)
    v1, k_one, med1, _, _ = sample(1, 1, 5, budget_s * 0.25)
    what = label or ("Cartpole K=%d T=%d" % (K, Tn))

    def slice_text(k_used):
        return "" if k_used == K else " on a slice of %d of the %d rollouts (rate scaled by 1/%d)" % (k_used, K, K // k_used)
    return {
        "value": round(v, 3), "unit": "MPPI iters/s", "cores": threads, "kind": "port",
        "sample": "5 warm-up + 50 timed iterations, median (%.4f s; min %.4f, max %.4f)%s: %s, one optimisation-loop body each, "
                  "the rollouts spread over %d OpenMP threads%s" % (med, lo, hi, slice_text(k_all), what, threads, noise_note),
        "value_1core": round(v1, 4),
        "sample_1core": "1 warm-up + 5 timed iterations, median (%.4f s), single thread (the reference's CPU path is "
                        "single-threaded)%s" % (med1, slice_text(k_one)),
    }


def launches_per_iteration(eng, n=100):
    """kernel launches one optimisation iteration costs on this handle, counted by the library (mppi_get_launch_counts): 2
    (rollout + merge) or — where the rollout kernel merges the previous records in its sampler waves — 1"""
    r0, g0 = eng.launchCounts()
    eng.optimize(n, True)
    r1, g1 = eng.launchCounts()
    return int(round((r1 - r0 + g1 - g0) / float(n)))


def latency_model(device, iteration_us, n_launch=2):
    """What bounds an iteration of this design at K=16384 (t_floor ~ T * t_step + n_launch * t_launch), measured
    live: the rollout kernel against T (slope = time of one dependent rollout step on the dynamics wave, intercept = launch
    ramp + prologue + block softmin epilogue + one launch boundary, since the launches are timed back to back), and the
    cost of a dependent kernel boundary on an idle stream (trivial kernels)."""
    import numpy as np
    import mppi_generic_amd as m
    from common import cartpole_cfg, make_engine
    # MPPI_BENCH_NO_TSCAN=1 (tools/profile_bench.sh sets it for the traced runs): only T = 100 is launched, so that the kernel
    # trace's row for rolloutPipelineKernel<Cartpole> holds nothing but the headline configuration
    if os.environ.get("MPPI_BENCH_NO_TSCAN"):
        return {"skipped": "MPPI_BENCH_NO_TSCAN set (traced run: the T scan shares the headline kernel's name)"}
    ts, us = [24, 52, 100, 200], []  # multiples of 4: every T runs the kernel form the headline runs (streamed merge)
    for tt in ts:
        e = make_engine(cartpole_cfg(K=K_PER_GPU, T=tt), device=device)
        e.uploadState(np.zeros(4, np.float32))
        e.optimize(20, True)
        _, ms_roll = e.timeIterations(100)
        us.append(ms_roll / 100 * 1e3)
        e.close()
    slope, intercept = np.polyfit(np.asarray(ts, np.float64), np.asarray(us, np.float64), 1)
    boundary = m.launch_boundary_us(device, 400)
    # rollout kernel (its own boundary is in the intercept) + the boundary of every further launch of an iteration
    floor = slope * T + intercept + (n_launch - 1) * boundary
    return {"t_scan_T": ts, "t_scan_kernel_us": [round(u, 2) for u in us], "step_ns": round(slope * 1e3, 1),
            "fixed_us": round(intercept, 2), "launch_boundary_us": round(boundary, 2), "n_launch": n_launch,
            "latency_floor_us": round(floor, 2), "iteration_us": round(iteration_us, 2),
            "fit_residual": round(1.0 - floor / iteration_us, 4),
            "definition": "latency_floor_us = step_ns * T + fixed_us + (n_launch - 1) * launch_boundary_us: the rollout kernel as "
                          "it is (n_launch = 1: it merges the previous iteration's records in its sampler waves, and the one "
                          "merge launch at the end of a loop is spread over its iterations) plus the bare boundary of a "
                          "separate merge launch where there is one.  fit_residual = 1 - latency_floor_us / iteration_us: how "
                          "well a line through the kernel's OWN timings reproduces the iteration — a consistency check of the "
                          "measurement, not a distance to any bound (issue_floor is the floor that does not depend on the "
                          "kernel's timing)"}


def isa_step_counts():
    """instruction counts per rollout step of the dynamics wave, from the ISA of the built kernels: static, produced on the
    build machine by tools/isa_step_count.py (tests/test_abi.py checks the file against the current sources)"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_isa_step_counts.json")))
    return (json.load(open(files[-1])), os.path.relpath(files[-1], REPO)) if files else ({}, None)


def issue_floor(device, iteration_us, rollout_kernel_us, key="cartpole_pipeline_dynamics_wave", t_steps=T, n_launch=2):
    """A floor for the iteration that does NOT come from timing the kernel being judged (round-2 review, weak 7):
    floor = (instructions per step on the dynamics wave, from the ISA) x (issue interval of a lone wave, measured live with an
    independent micro-kernel) x T + n_launch x (kernel boundary, measured live with trivial kernels).  `all` counts every
    instruction of the step loop (VALU, SALU, LDS, waits, nops: each takes an issue slot of the one wave a SIMD holds),
    `vector_only` only the v_* instructions — what would remain if every scalar / wait / nop were free."""
    import mppi_generic_amd as m
    counts, src = isa_step_counts()
    c = counts.get(key, {})
    if "instructions_per_step" not in c:
        return {"error": "no ISA step count for " + key}
    issue_ns = m.issue_interval_ns(device)
    boundary = m.launch_boundary_us(device, 400)
    out = {"instructions_per_step_static_from_isa": c["instructions_per_step"],
           "vector_instructions_per_step_static_from_isa": c["vector_instructions_per_step"], "isa_source": src,
           "issue_interval_ns_measured": round(issue_ns, 3), "launch_boundary_us_measured": round(boundary, 3),
           "n_launch": n_launch, "T": t_steps}
    for name, n in (("all", c["instructions_per_step"]), ("vector_only", c["vector_instructions_per_step"])):
        floor = n * issue_ns * 1e-3 * t_steps + n_launch * boundary
        out["floor_us_" + name] = round(floor, 2)
        out["frac_of_floor_" + name] = round(floor / iteration_us, 4)
    out["iteration_us"] = round(iteration_us, 2)
    out["rollout_kernel_us"] = round(rollout_kernel_us, 2)
    out["definition"] = ("floor_us = instructions_per_step x issue_interval_ns x T + n_launch x launch_boundary_us; "
                         "frac_of_floor = floor / measured iteration (1.0 = nothing left but the step loop and the boundaries)")
    return out


def compute_control_latency(device, x0):
    """what a control loop sees (wall time per compute_control): inputs handed over from the host, one
    iteration, smoothing + constraints + re-rollout of u*, results back — PCIe-inclusive, never `value`"""
    from common import cartpole_cfg, make_engine
    eng = make_engine(cartpole_cfg(K=K_PER_GPU, T=T), device=device)
    for _ in range(50):
        eng.computeControl(x0, 1)
    n = 300
    ready = 0.0
    for _ in range(n):
        eng.getTargetStateSeq()  # the previous call's trajectory has landed: the stream is idle
        t_a = time.perf_counter()
        eng.computeControl(x0, 1)
        ready += time.perf_counter() - t_a
    t_a = time.perf_counter()
    for _ in range(n):
        eng.computeControl(x0, 1)
        eng.getTargetStateSeq()
    full = time.perf_counter() - t_a
    t_a = time.perf_counter()
    for _ in range(n):
        eng.computeControl(x0, 1)
        eng.getControlSeq()
        eng.slideControlSequence(1)
    loop = time.perf_counter() - t_a
    eng.close()
    return {"control_ready_us": round(ready / n * 1e6, 2), "control_and_state_trajectory_us": round(full / n * 1e6, 2),
            "closed_loop_period_us": round(loop / n * 1e6, 2),
            "definition": "control_ready: mppi_compute_control from an idle stream until the control sequence is on the host "
                          "(the call returns then; the finalize kernel is still re-rolling the state trajectory); "
                          "control_and_state_trajectory: + mppi_get_state_seq; closed_loop_period: computeControl + "
                          "getControlSeq + slide back to back (split hand-over: the trajectory re-rollout of one call runs on a "
                          "side stream beside the next call's rollouts, DESIGN.md §2)"}


def kernel_stats_us(kernel_prefix):
    """average duration (us) of the kernel whose name starts with kernel_prefix in the latest committed rocprofv3 --stats summary
    (profiles/r*_kernel_stats.csv), with the file it came from — the static stand-in where a leg cannot time its kernel live"""
    import csv
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*kernel_stats.csv")), reverse=True):
        try:  # tools/stats_summary.py's format: comment lines, then calls,total_us,avg_us,...,kernel
            lines = [ln for ln in open(f) if not ln.startswith("#")]
            for row in csv.DictReader(lines):
                if kernel_prefix in (row.get("kernel") or "") and float(row.get("avg_us") or 0.0) > 0:
                    return float(row["avg_us"]), os.path.relpath(f, REPO)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def rollout_kernel_us(eng, n, kernel_prefix):
    """(us per launch of the handle's rollout kernel, how it was obtained): HIP events on the engine's own stream around n
    back-to-back launches (mppi_time_iterations), or the committed kernel-trace average when the handle cannot be timed so"""
    try:
        _, ms_roll = eng.timeIterations(n)
        if ms_roll > 0:
            return ms_roll / n * 1e3, "HIP events around %d back-to-back launches on the engine's stream (mppi_time_iterations)" % n
    except Exception as e:  # noqa: BLE001
        why = str(e)[:120]
    else:
        why = "no rollout pass"
    us, src = kernel_stats_us(kernel_prefix)
    return us, "static: rocprofv3 --kernel-trace --stats average from %s (live timing unavailable: %s)" % (src, why)


def robust_roofline(eng, n, kernel_prefix, flops_per_launch, b_alg, peak_note):
    """MFMA (NN model) or HBM (analytic model) roofline of a Robust MPPI rollout launch: two systems per rollout"""
    us, how = rollout_kernel_us(eng, n, kernel_prefix)
    traffic = pmc_traffic(kernel_prefix)
    if us is None:
        return {"error": "kernel could not be timed", "how": how}
    if flops_per_launch:
        ach = flops_per_launch / (us * 1e-6) / 1e12
        r = {"bound": "mfma", "achieved": round(ach, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 5),
             "algorithmic_flops_per_launch": flops_per_launch}
    else:
        ach = b_alg / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
    r.update({"kernel": kernel_prefix + "...>", "traffic": traffic, "traffic_static_from_profiles": PMC_FILE,
              "algorithmic_bytes_per_launch": b_alg,
              "traffic_over_algorithmic_bytes": round(traffic / b_alg, 3) if traffic else None,
              "avg_kernel_us": round(us, 3), "avg_kernel_us_source": how, "note": peak_note})
    return r


def robust_autorally_leg(device):
    """Robust MPPI (two coupled systems per rollout, DDP feedback term) on the AutoRally NeuralNetModel, K=16384, T=150: what a
    control loop sees per call.  The rollout runs as the role-pipelined kernel of engine/rmppi_pipeline_kernel.hpp."""
    import numpy as np
    import mppi_generic_amd as m
    from common import autorally_cfg
    K, Tn = 16384, 150
    cfg = autorally_cfg(K=K, T=Tn, lambda_=1.0)
    eng = m.RobustMPPIController(cfg["model"], K, Tn, cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42, device=device)
    eng.setCostParams(cfg["cost"])
    for name, blob in cfg["blobs"].items():
        eng.setModelBlob(name, blob)
    eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], [0.2, 0.1])
    eng.setRMPPIParams(500.0, 9, 32)
    g = np.random.default_rng(5).uniform(-0.3, 0.3, (Tn, 7, 2)).astype(np.float32)
    x = cfg["x0"].copy()

    def cycle():
        eng.updateImportanceSamplingControl(x, 1)
        eng.setFeedbackGains(g)
        eng.computeControl(x, 1)
    for _ in range(10):
        cycle()
    n = 50
    t_a = time.perf_counter()
    for _ in range(n):
        cycle()
    period = (time.perf_counter() - t_a) / n
    ready = 0.0
    for _ in range(n):
        eng.getTargetStateSeq()  # the previous call's trajectories have landed: the stream is idle
        t_a = time.perf_counter()
        eng.computeControl(x, 1)
        ready += time.perf_counter() - t_a
    u = eng.getControlSeq()
    # two systems per rollout: twice the Vanilla row's flops; B_alg with D = 2
    roof = robust_roofline(eng, 20, "rolloutRMPPIPipelineKernel<NeuralNetModelMFMA", 2 * 2.0 * (6 * 32 + 32 * 32 + 32 * 4) * K * Tn,
                           4.0 * 2 * (2.0 * K * Tn * 2 + 2.0 * K + 2.0 * Tn * 2),
                           "the one hot kernel whose sample rows leave the CU: `traffic` above the algorithmic bytes is spill scratch "
                           "(a 960-thread block leaves 128 VGPRs per wave; SGPR spills of the five plugin objects' kernel "
                           "arguments), not the write-back pattern — the double integrator's instantiation writes 1.06x its "
                           "rows (DESIGN.md §0 item 6)")
    eng.close()
    return {"workload": "RobustMPPI (nominal + real system, DDP gains [T][7][2], 9 x 32 candidate rollouts), AutoRally "
                        "NeuralNetModel<7,2,3> + ARStandardCost, K=16384, T=150",
            "roofline": roof,
            "control_ready_us": round(ready / n * 1e6, 2), "cycle_us": round(period * 1e6, 2), "finite": bool(np.isfinite(u).all()),
            "kernel": "rolloutRMPPIPipelineKernel<NeuralNetModelMFMA<7,2,3>,ARStandardCost,DeviceDDP,Gaussian>: 8 dynamics + 1 sampler "
                      "+ 6 cost waves per 64 rollouts x 2 systems (profiles/r03_b_robust_kernel_stats.csv)",
            "definition": "control_ready: mppi_compute_control from an idle stream until both control sequences and the statistics "
                          "are on the host; cycle: updateImportanceSamplingControl (candidate evaluation, slide, nominal "
                          "trajectory) + setFeedbackGains + computeControl back to back"}


def robust_di_leg(device):
    """Robust MPPI on the double integrator (the reference's example: examples/double_integrator_CORL2020.cu), K=8192, T=150"""
    import numpy as np
    import mppi_generic_amd as m
    from common import di_cfg
    cfg = di_cfg(K=8192, T=150, tube=True, num_iters=1)
    eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42, device=device)
    eng.setDynamicsParams(cfg["dyn"])
    eng.setCostParams(cfg["cost"])
    eng.setSamplingParams(cfg["std_dev"], [0.3, 0.2])
    eng.setRMPPIParams(25.0, 9, 32)
    g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], 4, 2)).astype(np.float32)
    x = cfg["x0"].copy()

    def cycle():
        eng.updateImportanceSamplingControl(x, 1)
        eng.setFeedbackGains(g)
        eng.computeControl(x, 1)
    for _ in range(20):
        cycle()
    n = 200
    t_a = time.perf_counter()
    for _ in range(n):
        cycle()
    period = (time.perf_counter() - t_a) / n
    ready = 0.0
    for _ in range(n):
        eng.getTargetStateSeq()
        t_a = time.perf_counter()
        eng.computeControl(x, 1)
        ready += time.perf_counter() - t_a
    u = eng.getControlSeq()
    roof = robust_roofline(eng, 50, "rolloutRMPPIPipelineKernel<DoubleIntegratorDynamics", 0.0,
                           4.0 * 2 * (2.0 * cfg["K"] * cfg["T"] * 2 + 2.0 * cfg["K"] + 2.0 * cfg["T"] * 2),
                           "HBM is the ceiling SURVEY.md §8d assigns; the kernel is bound by the issue rate of its dynamics waves")
    eng.close()
    return {"workload": "RobustMPPI, DoubleIntegrator + DoubleIntegratorCircleCost, K=8192, T=150, 9 x 32 candidate rollouts",
            "roofline": roof,
            "control_ready_us": round(ready / n * 1e6, 2), "cycle_us": round(period * 1e6, 2), "finite": bool(np.isfinite(u).all()),
            "kernel": "rolloutRMPPIPipelineKernel<DoubleIntegratorDynamics,...>: 2 dynamics + 2 sampler + 6 cost waves per 64 rollouts "
                      "x 2 systems"}


def robust_racer_leg(device):
    """Robust MPPI on the elevation-map RACER models, K=16384, T=100: the role-pipelined rollout kernel per launch (HIP events)
    — the kernels whose dynamics waves are short of REGISTERS (profiles/r06_robust_racer_ab.json)"""
    import numpy as np
    import mppi_generic_amd as m
    from test_racer_dubins_lstm_unc import uncertainty_cfg
    from test_racer_dubins_suspension import suspension_cfg
    out = {"workload": "RobustMPPI (nominal + real system) on RacerDubinsElevationSuspension and on the complete RACER model "
                       "(RacerDubinsElevationLSTMUncertainty), QuadraticCost, K=16384, T=100, 9 x 32 candidate rollouts; "
                       "rollout kernel per launch"}
    for name, mk in (("suspension", suspension_cfg), ("complete_model", uncertainty_cfg)):
        cfg = mk(K=16384, T=100, D=2)
        eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42, device=device)
        eng.setDynamicsParams(cfg["dyn"])
        eng.setCostParams(cfg["cost"])
        for bname, blob in cfg["blobs"].items():
            eng.setModelBlob(bname, blob)
        if cfg["ranges"] is not None:
            eng.setControlRanges(cfg["ranges"])
        eng.setSamplingParams(cfg["std_dev"], [0.2, 0.1])
        eng.setRMPPIParams(2000.0, 9, 32)
        g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], eng.STATE_DIM, eng.CONTROL_DIM)).astype(np.float32)
        x = cfg["x0"].copy()
        for _ in range(3):
            eng.updateImportanceSamplingControl(x, 1)
            eng.setFeedbackGains(g)
            eng.computeControl(x, 1)
        best = None
        for _ in range(3):
            _, roll = eng.timeIterations(10)
            best = roll if best is None else min(best, roll)
        out[name] = {"rollout_kernel_us": round(best / 10 * 1e3, 1), "finite": bool(np.isfinite(eng.getControlSeq()).all())}
        eng.close()
    return out


def autorally_leg(device, with_cpu_baseline=True):
    """AutoRally NeuralNetModel (FNN 6-32-32-4, synthetic weights) + ARStandardCost, K=16384, T=150, one GPU:
    iterations/s and the MFMA roofline of the NN forward (F_alg = 2 * sum(MAC) * K * T)."""
    from common import autorally_cfg, make_engine
    K, Tn = 16384, 150
    cfg = autorally_cfg(K=K, T=Tn, lambda_=1.0)
    eng = make_engine(cfg, device=device)
    eng.uploadState(cfg["x0"])
    eng.optimize(20, True)
    # 400 iterations per timed call, best of three: a call's fixed part (first enqueue, the wake-up after the synchronisation —
    # ~0.6 ms here) was 3 % of a 100-iteration call
    n = 400
    wall = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.optimize(n, True)
        wall = min(wall, time.perf_counter() - t0)
    ms_total, ms_roll = eng.timeIterations(50)
    roll_us = ms_roll / 50 * 1e3
    f_alg = 2.0 * (6 * 32 + 32 * 32 + 32 * 4) * K * Tn
    achieved = f_alg / (roll_us * 1e-6) / 1e12
    try:
        floor = issue_floor(device, ms_total / 50 * 1e3, roll_us, "autorally_mfma_pipeline_dynamics_wave", Tn)
    except Exception as e:  # noqa: BLE001
        floor = {"error": str(e)}
    eng.close()
    cpu = None
    if with_cpu_baseline:  # BASELINE.md §2 lists this configuration among the CPU-timed ones
        try:
            cpu = cpu_baseline(cfg, budget_s=8.0, label="AutoRally-NN K=%d T=%d" % (K, Tn))
        except Exception as e:  # noqa: BLE001
            cpu = {"error": str(e)}
    return {
        "workload": "AutoRally NeuralNetModel<7,2,3> (FNN 6-32-32-4, synthetic weights) + ARStandardCost (600x600 "
                    "generated track map), VanillaMPPI iteration, K=16384, T=150, block (64 rollouts x 4 MFMA lanes)",
        "value": round(n / wall, 3), "unit": "MPPI iters/s", "ms_per_step": round(wall / n * 1e3, 6),
        "cpu_baseline": cpu,
        "roofline": {"bound": "mfma", "kernel": "rolloutPipelineRepKernel<NeuralNetModelMFMA<7,2,3>,ARStandardCost,Gaussian,true>",
                     "achieved": round(achieved, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 5),
                     "traffic": pmc_traffic("rolloutPipelineRepKernel<NeuralNetModelMFMA"),
                     "pipe_utilisation_static_from_profiles": pmc_pipe_util("rolloutPipelineRepKernel<NeuralNetModelMFMA"),
                     "algorithmic_flops_per_launch": f_alg, "avg_kernel_us": round(roll_us, 3), "issue_floor": floor,
                     "note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32) peak = the fp32 vector peak; the kernel is bound by VALU "
                             "issue, not by the matrix cores: per 16 rollouts and step a dynamics wave issues 28 MFMAs and ~500 "
                             "VALU instructions (64 packed-fp32 tanh per rollout, kinematics, Euler); sampler and cost run "
                             "once per rollout in helper waves (two samplers, two relaying cost waves: one helper per SIMD); K=16384 is exactly one dynamics wave per SIMD"},
    }


def lstm_colored_leg(device, with_cpu_baseline=True):
    """BASELINE config 5: LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights) + ARStandardCost +
    ColoredNoise sampler (exponents 1, offset decay 0.97), ColoredMPPI iteration, K=65536, T=200, one GPU.
    MFMA roofline: F_alg = 2*(4H(I+H) + (H+I)*M + M*OUT)*K*T for the network; the colored-noise transform (radix-4 butterfly +
    a quarter-size GEMM, 2*(2T+2)*T*C*K/4 flops) is reported beside it, not counted."""
    from common import bicycle_lstm_cfg, make_engine
    K, Tn = 65536, 200
    cfg = bicycle_lstm_cfg(K=K, T=Tn, lambda_=1.0)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng = make_engine(cfg, device=device)
    eng.uploadState(cfg["x0"])
    eng.optimize(5, True)
    n = 60
    wall = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        eng.optimize(n, True)
        wall = min(wall, time.perf_counter() - t0)
    ms_total, ms_roll = eng.timeIterations(20)
    roll_us = ms_roll / 20 * 1e3
    f_net = 2.0 * (4 * 16 * (6 + 16) + (16 + 6) * 32 + 32 * 4) * K * Tn
    f_noise = 2.0 * (2 * Tn + 2) * Tn * 2 * K / 4.0  # T = 4P: two decimation steps in front of the GEMM leave a quarter of it
    achieved = f_net / (roll_us * 1e-6) / 1e12              # the MFMA roofline is for the NN forward only
    achieved_with_gemm = (f_net + f_noise) / (roll_us * 1e-6) / 1e12
    eng.close()
    cpu = None
    if with_cpu_baseline:  # at least 3 CPU iterations of config 5
        try:
            cpu = cpu_baseline(dict(cfg, colored=None), budget_s=6.0, label="LSTM bicycle-slip K=%d T=%d" % (K, Tn), min_iters=3,
                               one_core_k=K // 16,
                               noise_note=("; time-domain noise given — the reference's CPU rollout takes its noise from the "
                                           "device sampler (rollout_kernel_test.cu:516-517), so the colored-noise transform is "
                                           "not part of the CPU path"))
        except Exception as e:  # noqa: BLE001
            cpu = {"error": str(e)}
    return {
        "cpu_baseline": cpu,
        "workload": "LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights) + ARStandardCost + ColoredNoise "
                    "sampler (exponents [1,1], offset_decay_rate 0.97), ColoredMPPI iteration, K=65536, T=200, block (64 "
                    "rollouts x 4 MFMA lanes)",
        "value": round(n / wall, 3), "unit": "MPPI iters/s", "ms_per_step": round(wall / n * 1e3, 6),
        "roofline": {"bound": "mfma", "kernel": "rolloutPipelineRepKernel<BicycleSlipLSTMMFMA,ARStandardCost,ColoredNoise,false>",
                     "achieved": round(achieved, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 5),
                     "traffic": pmc_traffic("rolloutPipelineRepKernel<BicycleSlipLSTMMFMA"),
                     "pipe_utilisation_static_from_profiles": pmc_pipe_util("rolloutPipelineRepKernel<BicycleSlipLSTMMFMA"),
                     "algorithmic_flops_per_launch": f_net,
                     "algorithmic_flops_network": f_net, "algorithmic_flops_colored_noise_gemm": f_noise,
                     "frac_including_colored_noise_gemm": round(achieved_with_gemm / 157.3, 5),
                     "frac_definition": "frac counts the network's flops only; the colored-noise sampler's in-kernel transform (two "
                                        "FFT decimation steps in registers + a quarter-size MFMA GEMM, in place of the reference's "
                                        "cuFFT pass that moves 1.8 GB per iteration through HBM) is cost, not credit — "
                                        "frac_including_colored_noise_gemm is the number with its GEMM flops",
                     "avg_kernel_us": round(roll_us, 3),
                     "note": "reference data flow for this config moves ~1.8 GB per iteration through HBM (cuRAND spectrum, "
                             "cuFFT, rearrange, setGaussianControls, rollout, weighted reduction); here the samples never "
                             "leave the CU"},
    }


def di_tube_leg(device, with_cpu_baseline=True):
    """BASELINE config 3: DoubleIntegrator Tube-MPPI (CORL2020 parameters), K=8192, T=150, two systems per launch"""
    from common import di_cfg, make_engine
    cfg = di_cfg(K=8192, T=150, tube=True)
    eng = make_engine(cfg, device=device)
    eng.uploadState(np_tile(cfg["x0"], 2))
    eng.optimize(50, True)
    n = 500
    t0 = time.perf_counter()
    eng.optimize(n, True)
    wall = time.perf_counter() - t0
    # what a control loop sees: computeControl (optimisation pass, nominal / actual choice, smoothing pass) and the loop period
    x = cfg["x0"].copy()
    for _ in range(30):
        eng.computeControl(x, 1)
    m_ = 200
    ready = 0.0
    for _ in range(m_):
        eng.getTargetStateSeq()  # the previous call's trajectories have landed: the stream is idle
        t_a = time.perf_counter()
        eng.computeControl(x, 1)
        ready += time.perf_counter() - t_a
    t_a = time.perf_counter()
    for _ in range(m_):
        eng.computeControl(x, 1)
        eng.getControlSeq()
        eng.slideControlSequence(1)
    loop = (time.perf_counter() - t_a) / m_
    roll_us, how = rollout_kernel_us(eng, 100, "rolloutPipelineKernel<DoubleIntegratorDynamics")
    b_alg = 4.0 * 2 * (2.0 * cfg["K"] * cfg["T"] * 2 + 2.0 * cfg["K"] + 2.0 * cfg["T"] * 2)  # SURVEY.md §8d with D = 2 systems
    achieved = b_alg / (roll_us * 1e-6) / 1e9
    roofline = {"bound": "hbm", "kernel": "rolloutPipelineKernel<DoubleIntegratorDynamics, DoubleIntegratorCircleCost, Gaussian, 2 "
                                          "systems folded into the lanes of a wave>",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic("rolloutPipelineKernel<DoubleIntegratorDynamics"), "traffic_static_from_profiles": PMC_FILE,
                "algorithmic_bytes_per_launch": b_alg, "avg_kernel_us": round(roll_us, 3), "avg_kernel_us_source": how,
                "frac_on_iteration": round(b_alg / (wall / n) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "HBM is the ceiling SURVEY.md §8d assigns; the samples of both systems stay in LDS, the kernel is bound by "
                        "the issue rate of its dynamics wave (T dependent steps, 32 rollouts x 2 systems per wave)"}
    eng.close()
    cpu = None
    if with_cpu_baseline:
        try:
            cpu = cpu_baseline(cfg, budget_s=5.0, label="DoubleIntegrator Tube-MPPI K=8192 T=150, two systems")
        except Exception as e:  # noqa: BLE001
            cpu = {"error": str(e)}
    return {"workload": "DoubleIntegrator + DoubleIntegratorCircleCost, Tube-MPPI iteration (actual + nominal system in one "
                        "launch), K=8192, T=150", "value": round(n / wall, 3), "unit": "MPPI iters/s", "cpu_baseline": cpu,
            "roofline": roofline,
            "ms_per_step": round(wall / n * 1e3, 6), "compute_control_ready_us": round(ready / m_ * 1e6, 2),
            "closed_loop_period_us": round(loop * 1e6, 2)}


def racer_elevation_leg(device):
    """the elevation-map RACER models (RacerDubinsElevation, RacerDubinsElevationLSTMSteering with the
    colored-noise sampler of the RACER controllers, the suspension model and the complete model with the mean / uncertainty
    networks) over a synthetic terrain, K=16384, T=100, four lanes per rollout"""
    from common import make_engine
    from test_racer_dubins_elevation import elevation_cfg
    from test_racer_dubins_lstm_steering import steering_cfg
    from test_racer_dubins_lstm_unc import uncertainty_cfg
    from test_racer_dubins_suspension import suspension_cfg
    out = {}
    for key, cfg in (("elevation", elevation_cfg(K=16384, T=100)), ("lstm_steering_colored", steering_cfg(K=16384, T=100)),
                     ("suspension", suspension_cfg(K=16384, T=100)), ("complete_model", uncertainty_cfg(K=16384, T=100))):
        if key.endswith("colored"):
            cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
        eng = make_engine(cfg, device=device)
        eng.uploadState(np_tile(cfg["x0"], 1))
        eng.optimize(20, True)
        n = 200
        t0 = time.perf_counter()
        eng.optimize(n, True)
        wall = time.perf_counter() - t0
        out[key] = {"value": round(n / wall, 3), "unit": "MPPI iters/s", "ms_per_step": round(wall / n * 1e3, 6)}
        eng.close()
    out["workload"] = ("RacerDubinsElevation + QuadraticCost (Gaussian sampler), RacerDubinsElevationLSTMSteering + "
                       "QuadraticCost (colored noise), RacerDubinsElevationSuspension and RacerDubinsElevationLSTMUncertainty "
                       "(the complete RACER model; Gaussian sampler), 240 x 240 elevation / normals maps, K=16384, T=100, "
                       "block shape (64, 4)")
    return out


def reference_order_leg(device):
    """What the opt-in reference-order reduction (mppi_set_reduction_mode: global rho, eta in double in index order, per-rollout
    weight / eta, cells of 32 rollouts — the mode in which a free-running closed loop is bit-identical to the oracle) costs per
    iteration next to the default fused reduction: samples through HBM + three K-long serial sums."""
    import mppi_generic_amd as m
    from common import autorally_cfg, cartpole_cfg, make_engine
    out = {}
    for key, cfg, n in (("cartpole_16384x100", cartpole_cfg(K=K_PER_GPU, T=T), 400),
                        ("autorally_nn_16384x150", autorally_cfg(K=16384, T=150, lambda_=1.0), 100)):
        eng = make_engine(cfg, device=device)
        eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
        eng.uploadState(cfg["x0"])
        eng.optimize(20, True)
        t0 = time.perf_counter()
        eng.optimize(n, True)
        wall = time.perf_counter() - t0
        out[key] = {"ms_per_step": round(wall / n * 1e3, 6), "value": round(n / wall, 3), "unit": "MPPI iters/s"}
        eng.close()
    out["definition"] = ("one optimisation iteration with MPPI_REDUCTION_REFERENCE_ORDER: rollout kernel (samples dumped to HBM) + "
                         "exactWeightsKernel + exactReductionCellsKernel + exactReductionFinalKernel; parity mode, not the headline")
    return out


def np_tile(x, d):
    import numpy as np
    return np.tile(x, (d, 1))


def self_launch(n_gpus):
    """`python3 bench.py --gpus N` with no launcher around it: re-exec under torch.distributed.run, one rank per GPU
    (the driver's own form for N > 1; both forms end in the same main())"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the P2P mailbox and RCCL both need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


class ShardedRun:
    """One K-sharded problem over the ranks of this job: engine, the exchange that was negotiated for it, and the timed
    region.  `mode_hint` carries the exchange an earlier leg of the same job settled on, so that later legs do not retry
    paths that already failed."""

    def __init__(self, cfg, rank, world, local_rank, dist, mode_hint=None):
        import mppi_generic_amd as m
        from common import make_engine
        self.cfg, self.rank, self.world, self.dist = cfg, rank, world, dist
        self.make = lambda: make_engine(cfg, device=local_rank, rank=rank, world_size=world)
        self.eng = self.make()
        self.exchange = {"mode": "none", "text": "none", "tried": [], "bit_equal_u_across_ranks": None, "rccl_ranks": None}
        self.run = lambda n: self.eng.optimize(n, True)
        if world > 1:
            self._negotiate(m, mode_hint or os.environ.get("MPPI_BENCH_EXCHANGE", "auto"))

    # ---- exchange negotiation: P2P mailbox, then the library's RCCL communicator, then host-staged over gloo
    def _guarded(self, fn, limit_s, what):
        """run fn under a watchdog: an exchange that never completes must not hang the bench"""
        import threading
        res = {"ok": False, "why": what + " did not finish within %d s" % limit_s}

        def attempt():
            try:
                fn()
                res["ok"] = True
            except Exception as e:  # noqa: BLE001
                res["why"] = what + ": " + str(e)

        th = threading.Thread(target=attempt, daemon=True)
        th.start()
        th.join(limit_s)
        if th.is_alive():
            globals()["_HARD_EXIT"] = True  # a thread is parked inside a library call: leave with os._exit
            res["ok"] = False
        return res["ok"], res["why"]

    def _first_iteration_agrees(self):
        """one exchanged iteration: finite, and every rank ends with the same bits"""
        import hashlib
        import numpy as np
        self.eng.uploadState(self.cfg["x0"])
        self.eng.optimize(1, True)
        u = self.eng.getOptimalControlSeq()
        if not np.isfinite(u).all():
            raise RuntimeError("non-finite result after the first exchanged iteration")
        digests = [None] * self.world
        self.dist.all_gather_object(digests, hashlib.sha1(u.tobytes()).hexdigest())
        if len(set(digests)) != 1:
            raise RuntimeError("ranks disagree on u* after the first exchanged iteration")

    def _all_ok(self, flag_why):
        flags = [None] * self.world
        self.dist.all_gather_object(flags, flag_why)
        return all(f[0] for f in flags), next((f[1] for f in flags if not f[0]), "")

    def _negotiate(self, m, want):
        import ctypes as C
        from mppi_generic_amd.distributed import HostStagedExchange
        lib = m.load_library()
        x, dist, world = self.exchange, self.dist, self.world
        n_rec = self.eng.exchangeBuffers()[2]
        done = False
        if want in ("auto", "p2p"):
            def p2p_setup():
                handles = [None] * world
                dist.all_gather_object(handles, self.eng.p2pMailboxHandle())
                self.eng.p2pConnect(handles)
                self._first_iteration_agrees()
            ok, why = self._all_ok(self._guarded(p2p_setup, 120, "P2P mailbox setup"))
            x["tried"].append({"mode": "p2p", "ok": ok, "why": "" if ok else why[:200]})
            if ok:
                done = True
                x["mode"] = "p2p"
                x["text"] = ("p2p mailbox over xGMI: one record of %d floats written into every peer's memory per iteration "
                             "(no collective library, no host)" % n_rec)
            else:
                self.eng = self.make()
        if not done and want in ("auto", "rccl"):
            uid = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                nb = C.c_size_t()
                st = lib.mppi_rccl_unique_id(buf, 128, C.byref(nb))
                uid[0] = bytes(buf.raw) if st == 0 else None
            dist.broadcast_object_list(uid, src=0)
            if uid[0] is not None:
                def rccl_setup():
                    self.eng.commInitRccl(uid[0])
                    self._first_iteration_agrees()
                ok, why = self._all_ok(self._guarded(rccl_setup, 120, "RCCL communicator setup / first all-gather"))
            else:
                ok, why = False, "no RCCL unique id"
            x["tried"].append({"mode": "rccl", "ok": ok, "why": "" if ok else why[:200]})
            if ok:
                done = True
                x["mode"] = "rccl"
                x["rccl_ranks"] = world
                x["text"] = "rccl all-gather of %d floats per rank per iteration (library-owned communicator, %d ranks)" % (n_rec, world)
            else:
                self.eng = self.make()
        if not done:
            hx = HostStagedExchange(self.eng)
            self.run = lambda n: (hx.iterate(n), self.eng.synchronize())
            ok, why = self._all_ok(self._guarded(self._first_iteration_agrees_host(hx), 120, "host-staged exchange"))
            x["tried"].append({"mode": "host", "ok": ok, "why": "" if ok else why[:200]})
            x["mode"] = "host"
            x["text"] = "host-staged all-gather over gloo of %d floats per rank per iteration" % n_rec
            done = ok
        x["bit_equal_u_across_ranks"] = bool(done)
        lost = ["%s: %s" % (t["mode"], t["why"]) for t in x["tried"] if not t["ok"]]
        if lost:
            x["text"] += " [fell back after: " + "; ".join(lost) + "]"

    def _first_iteration_agrees_host(self, hx):
        def check():
            import hashlib
            import numpy as np
            self.eng.uploadState(self.cfg["x0"])
            hx.iterate(1)
            self.eng.synchronize()
            u = self.eng.getOptimalControlSeq()
            if not np.isfinite(u).all():
                raise RuntimeError("non-finite result after the first exchanged iteration")
            digests = [None] * self.world
            self.dist.all_gather_object(digests, hashlib.sha1(u.tobytes()).hexdigest())
            if len(set(digests)) != 1:
                raise RuntimeError("ranks disagree on u* after the first exchanged iteration")
        return check

    # ---- the timed region
    def barrier(self):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()
        self.eng.synchronize()

    def timed_region(self, steps):
        """EXACTLY `steps` iterations between barrier + synchronize on both sides; max over ranks"""
        import torch
        self.barrier()
        t0 = time.perf_counter()
        self.run(steps)
        torch.cuda.synchronize()
        self.barrier()
        dt_ = time.perf_counter() - t0
        if self.dist is not None:
            tt = torch.tensor([dt_], dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_

    def measure(self, steps, warmup, min_time, max_reps=400):
        """warm-up, then the K-step region repeated until min_time seconds are timed (every rank the same number of times,
        decided from the first, max-reduced, repetition); returns (median seconds per region, all repetitions)"""
        import numpy as np
        self.eng.uploadState(self.cfg["x0"])
        self.run(warmup)
        reps = [self.timed_region(steps)]
        n_rep = int(min(max_reps, max(1, -(-min_time // max(reps[0], 1e-6)))))
        for _ in range(n_rep - 1):
            reps.append(self.timed_region(steps))
        return float(np.median(reps)), reps

    def kernel_times_us(self, n_ev, fallback_us):
        """(iteration, rollout kernel) in us from HIP events on the engine's own stream (separate, untimed pass)"""
        try:
            ms_total, ms_roll = self.eng.timeIterations(n_ev)
            if ms_total == 0.0:  # host-staged fallback: the exchange is driven from here, use the wall-clock step
                ms_total = fallback_us * 1e-3 * n_ev
        except Exception:  # noqa: BLE001
            ms_total = ms_roll = fallback_us * 1e-3 * n_ev
        return ms_total / n_ev * 1e3, ms_roll / n_ev * 1e3

    def close(self):
        try:
            self.eng.close()
        except Exception:  # noqa: BLE001
            pass


WORKLOAD_TEXT = {
    "cartpole": ("Cartpole (CartpoleDynamics + CartpoleQuadraticCost, examples/cartpole_example.cu config) VanillaMPPI "
                 "optimisation iteration, T=100, dt=0.02, lambda=0.25, sigma=5, Philox noise fused in the rollout kernel"),
    "autorally": ("AutoRally NeuralNetModel<7,2,3> (FNN 6-32-32-4, synthetic weights) + ARStandardCost (600x600 generated track "
                  "map) VanillaMPPI optimisation iteration, T=150, MFMA forward, Philox noise fused in the rollout kernel"),
}


K_BASE = {"cartpole": K_PER_GPU, "autorally": K_PER_GPU, "lstm_colored": 65536}  # the BASELINE problem of every workload


def workload_cfg(workload, k_total):
    from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg
    if workload == "autorally":
        return autorally_cfg(K=k_total, T=150, lambda_=1.0), 150
    if workload == "lstm_colored":  # BASELINE config 5: "K=65536 T=200, 8xMI355X"
        cfg = bicycle_lstm_cfg(K=k_total, T=200, lambda_=1.0)
        cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
        return cfg, 200
    return cartpole_cfg(K=k_total, T=T), T


# What a first multi-GPU run should show, written down BEFORE one exists (no 8-GPU node was ever available to the builder) so
# that the driver's curve can falsify it.  Inputs are one-GPU measurements (profiles/r04_d_*, tools/kscan.py) and two
# estimates that have never been measured across devices: the sharded merge (combineShardedKernel: local merge + post + ticket +
# wait + global merge, ~3.5 us of body behind a 1.55 us boundary) and the xGMI hop of a ~0.4-1.6 KB record + flag (2-3 us).
#   * the rollout kernels do NOT get faster below one block per CU: K = 16384 is 256 blocks of 64 rollouts on 256 CUs, and a
#     block's time is T dependent steps of its dynamics wave whatever the other CUs do (kscan: Cartpole K = 2048 / 4096 / 16384
#     within 3 %) -> strong scaling of the two K = 16384 problems is < 1x at every N;
#   * config 5 (K = 65536: 1024 blocks, one per CU at a time = 4 rounds) is the problem that shards: N = 2 -> 2 rounds,
#     N = 4 -> 1 round, N = 8 -> 1 round on half the CUs (no further gain).
PREDICTED_MS_PER_STEP = {
    "inputs": {"cartpole_rollout_kernel_us": 22.7, "autorally_rollout_kernel_us": 176.0, "lstm_colored_round_us": 410.0,
               "sharded_merge_us": 5.0, "xgmi_hop_us": 2.5,
               "source": "one-GPU measurements (profiles/r05_a_kernel_stats.csv: plain Cartpole instantiation 22.5-22.9 us, "
                         "AutoRally-NN 176-179 us, config 5 1.62-1.64 ms = 4 rounds); merge + hop estimated, never measured across devices"},
    "cartpole_strong": {2: 0.0302, 4: 0.0302, 8: 0.0302},       # 22.7 + 5.0 + 2.5 us, against 0.0250 on one GPU (one launch)
    "cartpole_weak": {2: 0.0302, 4: 0.0302, 8: 0.0302},         # same step, N x the rollouts
    "autorally_strong": {2: 0.1835, 4: 0.1835, 8: 0.1835},      # 176 + 7.5 us, against 0.181 on one GPU
    "autorally_weak": {2: 0.1835, 4: 0.1835, 8: 0.1835},
    "lstm_colored_strong": {2: 0.8275, 4: 0.4175, 8: 0.4175},   # rounds x 410 us + 7.5 us, against 1.64 on one GPU
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="which line is the headline `value` at N > 1.  Default strong: the BASELINE problem (K = 16384 rollouts "
                         "in total) split over the GPUs — north_star's 'Cartpole (K=16384, T=100) at 1/2/4/8'; the other one "
                         "is reported beside it in the same JSON line")
    ap.add_argument("--workload", choices=["cartpole", "autorally"], default="cartpole")
    ap.add_argument("--min-time", type=float, default=0.25, help="repeat the K-step timed region until this many seconds are timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--primary-only", action="store_true", help="skip the secondary legs (AutoRally-NN, LSTM+colored, DI-Tube, RACER elevation)")
    args = ap.parse_args()

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        self_launch(args.gpus)  # does not return

    import numpy as np
    import torch
    import mppi_generic_amd as m  # noqa: F401

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MPPI_BENCH_DEVICE"):  # test hook: several ranks on one GPU (exercises the multi-rank control flow)
        local_rank = int(os.environ["MPPI_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    assert world == n_gpus, "--gpus %d but WORLD_SIZE=%d" % (n_gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU; the product has no CPU path"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # control plane only (barrier, max over ranks, shipping mailbox handles / the RCCL id): gloo.  The DATA path is the
        # library's own P2P mailbox or RCCL communicator on the engine's stream.  (This image's torch wheel bundles a second
        # ROCm runtime; device buffers and streams of libmppi_amd.so belong to the system runtime, so torch's NCCL backend is
        # not used on them.)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")

    scaling = args.scaling or ("strong" if world > 1 else "weak")
    strong = scaling == "strong"
    scaling_reported = scaling if world > 1 else "none"  # one GPU: nothing scales

    def leg(workload, strong_, hint, steps, warmup, min_time):
        k_total = K_BASE[workload] if strong_ else K_BASE[workload] * world
        cfg, t_steps = workload_cfg(workload, k_total)
        sr = ShardedRun(cfg, rank, world, local_rank, dist, hint)
        elapsed, reps = sr.measure(steps, warmup, min_time)
        units = steps if strong_ else world * steps
        res = {"value": round(units / elapsed, 3), "ms_per_step": round(elapsed / steps * 1e3, 6),
               "rollouts_per_gpu": k_total // world, "global_rollouts": k_total, "num_timesteps": t_steps,
               "repetitions": len(reps), "ms_per_step_min": round(min(reps) / steps * 1e3, 6),
               "ms_per_step_max": round(max(reps) / steps * 1e3, 6), "exchange": sr.exchange["text"],
               "exchange_mode": sr.exchange["mode"],
               "finite": bool(np.isfinite(sr.eng.getOptimalControlSeq()).all())}
        pred = PREDICTED_MS_PER_STEP.get("%s_%s" % (workload, "strong" if strong_ else "weak"), {}).get(world)
        if world > 1 and pred is not None:
            res["predicted_ms_per_step"] = pred
        return sr, cfg, res, elapsed, reps

    def exchange_paths(workload, steps, warmup, negotiated, head_iter_us, head_roll_us):
        """BOTH data paths of the K-sharded iteration on the strong-scaling problem, each forced in turn (the headline reports
        only the one that won the negotiation): event-timed iteration minus event-timed rollout kernel = what merge + exchange
        cost on that path.  A path that cannot run here says why (e.g. RCCL refuses two ranks on one device)."""
        out = {}
        for mode in ("p2p", "rccl"):
            key = "exchange_%s_us" % mode
            try:
                if mode == negotiated:
                    it_us, ro_us, ms_step, ok, why, ranks = head_iter_us, head_roll_us, None, True, "", world
                else:
                    cfg2, _ = workload_cfg(workload, K_BASE[workload])
                    sr2 = ShardedRun(cfg2, rank, world, local_rank, dist, mode)
                    ok = sr2.exchange["mode"] == mode
                    why = "; ".join(t["why"] for t in sr2.exchange["tried"] if t["mode"] == mode and not t["ok"])
                    it_us = ro_us = ms_step = None
                    ranks = sr2.exchange.get("rccl_ranks") if mode == "rccl" else world
                    if ok:
                        elapsed2, _ = sr2.measure(steps, warmup, 0.05)
                        ms_step = round(elapsed2 / steps * 1e3, 6)
                        it_us, ro_us = sr2.kernel_times_us(min(200, max(20, steps)), elapsed2 / steps * 1e6)
                    sr2.close()
                if ok:
                    out[key] = round(it_us - ro_us, 3)
                    out[mode] = {"ok": True, "iteration_us_event_timed": round(it_us, 3), "rollout_kernel_us": round(ro_us, 3),
                                 "ms_per_step": ms_step, "ranks": ranks, "negotiated": mode == negotiated}
                else:
                    out[key] = None
                    out[mode] = {"ok": False, "refused": (why or "path not available")[:300], "negotiated": False}
            except Exception as e:  # noqa: BLE001
                out[key] = None
                out[mode] = {"ok": False, "refused": str(e)[:300], "negotiated": False}
        out["rccl_ranks"] = out["rccl"].get("ranks") if out["rccl"]["ok"] else None
        out["definition"] = ("exchange_<path>_us = event-timed iteration - event-timed rollout kernel on the strong-scaling headline "
                             "problem with that path forced: local merge + post/all-gather + wait + global merge")
        return out

    # ------------------------------------------------------------------ the headline leg
    sr, cfg, head, elapsed, reps = leg(args.workload, strong, None, args.steps, args.warmup, args.min_time)
    hint = sr.exchange["mode"] if world > 1 else None
    k_total, k_local, t_steps = head["global_rollouts"], head["rollouts_per_gpu"], head["num_timesteps"]
    x0 = cfg["x0"]
    n_ev = min(200, max(20, args.steps))
    iter_us, roll_us = sr.kernel_times_us(n_ev, elapsed / args.steps * 1e6)
    C_dim = sr.eng.CONTROL_DIM
    if args.workload == "autorally":
        f_alg = 2.0 * (6 * 32 + 32 * 32 + 32 * 4) * k_local * t_steps
        achieved = f_alg / (roll_us * 1e-6) / 1e12
        roofline = {
            "bound": "mfma", "kernel": "rolloutPipelineRepKernel<NeuralNetModelMFMA<7,2,3>,ARStandardCost,Gaussian,true>",
            "achieved": round(achieved, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 5),
            "traffic": pmc_traffic("rolloutPipelineRepKernel<NeuralNetModelMFMA") if k_local == K_PER_GPU else None,
            "traffic_static_from_profiles": PMC_FILE,
            "algorithmic_flops_per_launch": f_alg, "avg_kernel_us": round(roll_us, 3),
            "avg_iteration_us_event_timed": round(iter_us, 3),
        }
    else:
        b_alg = 4.0 * (2.0 * k_local * t_steps * C_dim + 2.0 * k_local + 2.0 * t_steps * C_dim)
        achieved = b_alg / (roll_us * 1e-6) / 1e9
        # one GPU: the instantiation that merges the previous iteration's records in its sampler waves (STREAM_MERGE = true);
        # K-sharded: the plain one, the merge is combineShardedKernel
        tail = "false, false, true>" if world == 1 else "false, false, false>"
        roofline = {
            "bound": "hbm", "kernel": "rolloutPipelineKernel<CartpoleDynamics, CartpoleQuadraticCost, GaussianDistribution<"
                                      "CartpoleDynamicsParams>, 1, true, " + tail,
            "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic("rolloutPipelineKernel<CartpoleDynamics", tail) if k_local == K_PER_GPU else None,
            "traffic_static_from_profiles": PMC_FILE,
            "pipe_utilisation_static_from_profiles": pmc_pipe_util("rolloutPipelineKernel<CartpoleDynamics", tail),
            "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, " + PMC_FILE + " "
                              "(2*FETCH_SIZE + WRITE_SIZE), collected by the builder, NOT in this run: the sample tensor never "
                              "reaches HBM, so traffic << algorithmic bytes",
            "algorithmic_bytes_per_launch": b_alg, "avg_kernel_us": round(roll_us, 3),
            "avg_iteration_us_event_timed": round(iter_us, 3),
            # SURVEY.md §8d prices the roofline on the whole iteration (t_iter), the contract on the dominant kernel: both
            "frac_on_iteration": round(b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            "note": "the HBM roofline is the ceiling SURVEY.md §8d assigns, not the one that binds: the kernel is issue-bound on "
                    "the dynamics wave (T dependent Euler steps per rollout, one wave per CU at K=16384 — K=32768 costs only "
                    "~17 % more time, DESIGN.md §5); see issue_floor for a floor that does not depend on the kernel's own timing",
        }
        if world == 1:
            roofline["note"] += ("; on one GPU this kernel also does the softmin merge of the previous iteration's block records "
                                 "(in its sampler waves — until round 4 a second launch of 3.3 us + a 1.55 us boundary), so its "
                                 "duration is the whole iteration: compare frac with (rollout + merge) of earlier rounds, "
                                 "13.24 MB / 27.0 us = 0.061, not with the rollout kernel alone")
        if world == 1:
            n_launch = launches_per_iteration(sr.eng)
            roofline["launches_per_iteration"] = n_launch
            try:
                roofline["issue_floor"] = issue_floor(local_rank, iter_us, roll_us, n_launch=n_launch)
            except Exception as e:  # noqa: BLE001
                roofline["issue_floor"] = {"error": str(e)}
            try:
                roofline["latency_model"] = latency_model(local_rank, iter_us, n_launch)
            except Exception as e:  # noqa: BLE001
                roofline["latency_model"] = {"error": str(e)}
    exchange_info = dict(sr.exchange)
    sr.close()
    if world > 1:
        pred = PREDICTED_MS_PER_STEP.get("%s_%s" % (args.workload, scaling), {}).get(world)
        if pred is not None:
            head["predicted_ms_per_step"] = pred

    # ------------------------------------------------------------------ N > 1: the other scaling mode and the other workload
    extra = {}
    if world > 1 and not args.primary_only:
        other = "autorally" if args.workload == "cartpole" else "cartpole"
        try:
            extra["exchange_paths"] = exchange_paths(args.workload, max(20, min(args.steps, 200)), max(5, min(args.warmup, 50)),
                                                     exchange_info["mode"], iter_us, roll_us)
            extra["exchange_p2p_us"] = extra["exchange_paths"]["exchange_p2p_us"]
            extra["exchange_rccl_us"] = extra["exchange_paths"]["exchange_rccl_us"]
            extra["rccl_ranks"] = extra["exchange_paths"]["rccl_ranks"]
        except Exception as e:  # noqa: BLE001
            extra["exchange_paths"] = {"error": str(e)}
        WORKLOAD_TEXT.setdefault("lstm_colored", "LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights) + "
                                 "ARStandardCost + ColoredNoise sampler (exponents [1,1], offset decay 0.97), ColoredMPPI "
                                 "iteration, T=200 (BASELINE config 5)")
        for key, wl, st_ in (("weak" if strong else "strong", args.workload, not strong),
                             (other + "_strong", other, True), (other + "_weak", other, False),
                             ("lstm_colored_strong", "lstm_colored", True)):
            steps = args.steps if wl == "cartpole" else max(20, min(args.steps, 200))
            if wl == "lstm_colored":
                steps = max(5, min(args.steps, 40))
            try:
                sr2, _, res, _, _ = leg(wl, st_, hint, steps, max(5, min(args.warmup, steps)), args.min_time)
                res["steps"] = steps
                res["workload"] = WORKLOAD_TEXT[wl] + ", K=%d rollouts %s" % (
                    K_BASE[wl], "in total, split over the GPUs" if st_ else "per GPU")
                res["scaling"] = "strong" if st_ else "weak"
                sr2.close()
            except Exception as e:  # noqa: BLE001
                res = {"error": str(e)}
            extra[key] = res

    if rank == 0:
        wl = WORKLOAD_TEXT[args.workload] + ", K=%d rollouts %s" % (
            K_PER_GPU, "in total, split over the GPUs" if (strong and world > 1) else "per GPU")
        out = {
            "metric": "MPPI iters/sec (KxT rollouts)", "value": head["value"], "unit": "MPPI iters/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "predicted_ms_per_step": head.get("predicted_ms_per_step"),
            "higher_is_better": True, "scaling": scaling_reported,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wl,
                "rollouts_per_gpu": k_local, "global_rollouts": k_total, "num_timesteps": t_steps,
                "parallelism": "K-sharded x%d" % world, "exchange": head["exchange"],
                "exchange_negotiation": exchange_info,
                "unit_definition": ("one optimisation-loop body over the K=16384-rollout BASELINE problem (all ranks together)"
                                    if strong else
                                    "one optimisation-loop body over K=16384 rollouts; value sums the units of all ranks"),
            },
            "timed_region": {"repetitions": len(reps), "ms_per_step_min": head["ms_per_step_min"],
                             "ms_per_step_max": head["ms_per_step_max"],
                             "rule": "each repetition = exactly --steps iterations between barrier + synchronize; "
                                     "ms_per_step is the median repetition"},
            # scalars the driver's consistency check can read: seconds of the median K-step repetition (= steps * ms_per_step)
            # and of everything that was timed
            "timed_region_s": round(elapsed, 6),
            "timed_total_s": round(float(sum(reps)), 6),
            "finite": head["finite"],
            "roofline": roofline,
        }
        out.update(extra)
        if world > 1:
            out["predicted_ms_per_step_model"] = PREDICTED_MS_PER_STEP["inputs"]
            out["multi_gpu_note"] = ("headline = strong scaling of the BASELINE problem; DESIGN.md §6 predicts < 1.0x for it (the "
                                     "rollout kernel is T dependent steps of one wave per CU whatever K is; sharding adds one merge "
                                     "launch and the hop) and ~N x for the weak line beside it")
        # secondary workloads of the north star (not the headline `value`)
        if not args.primary_only and world == 1 and args.workload == "cartpole":
            for key, leg_fn in (("autorally_nn", autorally_leg), ("lstm_colored", lstm_colored_leg), ("di_tube", di_tube_leg),
                                ("racer_elevation", racer_elevation_leg), ("robust_autorally_nn", robust_autorally_leg),
                                ("robust_double_integrator", robust_di_leg), ("robust_racer", robust_racer_leg),
                                ("reference_order_reduction", reference_order_leg)):
                try:
                    if key in ("autorally_nn", "di_tube", "lstm_colored"):
                        out[key] = leg_fn(local_rank, with_cpu_baseline=not args.no_cpu_baseline)
                    else:
                        out[key] = leg_fn(local_rank)
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": str(e)}
        if world == 1 and args.workload == "cartpole":
            try:
                out["compute_control"] = compute_control_latency(local_rank, x0)
                out["compute_control_latency_us"] = out["compute_control"]["control_ready_us"]
                out["compute_control_calls_per_s"] = round(1e6 / out["compute_control"]["closed_loop_period_us"], 1)
            except Exception as e:  # noqa: BLE001
                out["compute_control"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, budget_s=12.0 if args.workload == "cartpole" else 20.0)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    if globals().get("_HARD_EXIT"):
        sys.stdout.flush()
        os._exit(0)
