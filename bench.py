#!/usr/bin/env python3
"""bench.py — MPPI iterations/s of the HIP rollout-and-reduce engine (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the optimisation-loop body (sample -> K x T rollout -> baseline/normExp -> weighted reduction,
mean <- u*) with x0 and the control mean already resident in HBM.  Workload at every N: Cartpole
(CartpoleDynamics + CartpoleQuadraticCost), K = 16384 rollouts PER GPU, T = 100, fp32, Philox noise drawn in the kernel.
N > 1 is weak scaling: rank r owns rollouts [r*K, (r+1)*K) of a K*N-rollout problem and the ranks exchange one
(T*C+4)-float record per iteration (RCCL all-gather) — value counts K-rollout iteration units over all ranks.

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
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_hbm_traffic.json")))
    return files[-1] if files else os.path.join("profiles", "none.json")


PMC_FILE = _latest_pmc_file()


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the kernel whose (shortened) name starts with kernel_prefix, from the committed PMC passes
    (tools/profile_bench.sh + tools/pmc_summary.py -> profiles/), or None"""
    try:
        d = json.load(open(os.path.join(REPO, PMC_FILE)))
        for name, e in d["kernels"].items():
            if name.startswith(kernel_prefix) and e.get("hbm_traffic_bytes_per_launch"):
                return round(e["hbm_traffic_bytes_per_launch"], 1)
    except Exception:  # noqa: BLE001
        pass
    return None


def pmc_pipe_util(kernel_prefix):
    """{"mfma_util_percent", "valu_busy_percent"} of the kernel from the committed SQ counter pass
    (tools/profile_bench.sh + tools/pmc_mfma_summary.py -> profiles/r*_pmc_mfma_valu.json), or None"""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_mfma_valu.json")))
        d = json.load(open(files[-1]))
        for name, e in d["kernels"].items():
            if name.startswith(kernel_prefix):
                return {"mfma_util_percent": round(e.get("MfmaUtil_percent", 0.0), 2),
                        "valu_busy_percent": round(e.get("VALUBusy_percent", 0.0), 2),
                        "source": os.path.relpath(files[-1], REPO) + " (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, "
                                  "GRBM_GUI_ACTIVE)"}
    except Exception:  # noqa: BLE001
        pass
    return None


def cpu_baseline(cfg, budget_s=12.0):
    """oracle iterations/s on the host cores, bounded sample (never the thing shipped or measured as `value`)"""
    import numpy as np
    import pyoracle as po
    from common import make_oracle
    o = make_oracle(cfg)
    K, Tn, C = cfg["K"], cfg["T"], o.C
    eps = po.philox_normal(42, 0, K, Tn, C)
    mean = np.zeros((1, Tn, C), np.float32)
    threads = max(1, min(po.max_threads(), usable_cores()))
    out = {}
    for label, th in (("all", threads), ("one", 1)):
        o.time_iterations(cfg["x0"], mean, eps, 1, th)  # warm-up
        t1 = o.time_iterations(cfg["x0"], mean, eps, 1, th)
        n = max(1, int(budget_s / 2 / max(t1, 1e-4)))
        n = min(n, 400)
        tt = o.time_iterations(cfg["x0"], mean, eps, n, th)
        out[label] = (n / tt, n, th)
    v, n, th = out["all"]
    return {
        "value": round(v, 3), "unit": "MPPI iters/s", "cores": th, "kind": "port",
        "sample": "%d iterations of the same workload (Cartpole K=%d T=%d, one optimisation-loop body each) with the "
                  "rollouts spread over %d OpenMP threads" % (n, K, Tn, th),
        "value_1core": round(out["one"][0], 3),
        "sample_1core": "%d iterations, single thread (the reference's CPU path is single-threaded)" % out["one"][1],
    }


def autorally_leg(device):
    """AutoRally NeuralNetModel (FNN 6-32-32-4, synthetic weights) + ARStandardCost, K=16384, T=150, one GPU:
    iterations/s and the MFMA roofline of the NN forward (F_alg = 2 * sum(MAC) * K * T)."""
    from common import autorally_cfg, make_engine
    K, Tn = 16384, 150
    cfg = autorally_cfg(K=K, T=Tn, lambda_=1.0)
    eng = make_engine(cfg, device=device)
    eng.uploadState(cfg["x0"])
    eng.optimize(20, True)
    n = 100
    t0 = time.perf_counter()
    eng.optimize(n, True)
    wall = time.perf_counter() - t0
    ms_total, ms_roll = eng.timeIterations(50)
    roll_us = ms_roll / 50 * 1e3
    f_alg = 2.0 * (6 * 32 + 32 * 32 + 32 * 4) * K * Tn
    achieved = f_alg / (roll_us * 1e-6) / 1e12
    return {
        "workload": "AutoRally NeuralNetModel<7,2,3> (FNN 6-32-32-4, synthetic weights) + ARStandardCost (600x600 "
                    "generated track map), VanillaMPPI iteration, K=16384, T=150, block (64 rollouts x 4 MFMA lanes)",
        "value": round(n / wall, 3), "unit": "MPPI iters/s", "ms_per_step": round(wall / n * 1e3, 6),
        "roofline": {"bound": "mfma", "kernel": "rolloutPipelineRepKernel<NeuralNetModelMFMA<7,2,3>,ARStandardCost,Gaussian,true>",
                     "achieved": round(achieved, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 5),
                     "traffic": pmc_traffic("rolloutPipelineRepKernel<NeuralNetModelMFMA"),
                     "pipe_utilisation_pmc": pmc_pipe_util("rolloutPipelineRepKernel<NeuralNetModelMFMA"),
                     "algorithmic_flops_per_launch": f_alg, "avg_kernel_us": round(roll_us, 3),
                     "note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32) peak = the fp32 vector peak; the kernel is bound by VALU "
                             "issue, not by the matrix cores: per 16 rollouts and step a dynamics wave issues 28 MFMAs and ~500 "
                             "VALU instructions (64 packed-fp32 tanh per rollout, kinematics, Euler); sampler and cost run "
                             "once per rollout in helper waves (two samplers, two relaying cost waves: one helper per SIMD); K=16384 is exactly one dynamics wave per SIMD"},
    }


def lstm_colored_leg(device):
    """BASELINE config 5: LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights) + ARStandardCost +
    ColoredNoise sampler (exponents 1, offset decay 0.97), ColoredMPPI iteration, K=65536, T=200, one GPU.
    MFMA roofline: F_alg = 2*(4H(I+H) + (H+I)*M + M*OUT)*K*T for the network + 2*(2T+2)*T*C*K for the colored-noise GEMM."""
    from common import bicycle_lstm_cfg, make_engine
    K, Tn = 65536, 200
    cfg = bicycle_lstm_cfg(K=K, T=Tn, lambda_=1.0)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng = make_engine(cfg, device=device)
    eng.uploadState(cfg["x0"])
    eng.optimize(5, True)
    n = 30
    t0 = time.perf_counter()
    eng.optimize(n, True)
    wall = time.perf_counter() - t0
    ms_total, ms_roll = eng.timeIterations(20)
    roll_us = ms_roll / 20 * 1e3
    f_net = 2.0 * (4 * 16 * (6 + 16) + (16 + 6) * 32 + 32 * 4) * K * Tn
    f_noise = 2.0 * (2 * Tn + 2) * Tn * 2 * K
    achieved = (f_net + f_noise) / (roll_us * 1e-6) / 1e12
    return {
        "workload": "LSTM bicycle-slip dynamics (LSTM(6,16) + MLP {22,32,4}, synthetic weights) + ARStandardCost + ColoredNoise "
                    "sampler (exponents [1,1], offset_decay_rate 0.97), ColoredMPPI iteration, K=65536, T=200, block (64 "
                    "rollouts x 4 MFMA lanes)",
        "value": round(n / wall, 3), "unit": "MPPI iters/s", "ms_per_step": round(wall / n * 1e3, 6),
        "roofline": {"bound": "mfma", "kernel": "rolloutPipelineRepKernel<BicycleSlipLSTMMFMA,ARStandardCost,ColoredNoise,false>",
                     "achieved": round(achieved, 4), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 5),
                     "traffic": pmc_traffic("rolloutPipelineRepKernel<BicycleSlipLSTMMFMA"),
                     "pipe_utilisation_pmc": pmc_pipe_util("rolloutPipelineRepKernel<BicycleSlipLSTMMFMA"),
                     "algorithmic_flops_per_launch": f_net + f_noise,
                     "algorithmic_flops_network": f_net, "algorithmic_flops_colored_noise_gemm": f_noise,
                     "avg_kernel_us": round(roll_us, 3),
                     "note": "reference data flow for this config moves ~1.8 GB per iteration through HBM (cuRAND spectrum, "
                             "cuFFT, rearrange, setGaussianControls, rollout, weighted reduction); here the samples never "
                             "leave the CU"},
    }


def di_tube_leg(device):
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
    return {"workload": "DoubleIntegrator + DoubleIntegratorCircleCost, Tube-MPPI iteration (actual + nominal system in one "
                        "launch), K=8192, T=150", "value": round(n / wall, 3), "unit": "MPPI iters/s",
            "ms_per_step": round(wall / n * 1e3, 6)}


def np_tile(x, d):
    import numpy as np
    return np.tile(x, (d, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--primary-only", action="store_true", help="skip the secondary legs (AutoRally-NN, LSTM+colored, DI-Tube)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import mppi_generic_amd as m
    from common import autorally_cfg, cartpole_cfg, make_engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MPPI_BENCH_DEVICE"):  # test hook: several ranks on one GPU (exercises the multi-rank control flow)
        local_rank = int(os.environ["MPPI_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    assert world == n_gpus or (world == 1 and n_gpus == 1), "launch with torch.distributed.run for --gpus > 1"
    assert torch.cuda.is_available(), "bench.py needs a GPU; the product has no CPU path"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # control plane only (barrier, max over ranks, shipping the RCCL id): gloo.  The DATA path is the library's own
        # RCCL communicator on the engine's stream.  (This image's torch wheel bundles a second ROCm runtime; device
        # buffers and streams of libmppi_amd.so belong to the system runtime, so torch's NCCL backend is not used on them.)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")

    cfg = cartpole_cfg(K=K_PER_GPU * world, T=T)
    eng = make_engine(cfg, device=local_rank, rank=rank, world_size=world)
    x0 = cfg["x0"]
    exchange = "none"
    run = lambda n: eng.optimize(n, True)  # noqa: E731
    if world > 1:
        import ctypes as C
        from mppi_generic_amd.distributed import HostStagedExchange
        lib = m.load_library()
        uid = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            nb = C.c_size_t()
            st = lib.mppi_rccl_unique_id(buf, 128, C.byref(nb))
            uid[0] = bytes(buf.raw) if st == 0 else None
        dist.broadcast_object_list(uid, src=0)
        native_ok, why = False, "no RCCL unique id"
        if uid[0] is not None:
            # the first collective runs under a watchdog: a communicator that never forms must not hang the bench
            import threading
            res = {"ok": False, "why": "RCCL communicator setup or first all-gather did not finish within 120 s"}

            def attempt():
                try:
                    eng.commInitRccl(uid[0])
                    eng.uploadState(x0)
                    eng.optimize(1, True)
                    res["ok"] = bool(np.isfinite(eng.getOptimalControlSeq()).all())
                    res["why"] = "non-finite result after the first exchanged iteration"
                except Exception as e:  # noqa: BLE001
                    res["why"] = str(e)

            th = threading.Thread(target=attempt, daemon=True)
            th.start()
            th.join(120.0)
            stuck = th.is_alive()
            native_ok, why = (res["ok"] and not stuck), res["why"]
            if stuck:
                globals()["_HARD_EXIT"] = True  # a thread is parked inside the collective library: leave with os._exit
        flags = [None] * world
        dist.all_gather_object(flags, (native_ok, why))
        if all(f[0] for f in flags):
            exchange = "rccl all-gather of %d floats per rank per iteration (library-owned communicator)" % eng.exchangeBuffers()[2]
        else:
            # fall back to the host-staged exchange over the gloo group: slower, but independent of the collective library
            reason = next(f[1] for f in flags if not f[0])
            eng = make_engine(cfg, device=local_rank, rank=rank, world_size=world)
            hx = HostStagedExchange(eng)
            run = lambda n: (hx.iterate(n), eng.synchronize())  # noqa: E731
            exchange = "host-staged all-gather over gloo of %d floats per rank per iteration (native RCCL unavailable: %s)" % (
                eng.exchangeBuffers()[2], reason[:120])

    eng.uploadState(x0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = bool(np.isfinite(eng.getOptimalControlSeq()).all())

    # dominant-kernel duration with HIP events on the engine's stream (separate, untimed pass)
    n_ev = min(200, max(20, args.steps))
    try:
        ms_total, ms_roll = eng.timeIterations(n_ev)
        if ms_total == 0.0:  # host-staged fallback: the exchange is driven from here, use the wall-clock step
            ms_total = elapsed / args.steps * 1e3 * n_ev
    except Exception:  # noqa: BLE001
        ms_total = ms_roll = elapsed / args.steps * 1e3 * n_ev
    C_dim = eng.CONTROL_DIM
    b_alg = 4.0 * (2.0 * K_PER_GPU * T * C_dim + 2.0 * K_PER_GPU + 2.0 * T * C_dim)
    roll_us = ms_roll / n_ev * 1e3
    achieved = b_alg / (roll_us * 1e-6) / 1e9

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": "MPPI iters/sec (KxT rollouts)", "value": round(value, 3), "unit": "MPPI iters/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "Cartpole (CartpoleDynamics + CartpoleQuadraticCost, examples/cartpole_example.cu config) "
                            "VanillaMPPI optimisation iteration, K=16384 rollouts per GPU, T=100, dt=0.02, lambda=0.25, "
                            "sigma=5, Philox noise fused in the rollout kernel",
                "rollouts_per_gpu": K_PER_GPU, "global_rollouts": K_PER_GPU * world, "num_timesteps": T,
                "parallelism": "K-sharded x%d" % world, "exchange": exchange,
                "unit_definition": "one optimisation-loop body over K=16384 rollouts; value sums the units of all ranks",
            },
            "finite": ok,
            "roofline": {
                "bound": "hbm", "kernel": "rolloutPipelineKernel<CartpoleDynamics,CartpoleQuadraticCost,Gaussian,1,true>",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic("rolloutPipelineKernel<CartpoleDynamics"),
                "pipe_utilisation_pmc": pmc_pipe_util("rolloutPipelineKernel<CartpoleDynamics"),
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, " + PMC_FILE + " "
                                  "(2*FETCH_SIZE + WRITE_SIZE): the sample tensor never reaches HBM, so traffic << algorithmic bytes",
                "algorithmic_bytes_per_launch": b_alg, "avg_kernel_us": round(roll_us, 3),
                "avg_iteration_us_event_timed": round(ms_total / n_ev * 1e3, 3),
                "note": "issue-bound on the dynamics wave: T=100 dependent Euler steps per rollout at ~83 instructions each; K=16384 is 256 blocks of 4 role waves (2 samplers, dynamics, cost) on 256 CUs",
            },
        }
        # secondary workload of the north star (not the headline `value`): AutoRally-NN, K=16384, T=150, MFMA forward
        if not args.primary_only and world == 1:
            for key, leg in (("autorally_nn", autorally_leg), ("lstm_colored", lstm_colored_leg), ("di_tube", di_tube_leg)):
                try:
                    out[key] = leg(local_rank)
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": str(e)}
        if world == 1:
            # what a control loop sees: one mppi_compute_control call (inputs handed over from the host, one iteration,
            # smoothing + re-rollout of u*, results back) — the PCIe-inclusive figure, never `value`
            try:
                lat_eng = make_engine(cartpole_cfg(K=K_PER_GPU, T=T), device=local_rank)
                for _ in range(50):
                    lat_eng.computeControl(x0, 1)
                t_a = time.perf_counter()
                for _ in range(300):
                    lat_eng.computeControl(x0, 1)
                out["compute_control_latency_us"] = round((time.perf_counter() - t_a) / 300 * 1e6, 2)
                lat_eng.close()
            except Exception as e:  # noqa: BLE001
                out["compute_control_latency_us"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cartpole_cfg(K=K_PER_GPU, T=T))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    if globals().get("_HARD_EXIT"):
        sys.stdout.flush()
        os._exit(0)
