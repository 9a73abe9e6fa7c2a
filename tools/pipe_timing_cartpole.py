#!/usr/bin/env python3
"""Where the role waves of rolloutPipelineKernel<Cartpole> spend a launch (K=16384, T=100: the BASELINE headline) — in-kernel
s_memtime stamps per role wave (A/B build with -DMPPI_PIPE_TIMING, never a product build).

  python mppi-generic_amd/buildlib.py --variant timing_cp cartpole.hip -DMPPI_PIPE_TIMING                        (CPU)
  MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_timing_cp.so python tools/pipe_timing_cartpole.py [out.json]  (GPU box)

Waves of a block: 0 sampler (trips 0, 2, ..), 1 dynamics, 2 cost, 3 second sampler (trips 1, 3, ..).  Slots per wave (ticks per
launch): 0 work in the role loop, 1 / 2 waiting (dynamics: for the sampler / for the cost ring; cost: for the dynamics),
3 kernel entry -> role loop, 4 own loop done -> whole block done (barrier), 5 block-softmin epilogue; 6 / 7: s_memrealtime
(100 MHz, chip-wide) at entry / exit — the launch ramp and the tail across the 256 blocks."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import cartpole_cfg, di_cfg, make_engine  # noqa: E402

BLOCKS, WAVES, SLOTS = 256, 24, 8


def main():
    lib = C.CDLL(m.library_path())
    args = [a for a in sys.argv[1:] if a != "di_tube"]
    di = "di_tube" in sys.argv[1:]  # the two-system form (32 rollouts x 2 systems per wave) on the double integrator, K=8192, T=150
    sys.argv[1:] = args
    cfg = di_cfg(K=8192, T=150, tube=True) if di else cartpole_cfg(K=16384, T=100)
    eng = make_engine(cfg)
    eng.uploadState(cfg["x0"])
    eng.optimize(50)
    n_it = 200
    tot, roll = eng.timeIterations(n_it)
    eng.optimize(2)  # the stamps of the LAST launch: the second one merges the first one's records in its sampler waves
    buf = (C.c_ulonglong * (BLOCKS * WAVES * SLOTS))()
    n = (lib.mppi_debug_read_pipe_timing_double_integrator if di else lib.mppi_debug_read_pipe_timing_cartpole)(buf, len(buf))
    assert n == len(buf), n
    t = np.frombuffer(buf, np.uint64).reshape(BLOCKS, WAVES, SLOTS).astype(np.float64)
    kernel_us = roll / n_it * 1e3
    dyn, smp, cost = t[:, 1], t[:, [0, 3]], t[:, 2]
    # a dynamics wave's slots 0..5 cover its whole life inside the kernel; entry -> exit in real time calibrates the tick
    life_ticks = dyn[:, :6].sum(axis=1)
    life_us = (dyn[:, 7] - dyn[:, 6]) / 100.0
    ticks_per_us = float(np.median(life_ticks / np.maximum(life_us, 1e-3)))
    entry = t[:, 1, 6]
    exit_ = t[:, 1, 7]
    t0 = entry.min()

    def us(x):
        return round(float(x) / ticks_per_us, 3)

    out = {
        "workload": ("Double integrator, Tube (2 systems folded into the lanes) K=8192 T=150" if di else "Cartpole K=16384 T=100") +
                    ", rolloutPipelineKernel: per block 1 dynamics + 2 sampler + 1 cost wave; 256 blocks",
        "rollout_kernel_us_hip_events_instrumented_build": round(kernel_us, 2),
        "iteration_us_instrumented_build": round(tot / n_it * 1e3, 2),
        "s_memtime_ticks_per_us": round(ticks_per_us, 1),
        "unit": "microseconds per launch and wave, mean over the 256 blocks (converted from s_memtime ticks)",
        "dynamics_wave": {"entry_to_loop": us(dyn[:, 3].mean()), "work_all_steps": us(dyn[:, 0].mean()),
                          "wait_sampler": us(dyn[:, 1].mean()), "wait_cost_ring": us(dyn[:, 2].mean()),
                          "loop_done_to_block_done": us(dyn[:, 4].mean()), "epilogue": us(dyn[:, 5].mean()),
                          "work_slowest_block": us(dyn[:, 0].max()), "work_fastest_block": us(dyn[:, 0].min())},
        "sampler_waves": {"entry_to_loop": us(smp[:, :, 3].mean()), "loop": us(smp[:, :, 0].mean()),
                          "loop_done_to_block_done": us(smp[:, :, 4].mean()), "epilogue": us(smp[:, :, 5].mean())},
        "cost_wave": {"entry_to_loop": us(cost[:, 3].mean()), "work": us(cost[:, 0].mean()),
                      "wait_dynamics": us(cost[:, 1].mean()), "loop_done_to_block_done": us(cost[:, 4].mean()),
                      "epilogue": us(cost[:, 5].mean())},
        "across_blocks_real_time_us": {"entry_first_to_last": round(float(entry.max() - t0) / 100.0, 2),
                                       "entry_median_after_first": round(float(np.median(entry) - t0) / 100.0, 2),
                                       "exit_first": round(float(exit_.min() - t0) / 100.0, 2),
                                       "exit_last": round(float(exit_.max() - t0) / 100.0, 2),
                                       "block_life_mean": round(float(life_us.mean()), 2)},
    }
    # per trip of four steps (dynamics wave): rows 4..7 wait for the sampler, 8..11 wait for the cost ring, 12..15 work
    trips = 25
    per = lambda k: t[:, 4 + 4 * k:8 + 4 * k].reshape(BLOCKS, -1)[:, :trips]
    out["dynamics_wave_per_trip_us"] = {
        "wait_sampler": [us(v) for v in per(0).mean(axis=0)],
        "wait_cost_ring": [us(v) for v in per(1).mean(axis=0)],
        "work_4_steps": [us(v) for v in per(2).mean(axis=0)],
        "sampler_wave_0_its_trips": [us(v) for v in per(3).mean(axis=0)[:13]],
    }
    # streamed merge: the stages of sampler wave 0's first trip, microseconds since kernel entry (zero when the launch did not stream)
    st = per(4).mean(axis=0)
    out["sampler_wave_0_first_trip_since_entry_us"] = {
        "record loads issued": us(st[0]), "draw done": us(st[1]), "tails merged (rho, scales, eta)": us(st[2]),
        "first four columns of the mean merged": us(st[3]), "samples shaped and stored": us(st[4])}
    d = out["dynamics_wave"]
    out["dynamics_work_share_of_block_life"] = round(d["work_all_steps"] / out["across_blocks_real_time_us"]["block_life_mean"], 3)
    out["dynamics_work_share_of_launch"] = round(d["work_all_steps"] / kernel_us, 3)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
