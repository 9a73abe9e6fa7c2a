#!/usr/bin/env python3
"""Where the role waves of rolloutPipelineRepKernel spend a launch (AutoRally-NN, K=16384, T=150) — a thread-trace substitute
built from s_memtime stamps inside the kernel (A/B build with -DMPPI_PIPE_TIMING, never a product build).

  python mppi-generic_amd/buildlib.py --variant timing autorally_nn.hip -DMPPI_PIPE_TIMING            (CPU)
  MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_timing.so python tools/pipe_timing.py [out.json]   (GPU box)

Per wave of the first 8 blocks: s_memtime ticks per launch in each category (the counter runs at the shader clock).  Dynamics waves: work / waiting for the sampler /
waiting for the cost waves (ring back-pressure).  Cost waves: evaluation inside the relay chain / waiting for the dynamics
waves / waiting for the relay / fetching the pair's outputs.  Sampler waves: the whole loop."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import autorally_cfg, make_engine  # noqa: E402

BLOCKS, WAVES, SLOTS = 256, 24, 8  # PIPE_TIMING_BLOCKS / _WAVES / _SLOTS of rollout_pipeline_kernel.hpp


def main():
    lib = C.CDLL(m.library_path())
    cfg = autorally_cfg(K=16384, T=150, lambda_=1.0)
    eng = make_engine(cfg)
    rep = eng.num_rollouts_local and getattr(eng, "block_y", None)
    eng.uploadState(cfg["x0"])
    eng.optimize(20)
    tot, roll = eng.timeIterations(50)
    eng.optimize(1)
    buf = (C.c_ulonglong * (BLOCKS * WAVES * SLOTS))()
    n = lib.mppi_debug_read_pipe_timing(buf, len(buf))
    assert n == len(buf), n
    t = np.frombuffer(buf, np.uint64).reshape(BLOCKS, WAVES, SLOTS).astype(np.float64)
    used = [w for w in range(WAVES) if t[:, w].sum() > 0]
    # PIPE_REP_SAMPLERS / PIPE_REP_COSTS of rollout_pipeline_kernel.hpp (A/B builds: PIPE_NS / PIPE_NC): dynamics waves, then
    # samplers, then cost waves
    ns, nc = int(os.environ.get("PIPE_NS", "2")), int(os.environ.get("PIPE_NC", "2"))
    dw = len(used) - ns - nc
    ticks_per_us = float(t[:, :dw, :3].sum(axis=2).mean()) / (roll / 50 * 1e3)  # a dynamics wave is busy for the whole launch
    out = {"workload": "AutoRally-NN K=16384 T=150, rolloutPipelineRepKernel, %d dynamics + %d sampler + %d cost waves per block"
                       % (dw, ns, nc),
           "rollout_kernel_us_hip_events": round(roll / 50 * 1e3, 2), "blocks_sampled": BLOCKS,
           "unit": "s_memtime ticks per launch and wave (the counter runs at the shader clock on gfx950: ~%.0f ticks per us "
                   "of this launch), mean over the sampled blocks" % ticks_per_us}
    dyn = t[:, :dw]
    out["dynamics_wave"] = {"work": round(dyn[:, :, 0].mean()), "wait_sampler": round(dyn[:, :, 1].mean()),
                            "wait_cost_ring": round(dyn[:, :, 2].mean()), "work_slowest_wave": round(dyn[:, :, 0].max()),
                            "work_per_wave": [round(x) for x in dyn[:, :, 0].mean(axis=0)]}
    smp = t[:, dw:dw + ns]
    out["sampler_wave"] = {"loop": round(smp[:, :, 0].mean())}
    x = t[:, dw + ns:dw + ns + nc]
    out["cost_wave"] = {"evaluation_ahead_of_relay": round(x[:, :, 0].mean()), "wait_dynamics": round(x[:, :, 1].mean()),
                        "wait_relay": round(x[:, :, 2].mean()), "fetch_pair": round(x[:, :, 3].mean()),
                        "re_evaluation_in_relay": round(x[:, :, 4].mean()), "relay": round(x[:, :, 5].mean())}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
