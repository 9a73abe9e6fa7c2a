"""Edge sizes through every form of the elevation-map RACER models against the oracle: a single rollout, partial waves and
blocks, horizons shorter than / not a multiple of the four replica lanes' group of steps, one and two systems.
(tools/racer_shape_sweep.py is the randomised version.)"""
import numpy as np
import pytest

from common import host_noise, make_engine, make_oracle, ulp_diff
from test_racer_dubins_elevation import elevation_cfg
from test_racer_dubins_lstm_steering import steering_cfg
from test_racer_dubins_lstm_unc import uncertainty_cfg
from test_racer_dubins_suspension import suspension_cfg

MODELS = {"elevation": elevation_cfg, "lstm_steering": steering_cfg, "suspension": suspension_cfg, "complete": uncertainty_cfg}
CASES = [  # K, T, systems, lanes per rollout, kernel variant (0 auto, 1 fused, 2 role-pipelined)
    (1, 1, 1, 4, 0), (1, 5, 2, 4, 1), (3, 7, 1, 4, 2), (17, 2, 1, 4, 1), (65, 3, 2, 4, 0), (63, 9, 1, 4, 2),
    (100, 1, 1, 1, 1), (1000, 6, 2, 1, 1), (1049, 13, 1, 4, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("model", list(MODELS))
def test_racer_edge_sizes_bit_exact(gpu, model):
    for K, T, D, by, variant in CASES:
        cfg = MODELS[model](K=K, T=T, D=D)
        eps = host_noise(1, K, T, 2, seed=K + T)
        o = make_oracle(cfg)
        (o.tube_compute_control if D == 2 else o.vanilla_compute_control)(cfg["x0"], 1, eps)
        shape = {} if (by == 4 and D == 2) else dict(block_x=64, block_y=by)  # two systems on four lanes: the default shape
        eng = make_engine(cfg, kernel_variant=variant, **shape)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
        assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0, (model, K, T, D, by, variant)
        assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5, (model, K, T, D, by, variant)
        eng.close()
