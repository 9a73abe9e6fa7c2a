"""The output layer's summation order of the two networks whose rollout forward runs on the matrix cores (AutoRally 6-32-32-4,
the bicycle LSTM's output net {22, 32, 4}).

The reference adds the 32 products of an output neuron in one chain, k ascending (utils/nn_helpers/fnn_helper.cu:458-462).
Here — in the engine's LDS, MFMA and one-rollout-per-wave forms and in the oracle alike (FNNHelper::split_output_sum_,
oracle_models.hpp: FNN::split_output_sum) — they are added as four interleaved chains, chain g = the inputs k with
(k >> 2) & 3 == g, combined (c0 + c1) + (c2 + c3): the order in which the MFMA form finds the previous layer's outputs in its
registers, which lets it skip two cross-lane transposes and 8 MFMAs that would use 4 of their 16 rows.  Same products, same set
of terms; only the ORDER of 31 additions per output differs.  This file pins what that does:
  * the reference's all-ones known answers (EXPECT_FLOAT_EQ, 4 ulp) hold in either order: exact in the reference's, 1 ulp in the split one;
  * against a float64 forward both orders are equally accurate (the split one slightly more: shorter chains);
  * one MPPI iteration at BASELINE scale: u* and the trajectory costs of the oracle in the two orders differ by far less than
    the 1e-5 parity bar — the same kind of bound tests/test_det_math.py gives for the transcendental flavour.
(HIP == oracle, bit for bit, is what tests/test_autorally.py, test_lstm.py and test_full_size_parity.py keep asserting — on the
split order, which is the one both sides evaluate.)"""
import numpy as np

import pyoracle as po
from common import autorally_cfg, bicycle_lstm_cfg, host_noise, make_oracle, ulp_diff


def test_all_ones_known_answers_hold_in_both_orders():
    """tests/nn_helpers/fnn_helper_test.cu:403-446: all parameters and inputs 1 -> 33 (6-32-32-4)"""
    n = 6 * 32 + 32 + 32 * 32 + 32 + 32 * 4 + 4
    # EXPECT_FLOAT_EQ there, i.e. 4 ulp: the hidden values are tanh(7) and tanh(33), not 1, so "33" is a rounded sum — exact
    # in the reference's order, one ulp below in the split one; both inside the reference's own tolerance
    out = po.fnn_forward([6, 32, 32, 4], np.ones(n), np.ones(6), split_output_sum=False)
    assert np.all(out == 33.0), out
    out = po.fnn_forward([6, 32, 32, 4], np.ones(n), np.ones(6), split_output_sum=True)
    assert ulp_diff(out, np.full(4, 33.0, np.float32)).max() <= 1, out


def test_both_orders_against_float64():
    rng = np.random.default_rng(11)
    layers = [6, 32, 32, 4]
    worst = {False: 0.0, True: 0.0}
    differ = 0
    for _ in range(200):
        theta = rng.uniform(-0.5, 0.5, 6 * 32 + 32 + 32 * 32 + 32 + 32 * 4 + 4).astype(np.float32)
        x = rng.uniform(-2, 2, 6).astype(np.float32)
        t = theta.astype(np.float64)
        W1, b1 = t[:192].reshape(32, 6), t[192:224]
        W2, b2 = t[224:1248].reshape(32, 32), t[1248:1280]
        W3, b3 = t[1280:1408].reshape(4, 32), t[1408:]
        ref = W3 @ np.tanh(W2 @ np.tanh(W1 @ x.astype(np.float64) + b1) + b2) + b3
        outs = {}
        for split in (False, True):
            outs[split] = po.fnn_forward(layers, theta, x, split_output_sum=split)
            worst[split] = max(worst[split], float(np.abs(outs[split] - ref).max()))
        differ += int(not np.array_equal(outs[False], outs[True]))
        assert np.abs(outs[False] - outs[True]).max() <= 2e-6
    assert differ > 20                      # the flag does something
    assert worst[True] <= 3e-6 and worst[False] <= 3e-6
    assert worst[True] <= 1.5 * worst[False]  # not less accurate than the reference's order


def _one_iteration(cfg, split, eps):
    o = make_oracle(cfg)
    o.set_split_output_sum(split)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    return o.control().copy(), o.costs().copy()


def test_effect_on_one_iteration_autorally_and_lstm():
    """K = 4096 rollouts of the BASELINE horizons: the two orders' u* within 1e-6 (bar: 1e-5), costs within 1e-5 relative"""
    for cfg in (autorally_cfg(K=4096, T=150, lambda_=1.0), bicycle_lstm_cfg(K=2048, T=200, lambda_=1.0)):
        eps = host_noise(1, cfg["K"], cfg["T"], 2)
        u_a, c_a = _one_iteration(cfg, False, eps)
        u_b, c_b = _one_iteration(cfg, True, eps)
        assert not np.array_equal(c_a, c_b)
        du = float(np.abs(u_a - u_b).max())
        dc = float((np.abs(c_a - c_b) / np.maximum(np.abs(c_a), 1.0)).max())
        print(cfg["model"], "u* L-inf between the orders", du, "costs rel", dc)
        assert du <= 1e-6, (cfg["model"], du)
        assert dc <= 1e-5, (cfg["model"], dc)


def _kat_theta():
    return np.ones(6 * 32 + 32 + 32 * 32 + 32 + 32 * 4 + 4, np.float32)


def test_reference_bias_and_weight_edit_known_answers_pin_the_blob_layout():
    """tests/nn_helpers/fnn_helper_test.cu:403-446, ALL THREE stages with the reference's literal inputs (input = 0):
    all-ones -> 33; output biases theta[1408..1411] = 2, 3, 4, 5 -> 34, 35, 36, 37; first weight of each output row
    theta[1280], [1312], [1344], [1376] = 2 -> 35, 36, 37, 38.  The second and third stages pin the LAYOUT of the blob
    ([W row-major out x in][b] per layer: output weights start at 1280 with a row stride of 32, biases at 1408) on a vector
    the reference holds — an oracle that transposed W3 or put the biases first would miss them (by >= 1.0, not by an ulp).
    The bar is the reference's own EXPECT_FLOAT_EQ (4 ulp): the hidden values are det::tanh(25.4) = 1 - 2^-24, not the 1.0f of
    CUDA's tanhf, so a sum that is not all ones may land an ulp below the integer; where the reference's order does hit the
    integer exactly (the first two stages) that is asserted too."""
    layers, x = [6, 32, 32, 4], np.zeros(6, np.float32)
    for split, tol in ((False, 0), (True, 4)):
        theta = _kat_theta()
        out = po.fnn_forward(layers, theta, x, split_output_sum=split)
        assert ulp_diff(out, np.full(4, 33.0, np.float32)).max() <= tol, (split, out)
        theta[1408:1412] = [2.0, 3.0, 4.0, 5.0]
        out = po.fnn_forward(layers, theta, x, split_output_sum=split)
        assert ulp_diff(out, np.array([34, 35, 36, 37], np.float32)).max() <= tol, (split, out)
        theta[[1280, 1312, 1344, 1376]] = 2.0
        out = po.fnn_forward(layers, theta, x, split_output_sum=split)
        assert ulp_diff(out, np.array([35, 36, 37, 38], np.float32)).max() <= 4, (split, out)
    # the edits address what the docstring says they address: a weight edit one column further scales hidden unit 1 instead
    # (tanh(33) rounds to 1.0f, so the answer is the same) while an edit one ROW off (index 1280 + 4) moves output 0 again, not 1
    theta = _kat_theta()
    theta[1284] = 3.0
    out = po.fnn_forward(layers, theta, x, split_output_sum=False)
    assert ulp_diff(out, np.array([35, 33, 33, 33], np.float32)).max() <= 4, out


def test_ar_dynamics_all_ones_with_the_reference_inputs_and_the_edits():
    """tests/dynamics/ar_dynamics_nn_test.cu:445-481: s = 0, u = (1, -1), all-ones network -> xdot = (0, 0, 0, 33, 33, 33, 33),
    state and control untouched; then the same bias / weight edits as the helper's test through the MODEL (they must land in
    xdot[3..6] in the same order: output i of the network is state derivative 3 + i, ar_nn_model.cu:160-166)."""
    cfg = autorally_cfg(K=64, T=4)
    for split, tol in ((False, 0), (True, 4)):
        o = make_oracle(cfg)
        o.set_split_output_sum(split)
        theta = _kat_theta()
        o.set_blob("dynamics_weights", theta)
        s, u = np.zeros(7, np.float32), np.array([1.0, -1.0], np.float32)
        xd = o.state_deriv(s, u)
        assert np.array_equal(xd[:3], np.zeros(3, np.float32))
        assert ulp_diff(xd[3:], np.full(4, 33.0, np.float32)).max() <= tol, (split, xd)
        assert np.array_equal(s, np.zeros(7, np.float32)) and np.array_equal(u, np.array([1.0, -1.0], np.float32))
        theta[1408:1412] = [2.0, 3.0, 4.0, 5.0]
        o.set_blob("dynamics_weights", theta)
        xd = o.state_deriv(s, u)
        assert ulp_diff(xd[3:], np.array([34, 35, 36, 37], np.float32)).max() <= tol, (split, xd)
        theta[[1280, 1312, 1344, 1376]] = 2.0
        o.set_blob("dynamics_weights", theta)
        xd = o.state_deriv(s, u)
        assert ulp_diff(xd[3:], np.array([35, 36, 37, 38], np.float32)).max() <= 4, (split, xd)
