"""The .npz reader and loaders (mppi_load_npz): the on-disk format of the reference's networks and costmaps
(FNNHelper::loadParams fnn_helper.cu:96-174, LSTMHelper::loadParams lstm_helper.cu:514-585, ARStandardCost::loadTrackData
ar_standard_cost.cu:84-142).  The archives in /root/reference/resources are git-LFS pointer files, so the fixtures are
generated the way the reference's own scripts generate theirs (scripts/autorally/test/generateTestNetwork.py,
generateTestMaps.py)."""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import autorally_cfg, bicycle_lstm_cfg, host_noise, lstm_npz, make_engine, make_oracle, standard_track_map, ulp_diff


def _autorally_npz(rng):
    layers = [6, 32, 32, 4]
    d = {}
    for i in range(1, 4):
        d["dynamics_W%d" % i] = rng.uniform(-0.3, 0.3, (layers[i], layers[i - 1])).astype(np.float64)
        d["dynamics_b%d" % i] = rng.uniform(-0.3, 0.3, layers[i]).astype(np.float64)
    return d


def _track_npz():
    cmap, (x0, x1, y0, y1) = standard_track_map()
    return {"xBounds": np.array([x0, x1], np.float32), "yBounds": np.array([y0, y1], np.float32),
            "pixelsPerMeter": np.array([20.0], np.float32), "channel0": cmap.reshape(-1),
            "channel1": np.zeros(cmap.size, np.float32), "channel2": np.zeros(cmap.size, np.float32),
            "channel3": np.zeros(cmap.size, np.float32)}


# ------------------------------------------------------------------ CPU: the reader itself -----------------------------
@pytest.mark.parametrize("compressed", [False, True])
def test_reader_matches_numpy(tmp_path, compressed, lib):
    rng = np.random.default_rng(0)
    d = {"f8": rng.normal(size=(5, 7)), "f4": rng.normal(size=(3, 4, 2)).astype(np.float32),
         "i8": np.arange(12, dtype=np.int64).reshape(3, 4), "i4": np.arange(6, dtype=np.int32), "scalar1": np.array([2.5]),
         "fortran": np.asfortranarray(rng.normal(size=(4, 3))), "model/lstm/weight_hh_l0": rng.normal(size=(8, 2))}
    path = tmp_path / "a.npz"
    (np.savez_compressed if compressed else np.savez)(path, **d)
    for k, v in d.items():
        got = m.npz_read_array(path, k)
        assert got.shape == v.shape and np.array_equal(got, v.astype(np.float64)), k
    with pytest.raises(m.MPPIError) as e:
        m.npz_read_array(path, "missing")
    assert "no key" in str(e.value)


def test_reader_reports_lfs_stub_and_garbage(tmp_path, lib):
    stub = tmp_path / "autorally_nnet_09_12_2018.npz"
    stub.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0000\nsize 12345\n")
    with pytest.raises(m.MPPIError) as e:
        m.npz_read_array(stub, "dynamics_W1")
    assert "LFS" in str(e.value)
    with pytest.raises(m.MPPIError):
        m.npz_read_array(tmp_path / "nope.npz", "x")
    obj = tmp_path / "obj.npz"
    np.savez(obj, o=np.array([{"a": 1}], dtype=object))
    with pytest.raises(m.MPPIError) as e:
        m.npz_read_array(obj, "o")
    assert "object" in str(e.value)


# ------------------------------------------------------------------ GPU: loaders == blob path ---------------------------
@pytest.mark.gpu
def test_load_npz_autorally_equals_blob_path(gpu, tmp_path):
    rng = np.random.default_rng(42)
    net, trk = _autorally_npz(rng), _track_npz()
    np.savez(tmp_path / "net.npz", **net)
    np.savez_compressed(tmp_path / "track.npz", **trk)
    cfg = autorally_cfg(K=512, T=40)  # same seed 42 -> the same synthetic network as _autorally_npz
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    ref = make_engine(cfg)
    ref.injectNoise(eps)
    want = ref.rolloutCosts(cfg["x0"], 1)
    eng = m.VanillaMPPIController("autorally_nn", cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
    eng.setCostParams(m.ARStandardCostParams())  # identity transform: loadNpz("costmap") must install the real one
    eng.loadNpz("dynamics", tmp_path / "net.npz")
    eng.loadNpz("costmap", tmp_path / "track.npz")
    eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.injectNoise(eps)
    assert ulp_diff(eng.rolloutCosts(cfg["x0"], 1), want).max() == 0
    with pytest.raises(m.MPPIError) as e:
        eng.loadNpz("dynamics", tmp_path / "track.npz")
    assert e.value.status == 1 and "dynamics_W1" in str(e.value)
    with pytest.raises(m.MPPIError):
        eng.loadNpz("lstm", tmp_path / "net.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("prefix,model_dir", [(None, ""), ("steering", "model/")])
def test_load_npz_lstm_equals_blob_path_and_oracle(gpu, tmp_path, prefix, model_dir):
    d = lstm_npz()
    pre = model_dir + (prefix + "/" if prefix else "")
    np.savez(tmp_path / "lstm.npz", **{pre + k: v for k, v in d.items()})
    np.savez(tmp_path / "track.npz", **_track_npz())
    cfg = bicycle_lstm_cfg(K=512, T=40)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    eng = m.VanillaMPPIController("bicycle_slip_lstm", cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
    eng.setCostParams(m.ARStandardCostParams())
    eng.loadNpz("lstm", tmp_path / "lstm.npz", prefix)
    eng.loadNpz("costmap", tmp_path / "track.npz")
    eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.injectNoise(eps)
    got = eng.rolloutCosts(cfg["x0"], 1)
    orc = make_oracle(cfg)
    mean = np.zeros((1, cfg["T"], 2), np.float32)
    v = orc.set_gaussian_controls(mean, eps[0], 1, 0)
    want, _ = orc.rollout_costs(cfg["x0"], mean, v)
    assert ulp_diff(got, want).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [((4, 20), (4, 20), (4, 20)), ((4, 20), (6, 10), (5, 12))], ids=["reference-test-shapes", "other-shapes"])
def test_load_npz_three_networks_of_the_racer_uncertainty_model(gpu, tmp_path, sizes):
    """one archive with the prefixes the reference's RacerDubinsElevationLSTMUncertainty(path) reads (steering/model/,
    terra/mean_network/, terra/uncertainty_network/; racer_dubins_elevation_lstm_unc.cu:30-33) through mppi_load_npz kinds
    "lstm" / "mean_lstm" / "unc_lstm" == the same networks through the blob path.  The loader sizes the networks from the
    archive as the reference does; other shapes than the reference's test shapes run one lane per rollout"""
    from test_racer_dubins_lstm_unc import uncertainty_cfg
    (hs, ms), (hm, mm), (hu, mu) = sizes
    nets = {"steering/model/": lstm_npz(seed=1, scale=0.06, I=4, H=hs, M=ms, OUT=1),
            "terra/mean_network/": lstm_npz(seed=2, scale=0.06, I=12, H=hm, M=mm, OUT=2),
            "terra/uncertainty_network/": lstm_npz(seed=3, scale=0.06, I=13, H=hu, M=mu, OUT=5)}
    np.savez(tmp_path / "rde.npz", **{pre + k: v for pre, d in nets.items() for k, v in d.items()})
    cfg = uncertainty_cfg(K=512, T=30)
    general = sizes != ((4, 20), (4, 20), (4, 20))
    shape = dict(block_x=64, block_y=1) if general else {}
    blobs = dict(cfg["blobs"])
    for stem, pre, (H, M), I, OUT in (("lstm", "steering/model/", sizes[0], 4, 1), ("mean_lstm", "terra/mean_network/", sizes[1], 12, 2),
                                      ("unc_lstm", "terra/uncertainty_network/", sizes[2], 13, 5)):
        lstm_blob, out_blob = m.lstm_blob_from_npz_dict(nets[pre])
        blobs[stem + "_weights"], blobs[stem + "_output_weights"] = lstm_blob, out_blob
        if general:
            blobs[stem + "_structure"] = np.array([H, H + I, M, OUT], np.float32)
    cfg["blobs"] = dict(sorted(blobs.items(), key=lambda kv: 0 if kv[0].endswith("_structure") else 1))
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    a = make_engine(cfg, **shape)
    a.injectNoise(eps)
    want = a.rolloutCosts(cfg["x0"], 1)
    # the second engine gets the networks from the archive only: shapes included
    cfg_b = dict(cfg)
    cfg_b["blobs"] = {k: v for k, v in cfg["blobs"].items() if "lstm" not in k}
    b = make_engine(cfg_b, **shape)
    b.loadNpz("lstm", tmp_path / "rde.npz", "steering/model")
    b.loadNpz("mean_lstm", tmp_path / "rde.npz", "terra/mean_network")
    b.loadNpz("unc_lstm", tmp_path / "rde.npz", "terra/uncertainty_network/")
    b.injectNoise(eps)
    got = b.rolloutCosts(cfg["x0"], 1)
    assert np.isfinite(want).all() and ulp_diff(got, want).max() == 0
    orc = make_oracle(cfg)
    mean = np.zeros((1, cfg["T"], 2), np.float32)
    v = orc.set_gaussian_controls(mean, eps[0], 1, 0)
    ref, _ = orc.rollout_costs(cfg["x0"], mean, v)
    assert ulp_diff(got, ref).max() == 0
