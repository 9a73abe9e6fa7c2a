"""Generic 2-D texture helper (include/mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp).
The oracle restates the reference's host lookup (two_d_texture_helper.cu:151-245) and is pinned on the known answers of
tests/texture_helpers/two_d_texture_helper_test.cu (QueryTextureAtMapPose :368-441, QueryTextureAtWorldPose :451-541);
the HIP helper must agree with it bit for bit."""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po

# normalised query points of the reference tests and the .x channel they expect (float4 texel i = (i, i+1, i+2, i+3),
# 10 wide x 20 high, resolution 10)
QUERY = [(0.0, 0.0), (0.05, 0.0), (0.95, 0.0), (1.0, 0.0), (0.45, 0.0), (0.5, 0.0), (0.55, 0.0), (0.0, 0.0), (0.0, 0.025),
         (0.0, 0.05), (0.0, 0.075), (0.0, 0.975), (0.0, 1.0), (0.0, 0.475), (0.0, 0.5), (0.0, 0.525)]
EXPECT = [0.0, 0.0, 9.0, 9.0, 4.0, 4.5, 5.0, 0.0, 0.0, 5.0, 10.0, 190.0, 190.0, 90.0, 95.0, 100.0]
W, H, RES = 10, 20, 10.0


def texels():
    i = np.arange(W * H, dtype=np.float32)
    return np.stack([i, i + 1, i + 2, i + 3], 1).reshape(H, W, 4)


def test_oracle_reproduces_reference_map_pose_kat():
    pts = np.array([[x * RES * W, y * RES * H, 0.0] for x, y in QUERY], np.float32)
    got = po.texture2d_query(texels(), pts, 1, resolution=(RES, RES, RES))
    for g, e in zip(got, EXPECT):
        assert np.allclose(g, [e, e + 1, e + 2, e + 3], rtol=0, atol=2e-4), (g, e)


def test_oracle_reproduces_reference_world_pose_kat():
    """rotation swaps x and y, origin (1, 2, 3) (two_d_texture_helper_test.cu:467-473, 503-517)"""
    pts = np.array([[y * RES * H + 1, x * RES * W + 2, 3.0] for x, y in QUERY], np.float32)
    got = po.texture2d_query(texels(), pts, 2, origin=(1, 2, 3), rotations=(0, 1, 0, 1, 0, 0, 0, 0, 1),
                             resolution=(RES, RES, RES))
    for g, e in zip(got, EXPECT):
        assert np.allclose(g, [e, e + 1, e + 2, e + 3], rtol=0, atol=2e-4), (g, e)


def test_oracle_point_filter_and_border():
    data = np.arange(12, dtype=np.float32).reshape(3, 4)
    centres = np.array([[(c + 0.5) / 4, (r + 0.5) / 3, 0] for r in range(3) for c in range(4)], np.float32)
    assert np.array_equal(po.texture2d_query(data, centres, 0)[:, 0], data.reshape(-1))  # exact at the cell centres
    assert np.array_equal(po.texture2d_query(data, centres, 0, filter_mode=1)[:, 0], data.reshape(-1))
    out = np.array([[-0.2, 0.5, 0], [0.5, 1.3, 0], [0.5, 0.5, 0]], np.float32)
    got = po.texture2d_query(data, out, 0, address_mode=(1, 1), border_color=(-7, 0, 0, 0))[:, 0]
    assert got[0] == -7 and got[1] == -7 and got[2] != -7
    clamp = po.texture2d_query(data, out, 0)[:, 0]
    assert clamp[0] == 0.5 * (data[1, 0] + data[1, 0]) and np.isfinite(clamp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("channels,address,filt", [(1, (0, 0), 0), (4, (0, 0), 0), (1, (1, 1), 0), (4, (0, 1), 1)])
def test_device_helper_equals_oracle_bitwise(gpu, channels, address, filt):
    rng = np.random.default_rng(channels * 10 + filt)
    h, w = 37, 53
    data = rng.standard_normal((h, w, channels)).astype(np.float32)
    th = 0.7
    rot = (np.cos(th), -np.sin(th), 0, np.sin(th), np.cos(th), 0, 0, 0, 1)
    origin, res = (3.0, -2.0, 0.5), (0.25, 0.2, 1.0)
    pts = rng.uniform(-6, 16, (20000, 3)).astype(np.float32)
    for frame in (0, 1, 2):
        q = (pts / np.array([16, 16, 1], np.float32)) if frame == 0 else pts
        want = po.texture2d_query(data, q, frame, origin=origin, rotations=rot, resolution=res, address_mode=address,
                                  filter_mode=filt, border_color=(9, 8, 7, 6))
        p = m.MppiTexture2dParams(origin, rot, res, address, filt, (9, 8, 7, 6))
        got = m.texture2d_query(data, q, frame, p)
        assert np.array_equal(got, want), (frame, np.abs(got - want).max())


@pytest.mark.gpu
def test_device_helper_reference_kat(gpu):
    pts = np.array([[y * RES * H + 1, x * RES * W + 2, 3.0] for x, y in QUERY], np.float32)
    p = m.MppiTexture2dParams((1, 2, 3), (0, 1, 0, 1, 0, 0, 0, 0, 1), (RES, RES, RES))
    got = m.texture2d_query(texels(), pts, 2, p)
    for g, e in zip(got, EXPECT):
        assert np.allclose(g, [e, e + 1, e + 2, e + 3], rtol=0, atol=2e-4)
