"""The oracle pinned against the reference's own golden vectors (lego_rust/tf_reference_samples.json ==
the literals of the reference's only unit test, src/lib.rs:759-847)."""
import numpy as np
import pytest

from conftest import golden

# Stated fp32 tolerance at the forward_batch seam; the reference's own test uses 1e-2 (src/lib.rs:732-742)
SIGMA_TOL = 1e-4   # |dsigma| <= SIGMA_TOL * (1 + |sigma|)
RGB_TOL = 1e-5


def _example_inputs(samples, ex):
    origin = np.float32(samples["camera_origin"]); z = np.float32(samples["z_vals"])
    rd = np.float32(ex["ray_d"]); vd = np.float32(ex["viewdir_unit"])
    pts = (origin[:, None] + rd[:, None] * z[None, :]).astype(np.float32)  # origin + ray_dir * t, un-normalised (src/lib.rs:855)
    return pts, np.tile(vd, (len(z), 1))


def test_120_golden_scalars(oracle, samples, oracle_nets):
    n = 0
    for ex in samples["examples"]:
        pts, dirs = _example_inputs(samples, ex)
        for net, ks, kr in ((oracle_nets[0], "coarse_sigma", "coarse_rgb"), (oracle_nets[1], "fine_sigma", "fine_rgb")):
            rgb, sg = net.forward_batch(pts, dirs)
            es, er = np.float32(ex[ks]), np.float32(ex[kr])
            assert np.all(np.abs(sg - es) <= SIGMA_TOL * (1 + np.abs(es)))
            assert np.all(np.abs(rgb - er) <= RGB_TOL)
            n += es.size + er.size
    assert n == 120


def test_reference_unit_test_literals(oracle, oracle_nets):
    """coarse_and_fine_match_reference_examples (src/lib.rs:753-916), example 0, with its own 1e-2 tolerance."""
    origin = np.float32([-0.053798322, 3.8454704, 1.2080823])
    ray_dir = np.float32([0.013345719, -0.95394367, -0.2996883]); view = np.float32([0.013345721, -0.9539438, -0.29968834])
    t = np.float32([2, 3, 4, 5, 6])
    pts = (origin[:, None] + ray_dir[:, None] * t[None, :]).astype(np.float32)
    rgb, sg = oracle_nets[0].forward_batch(pts, np.tile(view, (5, 1)))
    assert np.allclose(sg, [0.0, 0.0, 57.520237, 112.53807, 36.354565], atol=1e-2)
    assert np.allclose(rgb[1], [0.17455171, 0.1438829, 0.089204505], atol=1e-2)
    rgb, sg = oracle_nets[1].forward_batch(pts, np.tile(view, (5, 1)))
    assert np.allclose(sg, [0.0, 0.0, 116.636, 213.10716, 0.0], atol=1e-2)
    assert np.allclose(rgb[3], [0.8538921, 0.77298534, 0.6327022], atol=1e-2)


def test_naive_loop_order_is_bit_identical(oracle, oracle_nets):
    """forward_fallback's loop nest (src/network.rs:134-143) and the cache-blocked nest give the same bits."""
    g = golden("forward_batch_4096.npz")
    pts, dirs = g["pts"][:, :300], g["dirs"][:300]
    a = oracle_nets[1].forward_batch(pts, dirs)
    b = oracle_nets[1].forward_batch(pts, dirs, naive=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_forward_batch_fixture(oracle, oracle_nets):
    g = golden("forward_batch_4096.npz")
    for name, net in (("coarse", oracle_nets[0]), ("fine", oracle_nets[1])):
        rgb, sg = net.forward_batch(g["pts"], g["dirs"])
        assert np.array_equal(rgb, g[f"{name}_rgb"]) and np.array_equal(sg, g[f"{name}_sigma"])
    assert oracle_nets[0].forward_batch(np.zeros((3, 0), np.float32), np.zeros((0, 3), np.float32))[1].shape == (0,)


def test_positional_encoding_order(oracle):
    """[x,y,z] then per octave sin(x,y,z), cos(x,y,z), f doubling (src/network.rs:263-292)."""
    import ctypes as C
    p = np.float32([[0.3], [-1.1], [2.0]])
    enc = np.zeros((63, 1), np.float32)
    oracle.lib().oracle_positional_encoding(oracle._p(p), 1, 10, oracle._p(enc))
    assert np.array_equal(enc[:3, 0], p[:, 0])
    for o in range(10):
        f = np.float32(2.0 ** o)
        assert np.allclose(enc[3 + 6 * o: 6 + 6 * o, 0], np.sin(np.float64(f * p[:, 0])), atol=1e-6)
        assert np.allclose(enc[6 + 6 * o: 9 + 6 * o, 0], np.cos(np.float64(f * p[:, 0])), atol=1e-6)


def test_camera_anchor_json_rays(oracle, samples):
    """JSON pixel = [row, col]; its ray_d is the TF ray WITHOUT the half-pixel offset: get_ray_dir with the
    +0.5 removed reproduces it, with +0.5 (src/lib.rs:221-222) it is ~9e-4 off."""
    cam = oracle.camera_from_samples(samples, 400, 400)
    for ex in samples["examples"]:
        i, j = ex["pixel"]
        d0 = oracle.get_ray_dir(cam, i, j, half=False)
        assert np.abs(d0 - np.float32(ex["ray_d"])).max() < 5e-7
        assert np.abs(oracle.normalize(d0) - np.float32(ex["viewdir_unit"])).max() < 5e-7
        d1 = oracle.get_ray_dir(cam, i, j, half=True)
        assert 1e-4 < np.abs(d1 - np.float32(ex["ray_d"])).max() < 2e-3
    assert abs(np.tan(cam.alpha_width) - 0.36) < 1e-6


def test_philox_known_answer(oracle):
    """Philox-4x32-10 known-answer vectors (Random123 kat_vectors): counter/key all zero and all ones."""
    assert oracle.philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    u = [oracle.uniform(7, 123, 1, k) for k in range(64)]
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) > 60
