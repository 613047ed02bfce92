"""Oracle stage functions (a1-a3, a9-a14): regression against the committed fixtures (tests/golden/make_golden.py) plus the
semantic properties the reference's code implies (src/lib.rs:176-351).  No known-answer test exists for these in the
reference -- the fixtures pin the restatement, not the reference."""
import os

import numpy as np

from conftest import golden


def test_ray_dirs_fixture(oracle, samples):
    g = golden("ray_dirs.npz")
    for n in (400, 800):
        cam = oracle.camera_from_samples(samples, n, n)
        for k, (i, j) in enumerate(g[f"pix{n}"]):
            d = oracle.get_ray_dir(cam, int(i), int(j))
            assert np.array_equal(d, g[f"dir{n}"][k])
            assert np.array_equal(oracle.normalize(d), g[f"dirhat{n}"][k])
            assert abs(np.linalg.norm(g[f"dirhat{n}"][k].astype(np.float64)) - 1) < 1e-6


def test_stratified_properties(oracle):
    """t_k = lower + (upper - lower) * jitter inside stratum k, ascending (src/lib.rs:239-245)."""
    g = golden("ray_stages_800.npz")
    near, far, seed = float(g["near"]), float(g["far"]), int(g["seed"])
    for r, pix in enumerate(g["pixel_index"]):
        t = oracle.stratified_samples(seed, int(pix), near, far, 64)
        assert np.array_equal(t, g["t_coarse"][r])
        edges = near + np.arange(65, dtype=np.float32) * np.float32((far - near) / 64)
        assert np.all(t >= edges[:-1] - 1e-6) and np.all(t <= edges[1:] + 1e-6) and np.all(np.diff(t) > 0)
    assert oracle.stratified_samples(0, 1, near, far, 0).shape == (0,)
    assert not np.array_equal(oracle.stratified_samples(0, 1, near, far, 64), oracle.stratified_samples(1, 1, near, far, 64))


def test_compute_weights_fixture_and_cut(oracle):
    g = golden("ray_stages_800.npz")
    far = float(g["far"])
    for r in range(len(g["pixels"])):
        assert np.array_equal(oracle.compute_weights(g["sigma_coarse"][r], g["t_coarse"][r], far), g["w_coarse"][r])
        w = oracle.compute_weights(g["sigma_fine"][r], g["t_merged"][r], far)
        assert np.array_equal(w, g["w_fine"][r])
        assert 0 <= w.sum() <= 1 + 1e-5
    # early termination at T < 1e-4: the terminating sample keeps its weight, everything after is exactly 0
    t = np.linspace(2, 6, 8, endpoint=False).astype(np.float32)
    s = np.float32([0, 0, 50, 50, 50, 50, 50, 50])
    w = oracle.compute_weights(s, t, 6.0)
    T = np.cumprod(1 - (1 - np.exp(-s * 0.5)))
    cut = int(np.argmax(T < 1e-4))
    assert np.all(w[cut + 1:] == 0) and w[cut] > 0 and abs(w.sum() - (1 - T[cut])) < 1e-5
    # negative delta clamps to zero (src/lib.rs:267-269); last delta is far - t (not 1e10)
    assert oracle.compute_weights(np.float32([9, 9]), np.float32([3, 2]), 6.0)[0] == 0
    assert abs(oracle.compute_weights(np.float32([1.0]), np.float32([5.0]), 6.0)[0] - (1 - np.exp(-1.0))) < 1e-6


def test_sample_importance_fixture(oracle):
    g = golden("ray_stages_800.npz")
    seed = int(g["seed"])
    for r, pix in enumerate(g["pixel_index"]):
        u = np.float32([oracle.uniform(seed, int(pix), 1, k) for k in range(128)])
        assert np.array_equal(u, g["u_fine"][r])
        tn, cdf = oracle.sample_importance_u(u, g["t_coarse"][r], g["w_coarse"][r])
        assert np.array_equal(tn, g["t_new"][r]) and np.array_equal(cdf[:63], g["cdf"][r])
        assert cdf[0] == 0 and cdf[62] == 1 and np.all(np.diff(cdf[:63]) > 0)
        bins = 0.5 * (g["t_coarse"][r][1:] + g["t_coarse"][r][:-1])
        assert np.all(tn >= bins[0] - 1e-6) and np.all(tn <= bins[-1] + 1e-6)
        merged = oracle.sort_ascending(np.concatenate([g["t_coarse"][r], tn]))
        assert np.array_equal(merged, g["t_merged"][r]) and np.all(np.diff(merged) >= 0)
        assert np.array_equal(oracle.sample_importance(seed, int(pix), g["t_coarse"][r], g["w_coarse"][r], 128), tn)


def test_sample_importance_guards(oracle):
    """count == 0 or fewer than 3 samples -> no extra samples (src/lib.rs:295-307)."""
    t = np.float32([2, 3, 4]); w = np.float32([0.1, 0.5, 0.2])
    assert len(oracle.sample_importance(0, 0, t, w, 0)) == 0
    assert len(oracle.sample_importance(0, 0, t[:2], w[:2], 8)) == 0
    out = oracle.sample_importance(0, 0, t, w, 8)  # one bin [2.5, 3.5]
    assert len(out) == 8 and np.all((out >= 2.5) & (out <= 3.5))
    # a dominant bin attracts the draws
    t = np.linspace(2, 6, 64, endpoint=False).astype(np.float32); w = np.zeros(64, np.float32); w[30] = 0.9
    u = (np.arange(128, dtype=np.float32) + 0.5) / 128
    tn, _ = oracle.sample_importance_u(u, t, w)
    lo, hi = 0.5 * (t[29] + t[30]), 0.5 * (t[30] + t[31])
    assert np.mean((tn >= lo) & (tn <= hi)) > 0.95


def test_integrate_ray_fixture(oracle):
    g = golden("ray_stages_800.npz")
    far = float(g["far"])
    for r in range(len(g["pixels"])):
        assert np.array_equal(oracle.integrate_ray(g["rgb_fine"][r], g["sigma_fine"][r], g["t_merged"][r], far), g["rgb"][r])
    assert np.all(g["rgb"][g["is_empty"]] == 1.0)          # empty ray composites to the white background
    assert g["is_terminated"].any()
    assert np.array_equal(oracle.integrate_ray(np.zeros((0, 3)), np.zeros(0), np.zeros(0), far), np.zeros(3, np.float32))


def test_render_matches_ray_debug_and_fixtures(oracle, samples, oracle_nets):
    """render_image's block scheduler (src/lib.rs:503-557) places every ray where render_ray_debug says, independent
    of the crop / block decomposition; a band of the C3 crop reproduces the committed image."""
    g = golden("crop_c3_800_64_128.npz")
    cam = oracle.camera_from_samples(samples, 800, 800)
    x0, y0, w, h = (int(v) for v in g["crop"])
    band = oracle.render_image(*oracle_nets, cam, oracle.make_opts(64, 128, crop=(x0 + 3, y0 + 8, 11, 9), seed=0))
    assert np.array_equal(band, g["image"][8:17, 3:14])
    d = oracle.render_ray_debug(*oracle_nets, cam, oracle.make_opts(64, 128, seed=0), y0 + 10, x0 + 5)
    assert np.array_equal(d["rgb"], g["image"][10, 5])
    st = golden("ray_stages_800.npz")
    d0 = oracle.render_ray_debug(*oracle_nets, cam, oracle.make_opts(64, 0, seed=0), *map(int, st["pixels"][1]))
    assert np.array_equal(d0["rgb"], st["nofine_rgb"][1]) and d0["n_new"] == 0


def test_quantizer_and_ppm(oracle, tmp_path):
    """save_ppm (src/lib.rs:567-580): (clamp(v,0,1)*255+0.5) as u8, P6 header."""
    v = np.float32([[-1, 0, 0.001], [0.5, 1.0, 2.0], [0.49803922, 0.999, np.nan]])
    q = oracle.quantize_rgb8(v)
    assert q.tolist() == [[0, 0, 0], [128, 255, 255], [127, 255, 0]]
    img = np.linspace(0, 1, 2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    p = tmp_path / "o.ppm"
    assert oracle.save_ppm(p, img) == 0
    raw = p.read_bytes()
    assert raw.startswith(b"P6\n3 2\n255\n") and len(raw) == 11 + 18
    assert list(raw[11:]) == oracle.quantize_rgb8(img).reshape(-1).tolist()
