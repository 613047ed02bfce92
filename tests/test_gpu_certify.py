"""certify_zero (nerf_render_opts.certify_zero, ABI 5): a 16-bit pass over all samples (f16 where the network fits its range, else bf16) (Z) certifies those whose density pre-activation is
far below 0 and (C) predicts each ray's T < 1e-4 cut (src/lib.rs:276-279); the exact kernel (nerf_mlp_kernel<.., MLP_MODE_LIST>; for the
fine pass of a split arithmetic that arithmetic's kernel) evaluates only the other samples in front of the predicted cut, the exact
transmittance confirms the cut.  A certified sample has sigma = 0 in the exact network too, hence weight 0 (src/lib.rs:271-272), a
sample behind the cut has weight 0 whatever its density: the frame must be the plain frame BIT FOR BIT -- which the whole-frame
fixture tests hold to Gate 1 against the oracle (tests/test_gpu_frame_fixture.py).  (Z) is audited in every frame (1 certified sample in 16 near the margin,
1 in 128 below it, is evaluated exactly all the same): a network on which the margins are too tight must widen them, certify nothing, or fail --
never return a silently different frame."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENE

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOORS_F16 = (0.25, 0.5)   # nerf_internal.h kCertMargin*F16: the pre-filter runs in f16 (lego: every weight and activation inside the f16 range)
FLOORS_BF16 = (1.0, 3.0)   # kCertMarginCoarse / Fine: the bf16 pre-filter (f32's range)


def test_c3_whole_frame_is_the_f32_frame(renderer, native, samples):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref, s0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, return_stats=True)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, certify_zero=True, return_stats=True)
    assert np.array_equal(img, ref)
    fc, ff = st.n_exec_coarse_trunk / st.n_coarse_points, st.n_exec_fine_trunk / st.n_fine_points
    print(f"\ncertify_zero: {s0.ms_total:.1f} -> {st.ms_total:.1f} ms (coarse {st.ms_coarse_mlp:.1f}, fine {st.ms_fine_mlp:.1f}); the f32 kernel evaluates {fc:.3f} of "
          f"the coarse and {ff:.3f} of the fine samples; audited {st.n_certify_audited}, violations {st.n_certify_violations}, headroom "
          f"{st.certify_headroom} at margins {st.certify_margin}, {st.n_certify_fallback_rays} of {st.n_rays} rays fell back, {st.n_certify_retries} retries")
    # work fractions only (timing ratios are bench output: a throttled box must not turn a correctness suite red)
    assert 0.05 < fc < 0.45 and 0.08 < ff < 0.3 and 0.6 * st.n_exec_fine_trunk < st.n_exec_colour < st.n_exec_fine_trunk   # probable zeros and audited certificates skip the colour head tile-wise
    assert st.n_certify_retries == 0 and st.n_certify_violations == 0 and st.certify_margin == FLOORS_F16
    certified = st.n_coarse_points + st.n_fine_points - st.n_exec_coarse_trunk - st.n_exec_fine_trunk
    # audited: 1 in 16 of the samples certified by less than twice the margin, 1 in 128 of the others (in front of the predicted cuts)
    assert certified / 128 * 0.8 < st.n_certify_audited < certified / 16
    assert all(h >= 0.5 * m for h, m in zip(st.certify_headroom, st.certify_margin))
    assert st.n_certify_fallback_rays < 0.01 * st.n_rays


def test_c5_geometry_three_passes(renderer, native, samples):
    """Config C5's geometry (800 x 800 output, 2 x 2 SSAA = 2.56 M rays: three passes of whole ray rows) in the f16x2 arithmetic: the certified frame is the
    plain f16x2 frame bit for bit (bench.py reports its rate beside the bf16 rows: extra_c5_bf16_ssaa2.same_geometry_at_f32_accuracy_...)."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=3, dtype="f16x2", ssaa=2)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=3, dtype="f16x2", ssaa=2, certify_zero=True, return_stats=True)
    assert np.array_equal(img, ref)
    assert st.n_passes >= 2 and st.n_certify_violations == 0 and st.n_rays == 4 * 800 * 800


@pytest.mark.parametrize("nc,nf,W,crop,coarse_only,ssaa,pose_deg", [
    (64, 128, 800, (300, 300, 200, 64), False, 1, None), (40, 50, 800, (380, 360, 40, 24), False, 1, None), (64, 0, 800, (200, 200, 300, 100), True, 1, None),
    (33, 31, 800, (0, 0, 800, 8), False, 1, None), (64, 128, 800, (395, 400, 1, 1), False, 1, None), (64, 128, 128, None, False, 2, None),
    (64, 128, 200, None, False, 1, (130, 10)), (32, 64, 200, None, False, 1, (250, -20)), (3, 0, 64, None, True, 1, None),
    (300, 500, 400, (180, 190, 40, 12), False, 1, None), (70, 1000, 400, (150, 200, 24, 6), False, 1, None)])   # 800 / 1070 samples per ray: several 256-sample batches per ray in k_cert_plan
def test_windows_sample_counts_poses(renderer, native, samples, nc, nf, W, crop, coarse_only, ssaa, pose_deg):
    if pose_deg is None:
        cam = native.camera_from_samples(samples, W, W, nc)
    else:
        from scene_utils import pose
        cam = native.camera_from_pose(pose(samples, *pose_deg), samples["hwf"], samples["near"], samples["far"], W, W, nc)
    kw = dict(seed=5, crop=crop, coarse_only=coarse_only, ssaa=ssaa)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, nf, **kw)
    assert np.array_equal(native.render_image(renderer.coarse, renderer.fine, cam, nf, certify_zero=True, **kw), ref)


def test_through_bands_and_option_checks(renderer, native, samples):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (250, 300, 300, 61)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=2, crop=crop)
    rs = [native.Renderer(0) for _ in range(2)]
    try:
        for r in rs:
            r.load_scene(SCENE)
        for gather in ("host", "rccl"):
            assert np.array_equal(native.render_image_multi(rs, cam, 128, gather=gather, seed=2, crop=crop, certify_zero=True), ref)
    finally:
        for r in rs:
            r.close()
    for bad in (dict(skip_dead=True), dict(skip_empty=True), dict(dtype="bf16")):
        with pytest.raises(native.NerfError, match="certify_zero needs"):
            native.render_image(renderer.coarse, renderer.fine, cam, 128, crop=(0, 0, 8, 8), certify_zero=True, **bad)


@pytest.mark.parametrize("dtype", ["f16x2", "bf16x3"])
def test_split_arithmetics_whole_frame(renderer, native, samples, dtype):
    """The split arithmetics keep their exact-f32 sampling pass (certified as well) and run the fine pass on the certified list: the
    frame is the plain frame of that arithmetic bit for bit -- which test_gpu_frame_fixture.py holds to the unrelaxed Gate 1 against
    the oracle's whole frame -- and no operand leaves the arithmetic's range."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref, s0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype=dtype, return_stats=True)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype=dtype, certify_zero=True, return_stats=True)
    print(f"\n{dtype} + certify_zero: {s0.ms_total:.1f} -> {st.ms_total:.1f} ms; fine list {st.n_exec_fine_trunk / st.n_fine_points:.3f}, headroom {st.certify_headroom}")
    assert np.array_equal(img, ref) and st.n_nonfinite_points == 0
    assert st.n_certify_violations == 0 and st.n_certify_retries == 0 and st.n_exec_fine_trunk < 0.3 * st.n_fine_points
    co = native.render_image(renderer.coarse, renderer.fine, cam, 0, seed=0, dtype=dtype, coarse_only=True, crop=(300, 300, 200, 50))
    assert np.array_equal(native.render_image(renderer.coarse, renderer.fine, cam, 0, seed=0, dtype=dtype, coarse_only=True, crop=(300, 300, 200, 50),
                                              certify_zero=True), co)


def test_certify_fuzz_short(renderer):
    """tools/fuzz_certify.py for a few seconds (random poses, windows, sample counts, seeds, SSAA, coarse-only): bit-identical
    throughout (a 4-minute run: 43 500 cases, 126 M rays, 0 mismatching)."""
    import fuzz_certify
    res = fuzz_certify.fuzz(renderer, 6.0, 20261004)
    print("\n", res)
    assert res["cases"] > 100 and res["mismatching"] == 0 and res["f32_samples_evaluated"] < 0.6 * res["f32_samples_nominal"], res


def test_many_passes(native, samples, monkeypatch):
    """Small passes (NERF_MAX_RAYS_PER_PASS is read when a context is created): the list and its device-side length are per pass."""
    monkeypatch.setenv("NERF_MAX_RAYS_PER_PASS", "3000")
    with native.Renderer(0) as r:
        r.load_scene(SCENE)
        cam = native.camera_from_samples(samples, 800, 800, 64)
        crop = (300, 350, 200, 100)
        ref = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop)
        img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop, certify_zero=True, return_stats=True)
        assert st.n_passes >= 7 and np.array_equal(img, ref)
        assert 0 < st.n_exec_fine_trunk < 0.9 * st.n_fine_points


def _scaled_lego(tmp_path, name, scale):
    """The lego networks with dense7 (kernel and bias) of both scaled by `scale`: relu is positively homogeneous, so the trunk output and
    with it every density pre-activation (but for the alpha bias) grow by that factor -- and so do the bf16 pass's errors."""
    import shutil
    root = tmp_path / name
    for which in ("coarse", "fine"):
        shutil.copytree(os.path.join(SCENE, which), root / which)
        for t in ("dense7_kernel", "dense7_bias"):
            f = root / which / f"{t}.bin"
            (np.fromfile(f, "<f4") * np.float32(scale)).astype("<f4").tofile(f)
    return root


def _load(native, r, root):
    r.coarse = native.load_network_from_dir(r, 0, os.path.join(root, "coarse"))
    r.fine = native.load_network_from_dir(r, 1, os.path.join(root, "fine"))


def test_hot_network_widens_its_margins_instead_of_differing(native, samples, tmp_path):
    """Pre-activations 40 x the lego networks': the 16-bit pass is off by several units on zero-density samples, the shipped margins certify
    samples whose exact density is positive.  The audit must see it (violations, or headroom below half the margin), widen the
    margins and render again: the frame returned is the plain frame of that network, and the next frame starts from the widened margins."""
    with native.Renderer(0) as r:
        _load(native, r, _scaled_lego(tmp_path, "hot", 40.0))
        cam = native.camera_from_samples(samples, 400, 400, 64)
        crop = (100, 120, 200, 160)
        ref = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop)
        img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop, certify_zero=True, return_stats=True)
        print(f"\nhot network: {st.n_certify_retries} retries, {st.n_certify_violations} violations seen, margins {st.certify_margin}, headroom {st.certify_headroom}")
        assert np.array_equal(img, ref)
        assert st.n_certify_retries >= 1 and all(m > 2 * f for m, f in zip(st.certify_margin, FLOORS_F16))
        assert all(h >= 0.5 * m for h, m in zip(st.certify_headroom, st.certify_margin))
        img2, st2 = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop, certify_zero=True, return_stats=True)
        # the next frame starts from the widened margins (another seed's audit may widen them further, never back)
        assert all(b >= a for a, b in zip(st.certify_margin, st2.certify_margin)) and st2.n_certify_retries <= 2
        assert np.array_equal(img2, native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop))
        _load(native, r, SCENE)  # loading a network resets its margin
        _, st3 = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop, certify_zero=True, return_stats=True)
        assert st3.certify_margin == FLOORS_F16 and st3.n_certify_retries == 0 and st3.n_certify_violations == 0


def test_network_beyond_the_f16_range_falls_back_to_the_bf16_prefilter(native, samples, tmp_path):
    """dense0 x 2000, dense1 x 50: activations beyond 65 504 -- the f16 pre-filter's pre-activations come out non-finite there (never certified, and
    counted); render_device goes back to the bf16 pre-filter for that network, for good, and renders the frame again.  The frame returned is the
    plain f32 frame; the margins are the bf16 pass's from then on (at least its floors: such a network's errors widen them further)."""
    import shutil
    root = tmp_path / "big"
    shutil.copytree(SCENE, root)
    for which in ("coarse", "fine"):
        for t, k in (("dense0_kernel", 2000.0), ("dense0_bias", 2000.0), ("dense1_kernel", 50.0), ("dense1_bias", 50.0)):
            f = root / which / f"{t}.bin"
            (np.fromfile(f, "<f4") * np.float32(k)).astype("<f4").tofile(f)
    with native.Renderer(0) as r:
        _load(native, r, root)
        cam = native.camera_from_samples(samples, 400, 400, 64)
        crop = (100, 120, 200, 100)
        ref = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop)
        try:
            img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop, certify_zero=True, return_stats=True)
        except native.NerfError as e:   # eight widenings may not reach a network this far out: loud, never silently different
            print("\nbeyond the f16 range:", e.msg)
            assert e.code == -6 and "certify_zero" in e.msg
        else:
            print(f"\nbeyond the f16 range: {st.n_certify_retries} retries, margins {st.certify_margin}, lists {st.n_exec_coarse_trunk / st.n_coarse_points:.3f} / "
                  f"{st.n_exec_fine_trunk / st.n_fine_points:.3f}")
            assert np.array_equal(img, ref)
            assert st.n_certify_retries >= 1 and all(m >= f for m, f in zip(st.certify_margin, FLOORS_BF16))
            img2, st2 = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop, certify_zero=True, return_stats=True)
            assert np.array_equal(img2, native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop))
            assert all(m >= f for m, f in zip(st2.certify_margin, FLOORS_BF16))   # no way back to f16 until the network is loaded again
        _load(native, r, SCENE)
        _, st3 = native.render_image(r.coarse, r.fine, cam, 128, seed=4, crop=crop, certify_zero=True, return_stats=True)
        assert st3.certify_margin == FLOORS_F16 and st3.n_certify_retries == 0


def test_uncertifiable_network_fails_loudly_or_is_exact(native, samples, tmp_path):
    """Pre-activations 1e6 x lego's: eight widenings (x 4 each at most... from 3 to 2e5) may not reach the bf16 error.  Whatever happens, a
    frame that is returned is the plain frame; otherwise the render fails with NERF_ERR_STATE and says why."""
    with native.Renderer(0) as r:
        _load(native, r, _scaled_lego(tmp_path, "wild", 1.0e6))
        cam = native.camera_from_samples(samples, 400, 400, 64)
        crop = (150, 150, 100, 60)
        ref = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop)
        try:
            img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=3, crop=crop, certify_zero=True, return_stats=True)
        except native.NerfError as e:
            print("\nwild network:", e.msg)
            assert e.code == -6 and "certify_zero" in e.msg and "certify_zero = 0" in e.msg
        else:
            print(f"\nwild network: {st.n_certify_retries} retries, margins {st.certify_margin}")
            assert np.array_equal(img, ref) and st.n_certify_retries >= 1


@pytest.mark.parametrize("kw", [dict(), dict(alpha_bias=(2.0, 2.0), alpha_scale=0.01), dict(alpha_bias=(0.0, 0.0), alpha_scale=0.02)])
def test_random_weight_fogs(native, samples, tmp_path, kw):
    """Random-weight networks (fogs: pre-activations near 0, nothing to certify; with alpha bias 2 every sample is dense and the cut comes
    early on every ray -- the bf16-predicted cut against the exact one on smooth densities): identical frames; the sample list starts at
    half of a pass's samples, so a scene that lists everything must take the enlarge-and-render-again path once and only once."""
    from scene_utils import random_scene
    root = random_scene(tmp_path / "rnd", 321, **kw)
    with native.Renderer(0) as r:
        _load(native, r, root)
        W = 400 if not kw else 128   # 400 x 400 x 192 samples: beyond the size up to which the list simply holds every sample
        cam = native.camera_from_samples(samples, W, W, 64)
        ref = native.render_image(r.coarse, r.fine, cam, 128, seed=5)
        img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=5, certify_zero=True, return_stats=True)
        print(f"\nfog {kw}: lists {st.n_exec_coarse_trunk / st.n_coarse_points:.3f} / {st.n_exec_fine_trunk / st.n_fine_points:.3f}, {st.n_certify_retries} retries, "
              f"{st.n_certify_fallback_rays} of {st.n_rays} rays fell back, audited {st.n_certify_audited}, violations {st.n_certify_violations}, margins {st.certify_margin}")
        assert np.array_equal(img, ref)
        assert st.n_certify_retries == (1 if W == 400 else 0)   # the list grew once (a fog lists every sample), no margin moved
        img2, st2 = native.render_image(r.coarse, r.fine, cam, 128, seed=6, certify_zero=True, return_stats=True)
        assert np.array_equal(img2, native.render_image(r.coarse, r.fine, cam, 128, seed=6))
        assert st2.n_certify_retries == 0 and st2.certify_margin == st.certify_margin == FLOORS_F16
