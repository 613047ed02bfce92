"""certify_zero (nerf_render_opts.certify_zero, ABI 4): a bf16 pass over all samples certifies those whose density pre-activation is
far below 0; the f32 kernel (nerf_mlp_kernel<.., MLP_MODE_LIST>; for the fine pass of a split arithmetic that arithmetic's kernel)
evaluates only the others.  A certified sample has sigma = 0 in the
f32 network too, hence weight 0 (src/lib.rs:271-272): the frame must be the plain f32 frame BIT FOR BIT -- which the whole-frame
fixture tests hold to Gate 1 against the oracle (tests/test_gpu_frame_fixture.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENE

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_c3_whole_frame_is_the_f32_frame(renderer, native, samples):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref, s0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, return_stats=True)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, certify_zero=True, return_stats=True)
    assert np.array_equal(img, ref)
    fc, ff = st.n_exec_coarse_trunk / st.n_coarse_points, st.n_exec_fine_trunk / st.n_fine_points
    print(f"\ncertify_zero: {s0.ms_total:.1f} -> {st.ms_total:.1f} ms; the f32 kernel evaluates {fc:.3f} of the coarse and {ff:.3f} of the fine samples")
    assert 0.2 < fc < 0.6 and 0.1 < ff < 0.35 and st.n_exec_colour == st.n_exec_fine_trunk
    assert st.ms_total < 0.6 * s0.ms_total


@pytest.mark.parametrize("nc,nf,W,crop,coarse_only,ssaa,pose_deg", [
    (64, 128, 800, (300, 300, 200, 64), False, 1, None), (40, 50, 800, (380, 360, 40, 24), False, 1, None), (64, 0, 800, (200, 200, 300, 100), True, 1, None),
    (33, 31, 800, (0, 0, 800, 8), False, 1, None), (64, 128, 800, (395, 400, 1, 1), False, 1, None), (64, 128, 128, None, False, 2, None),
    (64, 128, 200, None, False, 1, (130, 10)), (32, 64, 200, None, False, 1, (250, -20)), (3, 0, 64, None, True, 1, None)])
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
    print(f"\n{dtype} + certify_zero: {s0.ms_total:.1f} -> {st.ms_total:.1f} ms")
    assert np.array_equal(img, ref) and st.n_nonfinite_points == 0
    assert st.ms_total < 0.65 * s0.ms_total
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
