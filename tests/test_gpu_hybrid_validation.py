"""hybrid_sampling beyond the one view its error model was fitted on.  The flag of k_resample (sampling_kernels.hip:
a first-order bound of |d cdf_j| per bin edge from a density-error bound min(2e-5 + 6e-6 sigma, 2e-4) per sample, plus 6 ulp of
rounding noise, tau = 1e-5) was fitted on dumped lego views (round-3 tools, removed in round 4; the emulation lives on in tests/helpers/hybrid_model.py).
Here: rotated poses, other sample counts, a random-weight scene -- each held to the f32 path's own Gate 1 against
the LIVE oracle -- and the documented promise itself (include/nerf_mi355x.h: "the others move by <= 1e-5 in t"), asserted on the
sample positions: unflagged rays of the split-arithmetic densities must draw within 1e-5 of the f32 densities' draws.
Reference semantics: sample_importance, src/lib.rs:289-351."""
import os

import numpy as np
import pytest

from conftest import ROOT, SCENE, psnr

import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
from scene_utils import pose as _pose, oracle_samples as _oracle_samples, random_scene as _random_scene  # noqa: E402

pytestmark = pytest.mark.gpu


def _gate1(img, ref):
    d = np.abs(img - ref)
    assert d.max() <= 5e-4 and d.mean() <= 1e-5 and psnr(img, ref) >= 90.0, (d.max(), d.mean(), psnr(img, ref))


@pytest.mark.parametrize("deg,tilt,nc,nf", [(25, 0, 64, 128), (130, 0, 64, 128), (250, 12, 64, 128), (75, 0, 32, 64), (200, -10, 20, 50)])
def test_hybrid_gate1_other_poses_and_sample_counts(renderer, native, oracle, oracle_nets, samples, deg, tilt, nc, nf):
    m = _pose(samples, deg, tilt)
    cam = native.camera_from_pose(m, samples["hwf"], samples["near"], samples["far"], 96, 96, nc)
    ref = oracle.render_image(*oracle_nets, oracle.camera_from_samples(_oracle_samples(samples, m), 96, 96), oracle.make_opts(nc, nf, seed=3))
    assert 0.2 < np.all(ref == 1.0, axis=2).mean() < 0.95                     # the view shows the model and some background
    f32 = native.render_image(renderer.coarse, renderer.fine, cam, nf, seed=3)
    _gate1(f32, ref)
    for dt in ("f16x2", "bf16x3", "f32"):
        img, st = native.render_image(renderer.coarse, renderer.fine, cam, nf, seed=3, dtype=dt, skip_dead=True, hybrid_sampling=True,
                                      return_stats=True)
        d = np.abs(img - f32)
        print(f"\npose {deg}/{tilt} {nc}+{nf} {dt}: vs oracle max {np.abs(img - ref).max():.2e}; vs f32 frame max {d.max():.2e} mean {d.mean():.2e}; "
              f"redone {st.n_hybrid_rays / st.n_rays:.3f}")
        _gate1(img, ref)
        assert 0 < st.n_hybrid_rays < st.n_rays


def _fog_gate(img, ref):
    """Gate for the random-weight scene.  A random network is a FOG: densities vary smoothly, the coarse CDF is flat in many bins and
    hierarchical sampling is ill-conditioned nearly everywhere, so even the exact-f32 GPU path relocates a few samples against the
    CPU oracle (f32 MFMA = fmaf chain, oracle = mul then add; measured: max 1.5e-3 on 1 pixel of 4096, mean 3e-6, 89.5 dB).  The
    mean and the PSNR are held as in Gate 1 (psnr floor 85 dB); single-pixel outliers are bounded in number and size instead."""
    d = np.abs(img - ref)
    assert d.mean() <= 1e-5 and psnr(img, ref) >= 85.0 and d.max() <= 5e-3 and (d.max(axis=2) > 5e-4).mean() <= 2e-3, \
        (d.max(), d.mean(), psnr(img, ref), (d.max(axis=2) > 5e-4).mean())


def test_hybrid_random_weight_scene(native, oracle, samples, tmp_path):
    root = _random_scene(tmp_path / "rnd", 321)
    co, fi = oracle.Net(str(root / "coarse")), oracle.Net(str(root / "fine"))
    ref = oracle.render_image(co, fi, oracle.camera_from_samples(samples, 64, 64), oracle.make_opts(64, 128, seed=5))
    with native.Renderer(0) as r:
        coarse = native.load_network_from_dir(r, 0, root / "coarse")
        fine = native.load_network_from_dir(r, 1, root / "fine")
        cam = native.camera_from_samples(samples, 64, 64, 64)
        f32, s0 = native.render_image(coarse, fine, cam, 128, seed=5, skip_dead=True, return_stats=True)
        _fog_gate(f32, ref)
        assert np.array_equal(f32, native.render_image(coarse, fine, cam, 128, seed=5))
        assert s0.n_exec_colour > 0.05 * s0.n_fine_points and ref.std() > 0.01     # a live, non-trivial field
        for dt in ("f16x2", "bf16x3", "f32"):
            img, st = native.render_image(coarse, fine, cam, 128, seed=5, dtype=dt, skip_dead=True, hybrid_sampling=True, return_stats=True)
            d = np.abs(img - f32)
            print(f"\nrandom scene {dt}: vs oracle max {np.abs(img - ref).max():.2e} mean {np.abs(img - ref).mean():.2e}; vs f32 frame max {d.max():.2e} "
                  f"mean {d.mean():.2e}; redone {st.n_hybrid_rays / st.n_rays:.3f}")
            _fog_gate(img, ref)                                  # no worse against the oracle than the exact-f32 path itself ...
            assert d.mean() <= 2e-6 and (d.max(axis=2) > 5e-4).mean() <= 2e-3   # ... and the same picture as the f32 frame


def test_skip_dead_dense_fog_every_sample_live(native, samples, tmp_path):
    """A uniform fog of density ~2 keeps EVERY sample live (T stays above 1e-4 over the whole ray): all four waves stage 32 columns in
    every step of the f32 skip_dead kernel, colour passes run in the MIDDLE of the staging loop (a wave's samples do not fit) as well
    as at its end, back to back -- the paths the lego scene rarely takes.  Still the bits of the fused kernel."""
    root = _random_scene(tmp_path / "fog", 7, alpha_bias=(2.0, 2.0), alpha_scale=0.01)
    with native.Renderer(0) as r:
        coarse = native.load_network_from_dir(r, 0, root / "coarse")
        fine = native.load_network_from_dir(r, 1, root / "fine")
        cam = native.camera_from_samples(samples, 96, 96, 64)
        for crop, nf in (((0, 0, 96, 96), 128), ((3, 5, 61, 7), 128), ((10, 10, 37, 3), 50)):
            ref = native.render_image(coarse, fine, cam, nf, seed=2, crop=crop)
            img, st = native.render_image(coarse, fine, cam, nf, seed=2, crop=crop, skip_dead=True, return_stats=True)
            assert np.array_equal(img, ref), crop
            assert st.n_exec_colour > 0.9 * st.n_exec_fine_trunk and st.n_exec_fine_trunk > 0.9 * st.n_fine_points, st
        co = native.render_image(coarse, fine, cam, 0, seed=2, coarse_only=True, skip_dead=True)
        assert np.array_equal(co, native.render_image(coarse, fine, cam, 0, seed=2, coarse_only=True))


def _coarse_sigmas(native, renderer_net, cam, rdr, x0, y0, w, h, nc, seed, dtype):
    t = rdr.stage_stratified(cam, x0, y0, w, h, nc, seed=seed).reshape(-1, nc)
    dirs = rdr.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
    o = cam.pos.astype(np.float32)
    pts = (o[None, None, :] + dirs[:, None, :] * t[:, :, None]).astype(np.float32)   # mul and add rounded separately, as the kernels do (src/lib.rs:396)
    flat = pts.reshape(-1, 3)
    _, sg = renderer_net.forward_batch(np.ascontiguousarray(flat.T), np.repeat(dirs, nc, axis=0), dtype=dtype)
    return t, sg.reshape(-1, nc)


@pytest.mark.parametrize("case", ["json camera", "pose 130", "pose 250 tilted, 32+64", "random weights"])
def test_unflagged_rays_move_by_at_most_1e5_in_t(renderer, native, samples, tmp_path, case):
    """The documented bound, directly: resample the SAME rays from the f32 densities and from the split arithmetics' densities (same
    uniforms); every ray the error model does not flag must place each of its draws within 1e-5 (in t) of the f32 placement --
    flagged rays are the ones the pipeline redoes in f32.  Also: the model must not flag everything (it would be useless)."""
    nc, nf, W = 64, 128, 128
    rdr, net, ctx = renderer, renderer.coarse, None
    if case == "json camera":
        cam = native.camera_from_samples(samples, W, W, nc)
    elif case == "pose 130":
        cam = native.camera_from_pose(_pose(samples, 130), samples["hwf"], samples["near"], samples["far"], W, W, nc)
    elif case == "pose 250 tilted, 32+64":
        nc, nf = 32, 64
        cam = native.camera_from_pose(_pose(samples, 250, 12), samples["hwf"], samples["near"], samples["far"], W, W, nc)
    else:
        root = _random_scene(tmp_path / "rnd", 99)
        ctx = native.Renderer(0)
        rdr, net = ctx, native.load_network_from_dir(ctx, 0, root / "coarse")
        cam = native.camera_from_samples(samples, W, W, nc)
    try:
        x0, y0, w, h = 16, 16, 96, 96
        pix = ((y0 + np.arange(h))[:, None] * W + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
        far = float(samples["far"])
        t, s32 = _coarse_sigmas(native, net, cam, rdr, x0, y0, w, h, nc, 7, "f32")
        ref = rdr.stage_resample(t, s32, nf, far, seed=7, pixel_index=pix)["t_new"]
        for dt in ("f16x2", "bf16x3"):
            _, ssp = _coarse_sigmas(native, net, cam, rdr, x0, y0, w, h, nc, 7, dt)
            flags, tn = rdr.stage_hybrid_flags(t, ssp, nf, far, seed=7, pixel_index=pix)
            move = np.abs(tn - ref).max(axis=1)
            ds = np.abs(ssp - s32)
            print(f"\n{case} {dt}: |d sigma| max {ds.max():.2e} (rel {np.max(ds / (1 + np.abs(s32))):.2e}); flagged {flags.mean():.3f}; "
                  f"unflagged rays: max |dt| {move[~flags].max() if (~flags).any() else 0:.2e}; flagged rays: max |dt| {move[flags].max() if flags.any() else 0:.2e}")
            assert (~flags).any() and flags.mean() < 0.9
            assert move[~flags].max() <= 1e-5, (case, dt, float(move[~flags].max()), int((move[~flags] > 1e-5).sum()))
    finally:
        if ctx is not None:
            ctx.close()


def test_flag_bound_fuzz_short(renderer):
    """tools/fuzz_hybrid_flags.py for a few seconds (random poses, windows, sample counts, seeds; ~3 M rays): the bound rests on a
    statistical model of the density error, so a fuzz finds about one unflagged ray per million beyond 1e-5 -- all below 3e-5.  Held here:
    nothing beyond 5e-5, at most 5 per million beyond 1e-5.  (The fuzz is what found the two holes of the first per-edge model: a draw
    next to a light bin changing bins, and the quantisation of T behind a nearly opaque sample -- displacements up to 3.6e-2.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_hybrid_flags as fz
    res = fz.fuzz(renderer, 6.0, 20261004)
    print("\n", res)
    assert res["rays"] > 500_000 and 0.02 < res["flagged_fraction"] < 0.6
    assert fz.acceptable(res), res
