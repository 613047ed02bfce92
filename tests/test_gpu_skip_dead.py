"""skip_dead (the exact part of dead-work skipping): rays are walked front to back in chunks of 32 samples and
retired at the reference's T < 1e-4 cut (src/lib.rs:276-279: every later weight is exactly 0); the colour head runs only on
samples whose weight is > 0.  Nothing that reaches a pixel changes, so every image must be BIT-identical to the non-skipping
render -- and is additionally held to Gate 1 against the oracle's fixtures.  nerf_stats.n_exec_* must show that work really
was removed."""
import os

import numpy as np
import pytest

from conftest import SCENE, golden, psnr

pytestmark = pytest.mark.gpu


def _gate1(img, ref):
    d = np.abs(img - ref)
    assert d.max() <= 5e-4 and d.mean() <= 1e-5 and psnr(img, ref) >= 90.0, (d.max(), d.mean(), psnr(img, ref))


def test_skip_dead_crop_is_bit_exact_and_meets_gate1(renderer, native, samples):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    g = golden("crop_c3_800_64_128.npz")
    crop = tuple(int(v) for v in g["crop"])
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop, skip_dead=True, return_stats=True)
    assert np.array_equal(img, ref)
    _gate1(img, g["image"])
    n = crop[2] * crop[3]
    assert st.n_rays == n and st.n_coarse_points == 64 * n and st.n_fine_points == 192 * n
    # this window looks at the model: rays are cut short, and only a fraction of the samples carries weight
    assert 0.3 * st.n_fine_points < st.n_exec_fine_trunk < 0.98 * st.n_fine_points and st.n_exec_fine_trunk % 32 == 0
    assert 0 < st.n_exec_colour < 0.7 * st.n_fine_points
    assert st.n_exec_coarse_trunk <= st.n_coarse_points and st.n_exec_coarse_trunk % 32 == 0
    assert st.n_colour_skipped_points == st.n_fine_points - st.n_exec_colour
    again = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop, skip_dead=True)
    assert np.array_equal(again, img)                          # the dynamic ray queue does not leak into the result


def test_skip_dead_variants_are_bit_exact(renderer, native, samples):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    R = lambda **kw: native.render_image(renderer.coarse, renderer.fine, kw.pop("cam", cam), kw.pop("nf", 128), **kw)  # noqa: E731
    crop = (250, 300, 300, 37)
    assert np.array_equal(R(seed=5, crop=crop, skip_dead=True), R(seed=5, crop=crop))
    # pure background: every ray runs all chunks (T stays 1), no colour work at all, exactly white
    corner, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=3, crop=(0, 0, 64, 16), skip_dead=True, return_stats=True)
    assert np.all(corner == 1.0) and st.n_exec_colour == 0 and st.n_exec_fine_trunk == st.n_fine_points
    # coarse-only (BASELINE C1): the coarse network's colours are composited -> trunk + colour split on the coarse net
    g = golden("crop_c1_400_coarse_only.npz")
    cam4 = native.camera_from_samples(samples, 400, 400, 64)
    c1 = tuple(int(v) for v in g["crop"])
    img, st = native.render_image(renderer.coarse, renderer.fine, cam4, 0, seed=0, coarse_only=True, crop=c1, skip_dead=True, return_stats=True)
    assert np.array_equal(img, native.render_image(renderer.coarse, renderer.fine, cam4, 0, seed=0, coarse_only=True, crop=c1))
    _gate1(img, g["image"])
    assert 0 < st.n_exec_colour < st.n_coarse_points and st.n_exec_fine_trunk == 0
    # sample counts that are not multiples of the 32-sample chunk (ragged last chunk), n_fine = 0, SSAA
    cam40 = native.camera_from_samples(samples, 800, 800, 40)
    assert np.array_equal(R(cam=cam40, nf=50, seed=2, crop=(380, 360, 40, 24), skip_dead=True), R(cam=cam40, nf=50, seed=2, crop=(380, 360, 40, 24)))
    cam7 = native.camera_from_samples(samples, 800, 800, 7)
    assert np.array_equal(R(cam=cam7, nf=5, seed=2, crop=(380, 360, 40, 24), skip_dead=True), R(cam=cam7, nf=5, seed=2, crop=(380, 360, 40, 24)))
    assert np.array_equal(R(nf=0, seed=2, crop=(380, 360, 40, 24), skip_dead=True), R(nf=0, seed=2, crop=(380, 360, 40, 24)))
    assert np.array_equal(R(seed=2, crop=(190, 180, 20, 12), ssaa=2, skip_dead=True), R(seed=2, crop=(190, 180, 20, 12), ssaa=2))
    # a single ray, and one row
    assert np.array_equal(R(seed=1, crop=(400, 400, 1, 1), skip_dead=True), R(seed=1, crop=(400, 400, 1, 1)))
    assert np.array_equal(R(seed=1, crop=(0, 400, 800, 1), skip_dead=True), R(seed=1, crop=(0, 400, 800, 1)))


def test_skip_dead_full_frame(renderer, native, samples):
    """BASELINE C3 at full size: bit-identical to the plain frame; Gate 1 against the oracle's whole frame when the fixture is
    present; and the executed-work figures the bench line reports."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0)
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, skip_dead=True, return_stats=True)
    assert np.array_equal(img, ref)
    path = os.path.join(os.path.dirname(__file__), "golden", "frame_c3_800_seed0.npz")
    if os.path.exists(path):
        _gate1(img, np.load(path)["image"])
    print(f"\nskip_dead full frame: passes {st.n_passes}, fine trunk {st.n_exec_fine_trunk / st.n_fine_points:.4f}, "
          f"colour {st.n_exec_colour / st.n_fine_points:.4f}, coarse trunk {st.n_exec_coarse_trunk / st.n_coarse_points:.4f}, "
          f"ms total {st.ms_total:.1f} (coarse {st.ms_coarse_mlp:.1f} fine {st.ms_fine_mlp:.1f} other {st.ms_other:.1f})")
    assert st.n_exec_fine_trunk < 0.99 * st.n_fine_points and st.n_exec_colour < 0.3 * st.n_fine_points
    assert st.n_passes == 1 and st.n_mlp_launches == 2     # round 3: live samples are compacted in LDS, colour passes run inside the trunk launch


def test_skip_dead_small_export_budget_means_more_passes(native, samples, monkeypatch):
    """Split arithmetics still export the live samples' trunk outputs through HBM (two launches): that buffer is sized for the worst
    case of a pass, and a small budget only means more passes.  The f32 kernel has no such buffer: one pass whatever the budget."""
    monkeypatch.setenv("NERF_MAX_EXPORT_BYTES", str(64 * 192 * 1024 * 5))      # five 64-ray rows of 192 samples
    with native.Renderer(0) as r:
        r.load_scene(SCENE)
        cam = native.camera_from_samples(samples, 800, 800, 64)
        crop = (368, 352, 64, 23)
        img, st = native.render_image(r.coarse, r.fine, cam, 128, seed=0, crop=crop, skip_dead=True, dtype="f16x2", return_stats=True)
        assert st.n_passes == 5
        f32, st32 = native.render_image(r.coarse, r.fine, cam, 128, seed=0, crop=crop, skip_dead=True, return_stats=True)
        assert st32.n_passes == 1
        monkeypatch.delenv("NERF_MAX_EXPORT_BYTES")
        assert np.array_equal(img, native.render_image(r.coarse, r.fine, cam, 128, seed=0, crop=crop, dtype="f16x2"))
        assert np.array_equal(f32, native.render_image(r.coarse, r.fine, cam, 128, seed=0, crop=crop))


def test_skip_dead_dense_and_sparse_staging(renderer, native, samples):
    """The in-LDS compaction of the f32 kernel: windows where nearly every chunk of every wave carries live samples (several colour
    passes per trunk step, mid-step flushes) and windows where live samples trickle in (partial final flush), odd widths so that
    the last workgroup has idle waves -- all bit-identical to the fused kernel."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    R = lambda **kw: native.render_image(renderer.coarse, renderer.fine, cam, 128, **kw)  # noqa: E731
    for crop in ((390, 380, 3, 1), (396, 300, 5, 7), (330, 420, 131, 3), (0, 0, 800, 2), (200, 398, 401, 2)):
        for seed in (0, 11):
            a, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=seed, crop=crop, skip_dead=True, return_stats=True)
            assert np.array_equal(a, R(seed=seed, crop=crop)), (crop, seed)
            assert st.n_exec_colour <= st.n_exec_fine_trunk
    # coarse_only + a 3-sample network pass (one partial chunk per ray, every wave nearly empty)
    cam3 = native.camera_from_samples(samples, 800, 800, 3)
    a = native.render_image(renderer.coarse, renderer.fine, cam3, 0, seed=1, crop=(380, 380, 33, 5), coarse_only=True, skip_dead=True)
    assert np.array_equal(a, native.render_image(renderer.coarse, renderer.fine, cam3, 0, seed=1, crop=(380, 380, 33, 5), coarse_only=True))


def test_skip_dead_fuzz_is_bit_exact():
    """tools/fuzz_skip_dead.py for 12 s (~2 500 random windows / sample counts / seeds / SSAA / coarse_only / arithmetics): every case
    must reproduce the non-skipping frame bit for bit.  (Round 3: 56 873 cases, 101 M rays in 240 s, 0 mismatching.)"""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_skip_dead.py"), "12", "7"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    last = p.stdout.strip().splitlines()[-1]
    assert "0 mismatching" in last and int(last.split()[1]) > 200, last


def test_skip_dead_argument_errors(renderer, native, samples):
    """bf16 is its own arithmetic: there are no f32 sample positions for hybrid_sampling to protect."""
    cam = native.camera_from_samples(samples, 64, 64, 64)
    with pytest.raises(native.NerfError) as e:
        native.render_image(renderer.coarse, renderer.fine, cam, 128, skip_dead=True, hybrid_sampling=True, dtype="bf16")
    assert e.value.code == -1 and "mlp_dtype F32, BF16X3 or F16X2" in e.value.msg


def test_skip_dead_in_bf16_arithmetic(renderer, native, samples, monkeypatch):
    """skip_dead with mlp_dtype = bf16 (BASELINE config C5's arithmetic): every pass ray-sequential in bf16 (two ray cursors per wave),
    the colour head on the compacted live samples from bf16-packed exported tiles (512 B per sample).  Per MFMA column the arithmetic
    is the fused bf16 kernel's, and the fused kernel is what tests/test_gpu_bf16*.py hold against the oracle's bf16 emulation: so the
    gate here is bit-identity with the non-skipping bf16 frame -- whole C3 frame, C5's 2x2 SSAA, ragged sample counts, coarse_only,
    a one-pixel crop, and a small export budget (more passes)."""
    r = renderer
    cam = native.camera_from_samples(samples, 800, 800, 64)
    for nc, nf, crop, co, ssaa in ((64, 128, (300, 300, 200, 64), False, 1), (40, 50, (380, 360, 40, 24), False, 1), (4, 0, (150, 150, 100, 40), True, 1),
                                   (64, 0, (200, 200, 300, 100), True, 1), (33, 31, (0, 0, 800, 8), False, 1), (64, 128, (395, 400, 1, 1), False, 1),
                                   (64, 128, (350, 380, 90, 30), False, 2)):
        c = native.camera_from_samples(samples, 800, 800, nc)
        kw = dict(seed=1, crop=crop, coarse_only=co, ssaa=ssaa, dtype="bf16")
        a = native.render_image(r.coarse, r.fine, c, nf, **kw)
        b, st = native.render_image(r.coarse, r.fine, c, nf, skip_dead=True, return_stats=True, **kw)
        assert np.array_equal(a, b), (nc, nf, crop, co, ssaa)
        assert st.n_mlp_launches == (2 if co else 3) * st.n_passes
    full_ref = native.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16")
    plain = native.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", return_stats=True)[1]
    full, st = native.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", skip_dead=True, return_stats=True)
    assert np.array_equal(full, full_ref)
    assert st.n_exec_coarse_trunk < 0.85 * st.n_coarse_points and st.n_exec_fine_trunk < 0.98 * st.n_fine_points
    assert 0 < st.n_exec_colour < 0.2 * st.n_fine_points
    print(f"\nskip_dead bf16 full frame: {st.n_rays / (st.ms_total * 1e-3):.0f} rays/s ({st.ms_total:.1f} ms, {st.n_passes} passes) vs plain bf16 "
          f"{plain.n_rays / (plain.ms_total * 1e-3):.0f} rays/s ({plain.ms_total:.1f} ms); coarse trunk {st.n_exec_coarse_trunk / st.n_coarse_points:.4f}, "
          f"fine trunk {st.n_exec_fine_trunk / st.n_fine_points:.4f}, colour {st.n_exec_colour / st.n_fine_points:.4f}")
    monkeypatch.setenv("NERF_MAX_EXPORT_BYTES", str(48 << 20))  # 48 MiB: a 200 x 64 crop of 192 samples (1.2 GiB worst case) needs many passes
    with native.Renderer(0) as r2:
        r2.load_scene(SCENE)
        crop = (300, 300, 200, 64)
        img, st = native.render_image(r2.coarse, r2.fine, cam, 128, seed=0, crop=crop, skip_dead=True, dtype="bf16", return_stats=True)
        assert st.n_passes > 8
        assert np.array_equal(img, native.render_image(r2.coarse, r2.fine, cam, 128, seed=0, crop=crop, dtype="bf16"))


def test_skip_dead_in_bf16x3_arithmetic(renderer, native, samples):
    """skip_dead with mlp_dtype = bf16x3: f32 ray-sequential coarse pass + bf16x3 ray-sequential fine trunk + bf16x3 colour head on
    the live samples.  Bit-identical to the non-skipping bf16x3 frame (whole C3 frame and ragged variants), Gate 1 against the
    oracle's crop."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    g = golden("crop_c3_800_64_128.npz")
    crop = tuple(int(v) for v in g["crop"])
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop, dtype="bf16x3")
    img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop, dtype="bf16x3", skip_dead=True, return_stats=True)
    assert np.array_equal(img, ref)
    _gate1(img, g["image"])
    assert st.n_exec_fine_trunk < 0.98 * st.n_fine_points and 0 < st.n_exec_colour < 0.7 * st.n_fine_points
    cam40 = native.camera_from_samples(samples, 800, 800, 40)
    a = native.render_image(renderer.coarse, renderer.fine, cam40, 50, seed=2, crop=(380, 360, 40, 24), dtype="bf16x3", skip_dead=True)
    assert np.array_equal(a, native.render_image(renderer.coarse, renderer.fine, cam40, 50, seed=2, crop=(380, 360, 40, 24), dtype="bf16x3"))
    cam4 = native.camera_from_samples(samples, 400, 400, 64)
    c = native.render_image(renderer.coarse, renderer.fine, cam4, 0, seed=0, coarse_only=True, crop=(150, 150, 100, 40), dtype="bf16x3", skip_dead=True)
    assert np.array_equal(c, native.render_image(renderer.coarse, renderer.fine, cam4, 0, seed=0, coarse_only=True, crop=(150, 150, 100, 40), dtype="bf16x3"))
    full_ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype="bf16x3")
    full, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype="bf16x3", skip_dead=True, return_stats=True)
    assert np.array_equal(full, full_ref)
    print(f"\nskip_dead bf16x3 full frame: {st.n_rays / (st.ms_total * 1e-3):.0f} rays/s, ms total {st.ms_total:.1f} "
          f"(coarse {st.ms_coarse_mlp:.1f} fine {st.ms_fine_mlp:.1f} other {st.ms_other:.1f}); fine trunk {st.n_exec_fine_trunk / st.n_fine_points:.4f}, "
          f"colour {st.n_exec_colour / st.n_fine_points:.4f}")
