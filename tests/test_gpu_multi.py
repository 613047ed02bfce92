"""Multi-GPU behind the C ABI (nerf_render_image_multi): row bands on per-context host threads + streams, gathered into one
framebuffer with no torch involved.  The test box has ONE MI355X, so the contexts share device 0: that exercises the band
split, the threads, the three gather paths' bookkeeping and the ragged cases; the result must be BIT-IDENTICAL to the
single-context frame (per-pixel counter RNG).  RCCL refuses two ranks on one GPU: ncclAllGather itself runs here at n = 1; for
n = 2, 3 on the shared device the RCCL path's equal-slot layout, stream ordering and ragged compaction run with the collective
step rehearsed as device-to-device copies (RCCL proper is used whenever the devices are distinct)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENE, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def three(native):
    rs = [native.Renderer(0) for _ in range(3)]
    for r in rs:
        r.load_scene(SCENE)
    yield rs
    for r in rs:
        r.close()


@pytest.mark.parametrize("gather", ["host", "peer", "rccl"])
@pytest.mark.parametrize("n", [1, 2, 3])
def test_multi_bands_are_bit_identical_to_one_context(native, renderer, samples, three, n, gather):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (200, 300, 400, 101)                       # 101 rows: 51+50 over 2 contexts, 34+34+33 over 3 (ragged)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
    img, st = native.render_image_multi(three[:n], cam, 128, gather=gather, seed=0, crop=crop, return_stats=True)
    assert np.array_equal(img, ref)
    assert [s.n_rays for s in st] == [400 * native.band_of_rank(101, i, n)[1] for i in range(n)]
    assert (np.abs(ref - 1.0) > 1e-3).mean() > 0.2    # the window really shows the model


def test_multi_more_contexts_than_rows_and_golden_crop(native, samples, three):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    g = golden("crop_c3_800_64_128.npz")
    x0, y0, w, h = (int(v) for v in g["crop"])
    img = native.render_image_multi(three, cam, 128, gather="host", seed=0, crop=(x0, y0, w, 2))   # band 2 is empty
    d = np.abs(img - g["image"][:2])
    assert d.max() <= 5e-4 and d.mean() <= 1e-5       # Gate 1 against the oracle fixture
    img = native.render_image_multi(three, cam, 128, gather="peer", seed=0, crop=(x0, y0, w, h), dtype="bf16x3", skip_empty=True)
    d = np.abs(img - g["image"])
    assert np.quantile(d, 0.999) <= 2e-5 and d.max() <= 2e-3


def test_multi_with_skip_dead_and_hybrid_sampling(native, renderer, samples, three):
    """The opt-in fast modes are per-ray decisions (ray queue, flagged-ray list), so banding must not change a bit of them."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (300, 330, 200, 47)
    kw = dict(seed=0, crop=crop, dtype="f16x2", skip_dead=True, hybrid_sampling=True)
    one = native.render_image(renderer.coarse, renderer.fine, cam, 128, **kw)
    for n, gather in ((2, "host"), (3, "peer")):
        assert np.array_equal(native.render_image_multi(three[:n], cam, 128, gather=gather, **kw), one)
    # the f32 skip_dead kernel (in-LDS compaction, in-kernel colour passes) through bands: the same bits as the plain single-context frame
    plain = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
    for n, gather in ((2, "rccl"), (3, "host")):
        assert np.array_equal(native.render_image_multi(three[:n], cam, 128, gather=gather, seed=0, crop=crop, skip_dead=True), plain)


@pytest.mark.parametrize("stripe", [0, 1, 3])
def test_band_option_renders_the_documented_rows(native, renderer, samples, stripe):
    """nerf_render_opts.band_*: band i of n holds exactly the rows band_row_indices lists, packed, with the bits of the whole-window
    render -- contiguous bands and stripes, with SSAA (stripes of whole pixel rows), with the modes whose cost follows the scene."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (250, 310, 120, 23)
    for kw in (dict(), dict(ssaa=2, dtype="bf16"), dict(certify_zero=True), dict(skip_dead=True, dtype="f16x2")):
        ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=9, crop=crop, **kw)
        for n in (2, 5):
            for i in range(n):
                img, st = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=9, crop=crop, band=(i, n, stripe), return_stats=True, **kw)
                rows = native.band_row_indices(23, i, n, stripe)
                assert img.shape == (len(rows), 120, 3) and np.array_equal(img, ref[rows]), (kw, n, i)
                assert st.n_rays == len(rows) * 120 * kw.get("ssaa", 1) ** 2
    with pytest.raises(native.NerfError, match="no rows"):
        native.render_image(renderer.coarse, renderer.fine, cam, 128, crop=(0, 0, 8, 2), band=(2, 3, 1))
    with pytest.raises(native.NerfError, match="out of range"):
        native.render_image(renderer.coarse, renderer.fine, cam, 128, crop=(0, 0, 8, 8), band=(3, 3, 1))


@pytest.mark.parametrize("gather", ["host", "peer", "rccl"])
def test_striped_partition_for_the_modes_whose_cost_follows_the_scene(native, renderer, samples, three, gather):
    """skip_dead / skip_empty / certify_zero deal single rows out round-robin (the lego background is nearly free there and sits in the
    top rows): the bands arrive packed and are put in place -- one strided D2H (host), slots + a copy kernel (peer, rccl).  Ragged
    (101 rows over 2 and 3), more contexts than rows, and the plain frame's bits throughout."""
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (200, 300, 400, 101)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
    for n in (2, 3):
        for kw in (dict(certify_zero=True), dict(skip_dead=True), dict(skip_empty=True)):
            img, st = native.render_image_multi(three[:n], cam, 128, gather=gather, seed=0, crop=crop, return_stats=True, **kw)
            assert np.array_equal(img, ref), (n, kw)
            assert [s.n_rays for s in st] == [400 * native.band_rows(101, i, n, 1) for i in range(n)]
    two = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=(200, 300, 400, 2))
    assert np.array_equal(native.render_image_multi(three, cam, 128, gather=gather, seed=0, crop=(200, 300, 400, 2), certify_zero=True), two)


def test_multi_whole_frame_three_contexts(native, renderer, samples, three):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0)
    img, st = native.render_image_multi(three, cam, 128, gather="peer", seed=0, return_stats=True)
    assert np.array_equal(img, ref)
    assert sum(s.n_rays for s in st) == 640000 and [s.n_rays // 800 for s in st] == [267, 267, 266]
    ss = native.render_image_multi(three[:2], cam, 128, gather="host", seed=0, ssaa=2, crop=(380, 360, 24, 9), dtype="bf16")
    one = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, ssaa=2, crop=(380, 360, 24, 9), dtype="bf16")
    assert np.array_equal(ss, one)                    # bands of SSAA pixels: whole output rows per context


def test_multi_rccl_gather(native, renderer, samples, three):
    cam = native.camera_from_samples(samples, 800, 800, 64)
    crop = (368, 352, 64, 7)
    ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
    img = native.render_image_multi(three[:1], cam, 128, gather="rccl", seed=0, crop=crop)       # one rank: ncclCommInitAll + all-gather
    assert np.array_equal(img, ref)
    # even bands (no compaction): 8 rows over 2 contexts, the slot buffer IS the frame; ragged: 7 rows over 2 and 3 contexts
    for n, c in ((2, (368, 352, 64, 8)), (2, crop), (3, crop), (3, (368, 352, 64, 2))):
        one = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=c)
        assert np.array_equal(native.render_image_multi(three[:n], cam, 128, gather="rccl", seed=0, crop=c), one)
    native.load_library().nerf_multi_release()


def test_multi_distinct_devices_when_the_box_has_them(native, renderer, samples):
    """On a node with >= 2 GPUs (the driver's scaling node; this suite's usual box has one and skips): one context per device, ragged
    and even bands, every gather -- RCCL proper (ncclCommInitAll + grouped in-place ncclAllGather over distinct devices), xGMI peer
    copies, host -- must reproduce the single-context frame bit for bit."""
    import ctypes as C
    n_dev = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipGetDeviceCount(C.byref(n_dev)) == 0
    if n_dev.value < 2:
        pytest.skip(f"needs >= 2 GPUs, this box has {n_dev.value}")
    n = min(n_dev.value, 8)
    rs = [native.Renderer(i) for i in range(n)]
    try:
        for r in rs:
            r.load_scene(SCENE)
        cam = native.camera_from_samples(samples, 800, 800, 64)
        for crop in ((200, 300, 400, 8 * n), (200, 300, 400, 8 * n + 5)):           # even and ragged bands
            ref = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, crop=crop)
            for gather in ("rccl", "peer", "host"):
                assert np.array_equal(native.render_image_multi(rs, cam, 128, gather=gather, seed=0, crop=crop), ref), (crop, gather)
            assert np.array_equal(native.render_image_multi(rs, cam, 128, gather="rccl", seed=0, crop=crop, skip_dead=True), ref)
    finally:
        native.load_library().nerf_multi_release()
        for r in rs:
            r.close()


def test_multi_argument_errors(native, samples, three):
    cam = native.camera_from_samples(samples, 64, 64, 64)
    with pytest.raises(native.NerfError) as e:
        native.render_image_multi([three[0], three[0]], cam, 128)
    assert e.value.code == -1 and "listed twice" in e.value.msg
    with pytest.raises(native.NerfError):
        native.render_image_multi([], cam, 128)
    with pytest.raises(native.NerfError) as e:
        native.render_image_multi(three[:2], cam, 128, crop=(0, 0, 65, 3))
    assert "crop window outside the frame" in e.value.msg
    with native.Renderer(0) as empty:                 # a band's failure (network not loaded) surfaces on the call
        with pytest.raises(native.NerfError) as e:
            native.render_image_multi([three[0], empty], cam, 128)
        assert e.value.code == -6 and "not loaded" in e.value.msg


def test_create_multi(native):
    import ctypes as C
    L = native.load_library()
    hs = (C.c_void_p * 2)()
    assert L.nerf_create_multi((C.c_int * 2)(0, 0), 2, hs) == 0 and hs[0] and hs[1] and hs[0] != hs[1]
    L.nerf_destroy(hs[0]); L.nerf_destroy(hs[1])
    assert L.nerf_create_multi((C.c_int * 2)(0, 99), 2, hs) == -1 and not hs[0] and not hs[1]    # all-or-nothing


def test_bench_self_launches_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` with NO outer launcher (the driver's scaling command): the parent starts the ranks, both
    render their band on the one GPU of this box, the gather is rehearsed over gloo (RCCL refuses two ranks per device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "96",
                        "--height", "80", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"].startswith("gloo") and "REHEARSAL" in d["backend"]
    assert d["value"] > 0 and d["scaling"] == "strong" and d["config"]["rays_per_step"] == 96 * 80
    pr = d["per_rank"]                                 # attribution of the step: render vs gather, per rank (max / min over ranks)
    assert len(pr["ms_render"]["by_rank"]) == 2 and pr["ms_render"]["min"] > 0 and pr["ms_gather"]["max"] > 0
    assert pr["ms_render"]["max"] + pr["ms_gather"]["max"] <= 1.5 * d["ms_per_step"] + 50   # the marks bracket the step's own work


@pytest.mark.parametrize("gather,extra", [("rccl", []), ("peer", ["--certify-zero"]), ("host", ["--skip-dead"])])
def test_bench_inproc_times_the_c_abi_multi_entry_point(gather, extra):
    """`bench.py --launch inproc --gather G --gpus 2`: the timed call is nerf_render_image_multi (what a Rust host calls), two contexts
    on this box's one GPU -- labelled a rehearsal -- and the frame must be the one-context frame bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch", "inproc", "--gather", gather, "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--width", "96", "--height", "81"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["launch"].startswith("inproc") and d["gather"] == gather and "REHEARSAL" in d["backend"]
    assert d["image_bit_identical_to_one_context"] is True and d["value"] > 0
    assert d["config"]["partition"] == ("contiguous row bands" if not extra else "rows round-robin (band_stripe_rows = 1)")
    pc = d["per_ctx"]
    assert pc["rays_by_ctx"] == [96 * 41, 96 * 40] and len(pc["ms_render"]["by_ctx"]) == 2 and pc["ms_render"]["min"] > 0
    assert ("roofline" in d) == (not extra)               # executed-flop accounting of the skip modes is the per-rank bench's business


def test_bench_rccl_paths_at_world_size_one():
    """The RCCL (torch "nccl") collective of bench.py's N > 1 path, driven at world size 1 on this box's one GPU: (a) the direct
    init_process_group("nccl") the 8-GPU run takes, (b) the branch for launchers that give every rank ONE visible GPU -- identities
    exchanged over gloo, then an RCCL sub-group for the data path."""
    for extra_env in ({}, {"NERF_BENCH_BACKEND": ""}):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        with socket.socket() as sk:                       # a free port for the one-rank rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env.update(NERF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--width", "96",
                            "--height", "80", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert d["ranks"] == 1 and d["backend"].startswith("nccl") and d["value"] > 0
        assert d["per_rank"]["ms_gather"]["max"] > 0 and d["per_rank"]["ms_dominant_kernel_per_launch"]["max"] > 0   # HIP events around RCCL's all-gather


def test_multi_fuzz_short(native, three):
    """tools/fuzz_multi.py for a few seconds: random context counts, frame sizes (down to one pixel), windows, SSAA, sample counts,
    arithmetics, skip modes, seeds and gathers -- every multi-context frame bit-identical to the single-context one (a 2-minute run
    of the tool: 47 869 cases, 0 mismatching)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_multi
    res = fuzz_multi.fuzz(three, 5.0, 20261004)
    print("\n", res)
    assert res["cases"] > 200 and res["mismatching"] == 0 and min(res["by_gather"].values()) > 20, res
