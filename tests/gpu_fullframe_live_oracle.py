"""Full-frame parity of the headline configuration (BASELINE config C3: lego 800x800, 64 + 128 samples, f32): the whole GPU
frame against the whole CPU-oracle frame, same seed.  The oracle needs ~3 minutes of 16 host threads for the 640 000 rays,
so it is NOT part of the regular suite (the file name keeps pytest from collecting it): the regular -m gpu suite holds the same
whole frame to the unrelaxed Gate 1 against the COMMITTED oracle frame (tests/test_gpu_frame_fixture.py).  Run it by name:

    NERF_FULLFRAME=1 python -m pytest tests/gpu_fullframe_live_oracle.py -q -m gpu -s

It prints one JSON line (round 1: profiles/parity_fullframe_r01.json) and enforces: PSNR(GPU, CPU) >= 90 dB; |d| > 5e-5 on at
most 0.1 % of the channel values and never above 5e-4 (the path is discontinuous in a few places -- a 1e-5 relative
difference in a coarse density can move a CDF entry across a fixed uniform draw, which relocates one fine sample: the
measured frame has a handful of such pixels, max 2.4e-4, against a mean of 4.9e-8); 8-bit output identical in >= 99.99 % of
the channel values and never more than one step apart; Gate 2 of the north star on the full frame
(|PSNR(GPU seed 1, CPU seed 0) - PSNR(CPU seed 1, CPU seed 0)| <= 0.1 dB) in f32 and bf16.
"""
import json
import os
import time

import numpy as np
import pytest

from conftest import psnr, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("NERF_FULLFRAME") != "1", reason="set NERF_FULLFRAME=1 (takes ~6 minutes of host CPU)")
def test_full_frame_c3_matches_oracle(renderer, native, oracle, oracle_nets, samples):
    W = H = int(os.environ.get("NERF_FULLFRAME_SIZE", "800"))
    cam = native.camera_from_samples(samples, W, H, 64)
    ocam = oracle.camera_from_samples(samples, W, H)
    import sys
    sys.path.insert(0, ROOT)
    from bench import host_cores  # cgroup-aware thread count: the GPU box shows 256 cores but grants a share of them
    nthr = host_cores()
    t0 = time.time()
    cpu0 = oracle.render_image(*oracle_nets, ocam, oracle.make_opts(64, 128, seed=0, threads=nthr))
    t_cpu = time.time() - t0
    print(f"\noracle frame: {t_cpu:.0f} s on {nthr} threads", flush=True)
    cpu1 = oracle.render_image(*oracle_nets, ocam, oracle.make_opts(64, 128, seed=1, threads=nthr))
    gpu0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0)
    gpu1 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=1)
    x3_0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype="bf16x3")
    x3_1 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=1, dtype="bf16x3")
    b16_0 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=0, dtype="bf16")
    b16_1 = native.render_image(renderer.coarse, renderer.fine, cam, 128, seed=1, dtype="bf16")
    d = np.abs(gpu0 - cpu0)
    dx = np.abs(x3_0 - cpu0)
    q_gpu, q_cpu, q_x3 = native.quantize_rgb8(gpu0), oracle.quantize_rgb8(cpu0), native.quantize_rgb8(x3_0)
    rec = {
        "config": f"lego {W}x{H}, 64+128 samples/ray, seed 0, full frame", "cpu_seconds_per_frame": round(t_cpu, 1), "cpu_threads": nthr,
        "f32_max_abs_diff": float(d.max()), "f32_mean_abs_diff": float(d.mean()), "f32_psnr_gpu_vs_cpu_db": psnr(gpu0, cpu0),
        "f32_fraction_above_5e-5": float((d > 5e-5).mean()), "f32_p9999_abs_diff": float(np.quantile(d, 0.9999)),
        "rgb8_equal_fraction": float((q_gpu == q_cpu).mean()), "rgb8_max_step": int(np.abs(q_gpu.astype(int) - q_cpu.astype(int)).max()),
        "gate2_cpu_seed1_vs_cpu_seed0_db": psnr(cpu1, cpu0), "gate2_gpu_f32_seed1_vs_cpu_seed0_db": psnr(gpu1, cpu0),
        "gate2_gpu_bf16_seed1_vs_cpu_seed0_db": psnr(b16_1, cpu0), "bf16_vs_cpu_same_seed_db": psnr(b16_0, cpu0),
        "bf16_vs_f32_gpu_same_seed_db": psnr(b16_0, gpu0),
        "bf16x3_max_abs_diff": float(dx.max()), "bf16x3_mean_abs_diff": float(dx.mean()), "bf16x3_psnr_gpu_vs_cpu_db": psnr(x3_0, cpu0),
        "bf16x3_fraction_above_5e-5": float((dx > 5e-5).mean()), "bf16x3_p9999_abs_diff": float(np.quantile(dx, 0.9999)),
        "bf16x3_rgb8_equal_fraction": float((q_x3 == q_cpu).mean()), "bf16x3_rgb8_max_step": int(np.abs(q_x3.astype(int) - q_cpu.astype(int)).max()),
        "gate2_gpu_bf16x3_seed1_vs_cpu_seed0_db": psnr(x3_1, cpu0),
    }
    print("\nFULLFRAME " + json.dumps(rec))
    assert np.isfinite(gpu0).all() and np.isfinite(b16_0).all()
    assert rec["f32_max_abs_diff"] <= 5e-4 and rec["f32_fraction_above_5e-5"] <= 1e-3 and rec["f32_psnr_gpu_vs_cpu_db"] >= 90.0
    assert rec["rgb8_equal_fraction"] >= 0.9999 and rec["rgb8_max_step"] <= 1
    assert abs(rec["gate2_gpu_f32_seed1_vs_cpu_seed0_db"] - rec["gate2_cpu_seed1_vs_cpu_seed0_db"]) <= 0.1
    assert abs(rec["gate2_gpu_bf16_seed1_vs_cpu_seed0_db"] - rec["gate2_cpu_seed1_vs_cpu_seed0_db"]) <= 0.1
    assert rec["bf16_vs_f32_gpu_same_seed_db"] >= 45.0
    # the opt-in bf16x3 arithmetic is held to the f32 path's bounds (unrelaxed since the coarse pass runs on the f32 kernel)
    assert rec["bf16x3_max_abs_diff"] <= 5e-4 and rec["bf16x3_fraction_above_5e-5"] <= 1e-3 and rec["bf16x3_psnr_gpu_vs_cpu_db"] >= 90.0
    assert rec["bf16x3_rgb8_equal_fraction"] >= 0.9999 and rec["bf16x3_rgb8_max_step"] <= 1
    assert abs(rec["gate2_gpu_bf16x3_seed1_vs_cpu_seed0_db"] - rec["gate2_cpu_seed1_vs_cpu_seed0_db"]) <= 0.1
