#!/usr/bin/env python3
"""The bf16 mode (NERF_MLP_BF16, BASELINE config C5) against its f16 twin (experiment: a variant library built with -DNERF_V2_F16_FULL=1 runs
NERF_MLP_BF16 renders on mlp_kernel_f16v2.hip's full kernel): the C3 frame in the 16-bit arithmetic against the f32 frame of the same seed -- error
statistics, PSNR -- and the frame time.  Run once per library:  [NERF_ALLOW_VARIANT=1 NERF_MI355X_LIB=...] python tools/compare_16bit_modes.py"""
import os, sys
os.environ.setdefault("NERF_ALLOW_VARIANT", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
    best = None
    for _ in range(4):
        img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", return_stats=True)
        best = st if best is None or st.ms_total < best.ms_total else best
    d = np.abs(img - ref)
    mse = float(np.mean((img - ref) ** 2))
    print(f"{os.path.basename(N.lib_path())} [{N.build_variant() if hasattr(N, 'build_variant') else ''}]: 16-bit frame vs the f32 frame (same seed): max {d.max():.3e} mean {d.mean():.3e} "
          f"PSNR {10 * np.log10(1.0 / mse):.2f} dB; pixels beyond 5e-4: {(d.max(axis=2) > 5e-4).mean():.4f}; frame {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f}, fine {best.ms_fine_mlp:.1f})", flush=True)
