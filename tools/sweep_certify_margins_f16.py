#!/usr/bin/env python3
"""What the margin floors of the f16 pre-filter cost in work (tuning variant only: NERF_CERTIFY_MARGINS_F16 is not read by the product build).
    NERF_ALLOW_VARIANT=1 NERF_MI355X_LIB=nerf-rs_amd/libnerf_mi355x_certtune.so python tools/sweep_certify_margins_f16.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
cam = N.camera_from_samples(S, 800, 800, 64)
ref = None
for m in ("1.0,3.0", "0.5,1.0", "0.25,0.5", "0.15,0.4", "0.1,0.3", "0.07,0.25", "0.25,0.3", "0.1,0.5"):
    os.environ["NERF_CERTIFY_MARGINS_F16"] = m
    with N.Renderer(0) as r:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
        os.environ.pop("NERF_CERTIFY_MARGINS_F16")   # (read at nerf_create; load_scene resets to the floors set there)
        if ref is None:
            ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2")
        best = None
        for k in range(3):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        print(f"f16 margins {m}: identical={np.array_equal(img, ref)} {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f}); lists "
              f"{best.n_exec_coarse_trunk / best.n_coarse_points:.4f} / {best.n_exec_fine_trunk / best.n_fine_points:.4f}; margins in force {best.certify_margin} "
              f"max_err {best.certify_max_error} headroom {best.certify_headroom} retries {best.n_certify_retries}", flush=True)
os.environ["NERF_CERTIFY_PREFILTER_F16"] = "0"
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    best = None
    for k in range(3):
        img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", certify_zero=True, return_stats=True)
        if best is None or st.ms_total < best.ms_total: best = st
    print(f"bf16 pre-filter: identical={np.array_equal(img, ref)} {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f}); lists "
          f"{best.n_exec_coarse_trunk / best.n_coarse_points:.4f} / {best.n_exec_fine_trunk / best.n_fine_points:.4f}; margins in force {best.certify_margin}", flush=True)
