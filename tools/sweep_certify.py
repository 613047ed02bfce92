#!/usr/bin/env python3
"""certify_zero's two work knobs on the C3 frame (tuning variant only: make -C nerf-rs_amd/csrc variant NAME=ctune DEFS=-DNERF_CERT_TUNING=1;
the product build reads none of these variables): the bf16 optical depth at which a ray's cut is PREDICTED (the exact cut is at
-ln 1e-4 = 9.21; a prediction that comes too early costs a second launch for that ray, one that comes late costs samples) and the audit rate.
Every frame is compared with the plain frame.  Usage: NERF_MI355X_LIB=nerf-rs_amd/libnerf_mi355x_ctune.so NERF_ALLOW_VARIANT=1 sweep_certify.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
cam = N.camera_from_samples(S, 800, 800, 64)
ref = None
for depth, mask in [(4.0, 63), (7.0, 63), (8.5, 63), (11.51, 63), (10.5, 63), (10.0, 63), (9.7, 63), (9.5, 63), (9.35, 63), (9.25, 63), (9.5, 127), (9.5, 31)]:
    os.environ["NERF_CERTIFY_CUT_DEPTH"] = str(depth); os.environ["NERF_CERTIFY_AUDIT_MASK"] = str(mask)
    with N.Renderer(0) as r:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
        if ref is None:
            ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
        best = None
        for k in range(3):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        print(f"depth {depth} audit 1/{mask + 1}: identical={np.array_equal(img, ref)} {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f}); lists "
              f"{best.n_exec_coarse_trunk / best.n_coarse_points:.4f} / {best.n_exec_fine_trunk / best.n_fine_points:.4f}; fallback rays {best.n_certify_fallback_rays}; audited "
              f"{best.n_certify_audited} headroom {best.certify_headroom} max_err {best.certify_max_error} retries {best.n_certify_retries}", flush=True)
