#!/usr/bin/env python3
"""A/B of certify_zero's bf16 pre-filter on the C3 frame (tuning variant: NERF_CERTIFY_SEQ_PREFILTER is read in variant builds only):
0 = the fused bf16 kernel over all samples, 1 = the ray-sequential bf16 trunk that stops at its own predicted cut; and of
NERF_CERTIFY_ZERO_TILES (probable zeros + audited certificates in the list's back part, evaluated with skip_empty)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
cam = N.camera_from_samples(S, 800, 800, 64)
ref = {}
for rep in range(2):
    for seq, zt in ((0, 0), (1, 0), (1, 1)):
        os.environ["NERF_CERTIFY_SEQ_PREFILTER"] = str(seq); os.environ["NERF_CERTIFY_ZERO_TILES"] = str(zt)
        with N.Renderer(0) as r:
            r.load_scene(os.path.join(ROOT, "lego_rust"))
            for dt in ("f32", "f16x2"):
                if dt not in ref:
                    ref[dt] = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt)
                best = None
                for k in range(4):
                    img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt, certify_zero=True, return_stats=True)
                    if best is None or st.ms_total < best.ms_total: best = st
                print(f"seq_prefilter={seq} zero_tiles={zt} {dt}: identical={np.array_equal(img, ref[dt])} {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f}); lists "
                      f"{best.n_exec_coarse_trunk / best.n_coarse_points:.4f} / {best.n_exec_fine_trunk / best.n_fine_points:.4f}; fallback rays {best.n_certify_fallback_rays}; "
                      f"max_err {best.certify_max_error} retries {best.n_certify_retries}", flush=True)
