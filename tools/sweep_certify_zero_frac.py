#!/usr/bin/env python3
"""Which listed samples count as "probably zero" (their tiles may skip the colour head): threshold = margin x frac.  Tuning variant only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
cam = N.camera_from_samples(S, 800, 800, 64)
ref = None
for frac in ("0.5", "0.3333", "0.25", "0.15", "0.1", "0.05", "0.02"):
    os.environ["NERF_CERTIFY_ZERO_FRAC"] = frac
    with N.Renderer(0) as r:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
        if ref is None:
            ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
        best = None
        for k in range(3):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        print(f"zero_frac {frac}: identical={np.array_equal(img, ref)} {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f}); fine list "
              f"{best.n_exec_fine_trunk / best.n_fine_points:.4f}, colour heads {best.n_exec_colour / best.n_fine_points:.4f}", flush=True)
