#!/usr/bin/env python3
"""Quick check of skip_dead on the 800x800 frame: bit-identity with the plain frame + device times (best of n) per mode."""
import os, sys
os.environ.setdefault("NERF_ALLOW_VARIANT", "1")  # these tools exist to time variant builds
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32"]
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    for dt in modes:
        ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt)
        best = None
        for k in range(n):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt, skip_dead=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total:
                best = st
        print(f"{dt}: skip_dead identical={np.array_equal(img, ref)} passes {best.n_passes} launches {best.n_mlp_launches} total {best.ms_total:.2f} ms coarse {best.ms_coarse_mlp:.2f} "
              f"fine {best.ms_fine_mlp:.2f} other {best.ms_other:.2f}; exec coarse {best.n_exec_coarse_trunk / best.n_coarse_points:.4f} fine {best.n_exec_fine_trunk / best.n_fine_points:.4f} "
              f"colour {best.n_exec_colour / best.n_fine_points:.4f}" +
              (f" DIAG passes {best.n_hybrid_rays} raw_coarse_exec {best.n_exec_coarse_trunk}" if os.environ.get("NERF_MI355X_LIB") else ""), flush=True)
