#!/usr/bin/env python3
"""Zero certification (nerf_render_opts.certify_zero): the certified frame must be the plain f32 frame bit for bit; device times
(best of n) and the fraction of the samples the f32 kernel still evaluates."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
ok = True
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    for nc, nf, W, crop, co, ssaa in ((64, 128, 800, (300, 300, 200, 64), False, 1), (40, 50, 800, (380, 360, 40, 24), False, 1), (64, 0, 800, (200, 200, 300, 100), True, 1),
                                      (33, 31, 800, (0, 0, 800, 8), False, 1), (64, 128, 800, (395, 400, 1, 1), False, 1), (64, 128, 257, None, False, 2), (64, 128, 800, None, False, 1)):
        cam = N.camera_from_samples(S, W, W, nc)
        best0 = None
        for k in range(n if crop is None else 1):
            a, s0 = N.render_image(r.coarse, r.fine, cam, nf, seed=1, crop=crop, coarse_only=co, ssaa=ssaa, return_stats=True)
            if best0 is None or s0.ms_total < best0.ms_total: best0 = s0
        best = None
        for k in range(n if crop is None else 1):
            b, st = N.render_image(r.coarse, r.fine, cam, nf, seed=1, crop=crop, coarse_only=co, ssaa=ssaa, certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        same = np.array_equal(a, b)
        ok &= same
        print(f"{W}x{W} crop {crop} {nc}+{nf} coarse_only {co} ssaa {ssaa}: identical={same} (max diff {np.abs(a - b).max():.2e}); plain {best0.ms_total:.1f} ms -> certified {best.ms_total:.1f} ms "
              f"(coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f} other {best.ms_other:.1f}); f32 kernel evaluates coarse {best.n_exec_coarse_trunk / max(best.n_coarse_points, 1):.3f} "
              f"fine {best.n_exec_fine_trunk / max(best.n_fine_points, 1):.3f} of the samples; audited {best.n_certify_audited} violations {best.n_certify_violations} "
              f"headroom {best.certify_headroom} margins {best.certify_margin} fallback rays {best.n_certify_fallback_rays} retries {best.n_certify_retries}", flush=True)
    cam = N.camera_from_samples(S, 800, 800, 64)
    for dt in ("f16x2", "bf16x3"):
        a = N.render_image(r.coarse, r.fine, cam, 128, seed=1, dtype=dt)
        best = None
        for k in range(n):
            b, st = N.render_image(r.coarse, r.fine, cam, 128, seed=1, dtype=dt, certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        same = np.array_equal(a, b); ok &= same
        print(f"{dt} C3 frame: identical={same}; certified {best.ms_total:.1f} ms (coarse {best.ms_coarse_mlp:.1f} fine {best.ms_fine_mlp:.1f} other {best.ms_other:.1f}); "
              f"nonfinite {best.n_nonfinite_points}; fine list {best.n_exec_fine_trunk / max(best.n_fine_points, 1):.3f} headroom {best.certify_headroom} fallback rays {best.n_certify_fallback_rays}", flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
