import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import nerf_rs_amd as N
ROOT='/root/repo'
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
    for dt in ("f32", "f16x2"):
        refd = ref if dt == "f32" else N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt)
        best=None
        for k in range(3):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt, certify_zero=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        print(dt, "identical", bool(np.array_equal(img, refd)), "ms %.1f coarse %.1f fine %.1f" % (best.ms_total, best.ms_coarse_mlp, best.ms_fine_mlp),
              "exec coarse %.4f fine %.4f" % (best.n_exec_coarse_trunk / best.n_coarse_points, best.n_exec_fine_trunk / best.n_fine_points),
              "margins", best.certify_margin, "headroom", best.certify_headroom, "max_err", best.certify_max_error, "retries", best.n_certify_retries,
              "violations", best.n_certify_violations, "audited", best.n_certify_audited, "fallback", best.n_certify_fallback_rays, flush=True)
