import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import nerf_rs_amd as N
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    ref = None
    for skip in (False, True, False, True):
        img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, skip_empty=skip, return_stats=True)
        if ref is None: ref = img
        print(f"skip_empty={skip}: total {st.ms_total:.1f} ms fine {st.ms_fine_mlp:.1f} -> {st.n_rays/st.ms_total*1e3:.0f} rays/s; skipped {st.n_colour_skipped_points/max(st.n_fine_points,1):.3f} of fine samples; identical {np.array_equal(img, ref)}")
