#!/usr/bin/env python3
"""Quick check of hybrid_sampling on the 800x800 frame: device times (best of n) and the redone fraction per arithmetic, the frame
against the f32 frame, whole-frame Gate 1 against the committed oracle frame."""
import os, sys
os.environ.setdefault("NERF_ALLOW_VARIANT", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16x2", "bf16x3", "f32"]
fx = os.path.join(ROOT, "tests", "golden", "frame_c3_800_seed0.npz")
oracle = np.load(fx)["image"] if os.path.exists(fx) and "image" in np.load(fx).files else None
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    f32 = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
    for dt in modes:
        best = None
        for k in range(n):
            img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt, skip_dead=True, hybrid_sampling=True, return_stats=True)
            if best is None or st.ms_total < best.ms_total:
                best = st
        d = np.abs(img - f32)
        line = (f"{dt}: hybrid total {best.ms_total:.2f} ms coarse+redo {best.ms_coarse_mlp:.2f} fine {best.ms_fine_mlp:.2f} other {best.ms_other:.2f}; "
                f"redone {best.n_hybrid_rays / best.n_rays:.4f}; vs f32 frame max {d.max():.2e} mean {d.mean():.2e}")
        if oracle is not None:
            do = np.abs(img - oracle.reshape(img.shape))
            line += f"; vs oracle frame max {do.max():.2e} mean {do.mean():.2e}"
        print(line, flush=True)
