#!/usr/bin/env python3
"""Probe: hybrid_sampling threshold (NERF_HYBRID_TAU, read at nerf_create) vs redone rays, frame time and pixel difference to the
f32-sampling frame, whole C3 frame, f16x2 + skip_dead."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
scene = os.path.join(ROOT, "lego_rust")
A = np.load(os.path.join(ROOT, "tests/golden/frame_c3_800_seed0.npz"))["image"]
base = None
for tau in ("3e-6", "1e-5", "3e-5", "1e-4", "1e-3", "1"):
    os.environ["NERF_HYBRID_TAU"] = tau
    with N.Renderer(0) as r:
        r.load_scene(scene)
        cam = N.camera_from_samples(os.path.join(scene, "tf_reference_samples.json"), 800, 800, 64)
        if base is None:
            base = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", skip_dead=True)
        N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", skip_dead=True, hybrid_sampling=True)
        img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", skip_dead=True, hybrid_sampling=True, return_stats=True)
        d = np.abs(img - base); o = np.abs(img - A)
        print(f"tau {tau:>5}: redone {st.n_hybrid_rays / st.n_rays:.4f}  frame {st.ms_total:6.1f} ms (coarse {st.ms_coarse_mlp:5.1f})  "
              f"vs f32-sampling max {d.max():.2e} mean {d.mean():.2e} >5e-5: {(d > 5e-5).mean():.2e}   vs oracle max {o.max():.2e} mean {o.mean():.2e}", flush=True)
