#!/usr/bin/env python3
"""Determinism soak: the same 800x800 frame N times per arithmetic (with and without skip_empty); every repetition must be
bit-identical to the first -- a race in one of the LDS weight pipelines would show up as a sporadic mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    for dtype in ("f32", "bf16x3", "bf16"):
        for skip in (False, True):
            t0 = time.time()
            ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dtype, skip_empty=skip)
            mism = 0
            for k in range(n - 1):
                img = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dtype, skip_empty=skip)
                mism += int(not np.array_equal(img, ref))
            bad += mism
            print(f"{dtype:7s} skip_empty={int(skip)}: {n} frames, {mism} mismatching, finite={bool(np.isfinite(ref).all())}, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
