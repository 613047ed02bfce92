#!/usr/bin/env python3
"""Determinism soak: the same 800x800 frame N times per arithmetic and mode; every repetition must be bit-identical to the first --
a race in one of the LDS weight pipelines, in the device-side ray queue / live-sample export of skip_dead, or in the flagged-ray
list of hybrid_sampling, or in the sample list of certify_zero (all filled in arbitrary order) would show up as a sporadic mismatch."""
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
    modes = [(d, dict(skip_empty=sk)) for d in ("f32", "bf16x3", "f16x2", "bf16") for sk in (False, True)]
    modes += [(d, dict(skip_dead=True)) for d in ("f32", "bf16x3", "f16x2", "bf16")]
    modes += [(d, dict(skip_dead=True, hybrid_sampling=True)) for d in ("f32", "bf16x3", "f16x2")]
    modes += [(d, dict(certify_zero=True)) for d in ("f32", "bf16x3", "f16x2")]  # the sample list is filled in arbitrary order
    for dtype, kw in modes:
        t0 = time.time()
        ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dtype, **kw)
        mism = 0
        for k in range(n - 1):
            img = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dtype, **kw)
            mism += int(not np.array_equal(img, ref))
        bad += mism
        print(f"{dtype:7s} {kw}: {n} frames, {mism} mismatching, finite={bool(np.isfinite(ref).all())}, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
