#!/usr/bin/env python3
"""How conservative is hybrid_sampling's flag?  For a window of the C3 frame: per ray, the flag (nerf_stage_hybrid_flags on the split
arithmetic's densities) and the ACTUAL largest displacement of its draws against the f32 densities' draws."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
W = 800; nc, nf = 64, 128
x0, y0, w, h = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (200, 200, 400, 400)))
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), W, W, nc)
    t = r.stage_stratified(cam, x0, y0, w, h, nc, seed=0).reshape(-1, nc)
    dirs = r.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
    o = cam.pos.astype(np.float32)
    pts = (o[None, None, :] + dirs[:, None, :] * t[:, :, None]).astype(np.float32).reshape(-1, 3)
    dd = np.repeat(dirs, nc, axis=0)
    pix = ((y0 + np.arange(h))[:, None] * W + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
    sg = {}
    for dt in ("f32", "f16x2"):
        out = np.empty(pts.shape[0], np.float32)
        for a in range(0, pts.shape[0], 1 << 22):
            _, s = r.coarse.forward_batch(np.ascontiguousarray(pts[a:a + (1 << 22)].T), dd[a:a + (1 << 22)], dtype=dt)
            out[a:a + (1 << 22)] = s
        sg[dt] = out.reshape(-1, nc)
    ref = r.stage_resample(t, sg["f32"], nf, 6.0, seed=0, pixel_index=pix)["t_new"]
    for tau in (1e-5,):
        flags, tn = r.stage_hybrid_flags(t, sg["f16x2"], nf, 6.0, seed=0, pixel_index=pix, tau=tau)
        move = np.abs(tn - ref).max(axis=1)
        print(f"window {w}x{h}: rays {move.size}; flagged {flags.mean():.4f}; actually moving > 1e-5: {(move > 1e-5).mean():.4f}, > 3e-6: {(move > 3e-6).mean():.4f}, "
              f"> 1e-6: {(move > 1e-6).mean():.4f}; unflagged max {move[~flags].max():.2e}; flagged & moving > 1e-5: {((move > 1e-5) & flags).mean():.4f}; "
              f"empty rays (all sigma 0): {(sg['f32'].max(axis=1) == 0).mean():.4f}; flagged among empty: {flags[sg['f32'].max(axis=1) == 0].mean():.4f}")
