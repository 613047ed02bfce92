#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): explain the largest per-pixel difference between the f32 and the bf16x3 frame.
Finds the pixel, then replays its ray stage by stage with both arithmetics: coarse densities -> resampling -> which fine
samples differ -> final colour."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    a = N.render_image(r.coarse, r.fine, cam, 128, seed=0)
    b = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16x3")
    d = np.abs(a - b).max(axis=2)
    i, j = np.unravel_index(np.argmax(d), d.shape)
    print(f"largest difference {d[i, j]:.3e} at pixel (row {i}, col {j}); f32 {a[i, j]}, bf16x3 {b[i, j]}; values > 5e-5: {(d > 5e-5).sum()} pixels")
    dirs = r.stage_ray_dirs(cam, j, i, 1, 1)[0, 0]
    tc = r.stage_stratified(cam, j, i, 1, 1, 64, seed=0)[0]            # (1, 64)
    org = np.array(cam.c.pos, np.float32)
    pts = (org[:, None] + dirs[:, None] * tc[0][None, :]).astype(np.float32)
    vd = np.tile(dirs, (64, 1))
    sg = {dt: r.coarse.forward_batch(pts, vd, dtype=dt)[1] for dt in ("f32", "bf16x3")}
    print(f"coarse sigma: max rel diff {np.max(np.abs(sg['f32'] - sg['bf16x3']) / (1 + np.abs(sg['f32']))):.2e}")
    pix = np.array([i * 800 + j], np.uint32)
    rs = {dt: r.stage_resample(tc, sg[dt][None, :], 128, cam.c.far, seed=0, pixel_index=pix) for dt in sg}
    dt_new = np.abs(rs["f32"]["t_new"] - rs["bf16x3"]["t_new"])[0]
    moved = np.nonzero(dt_new > 1e-4)[0]
    print(f"cdf max diff {np.abs(rs['f32']['cdf'] - rs['bf16x3']['cdf']).max():.2e}; fine draws that moved by > 1e-4: {moved.tolist()} "
          f"(by {dt_new[moved].tolist()}); all other draws differ by <= {np.delete(dt_new, moved).max():.2e}")
