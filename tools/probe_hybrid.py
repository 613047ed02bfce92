#!/usr/bin/env python3
"""Feasibility probe (not product code): how far do the hierarchical sample positions move when the COARSE densities come from
the f16x2 arithmetic instead of the f32 MFMA kernel, and how much of that sits in ill-conditioned (flat) CDF bins?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N

with N.Renderer(0) as r:
    scene = os.path.join(ROOT, "lego_rust")
    r.load_scene(scene)
    S = N.api.load_tf_samples(os.path.join(scene, "tf_reference_samples.json"))
    cam = N.camera_from_samples(S, 800, 800, 64)
    x0, y0, w, h = 300, 300, 200, 200
    dirs = r.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
    tc = r.stage_stratified(cam, x0, y0, w, h, 64, seed=0).reshape(-1, 64)
    R = dirs.shape[0]
    o = cam.pos.astype(np.float32)
    pts = (o[None, None, :] + (dirs[:, None, :] * tc[:, :, None]).astype(np.float32)).astype(np.float32)  # mul then add, f32
    P = np.ascontiguousarray(pts.reshape(-1, 3).T)
    D = np.repeat(dirs, 64, axis=0)
    s32 = r.coarse.forward_batch(P, D)[1].reshape(R, 64)
    s16 = r.coarse.forward_batch(P, D, dtype="f16x2")[1].reshape(R, 64)
    pix = ((y0 + np.arange(h))[:, None] * 800 + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
    a = r.stage_resample(tc, s32, 128, cam.far, seed=0, pixel_index=pix)
    b = r.stage_resample(tc, s16, 128, cam.far, seed=0, pixel_index=pix)
    dt = np.abs(a["t_new"] - b["t_new"])
    cdf = a["cdf"]                                      # R x 63
    dc = np.diff(cdf, axis=1)                           # R x 62 bin masses
    print(f"rays {R}; sigma rel diff max {np.abs(s32 - s16).max() / (1 + np.abs(s32).max()):.2e}; |dt_new| max {dt.max():.3e} mean {dt.mean():.3e}")
    # which bin did each draw land in (from the f32 cdf) -> its mass
    bins = 0.5 * (tc[:, :-1] + tc[:, 1:])               # R x 63 edges
    tn = a["t_new"]
    j = np.clip((tn[:, :, None] >= bins[:, None, :]).sum(-1) - 1, 0, 61)
    mass = np.take_along_axis(dc, j, axis=1)
    for tau in (1e-5, 1e-4, 1e-3, 1e-2):
        ill = mass < tau
        flagged = ill.any(1)
        print(f"tau {tau:g}: draws in bins lighter than tau {ill.mean():.4f}; rays flagged {flagged.mean():.4f}; "
              f"max |dt| over UNflagged rays {dt[~flagged].max() if (~flagged).any() else 0:.3e}, over well-conditioned draws {dt[~ill].max():.3e}")
    fg = (s32 > 0).any(1)
    print("foreground rays in window", fg.mean())
