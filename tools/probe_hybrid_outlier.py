#!/usr/bin/env python3
"""Diagnose the largest pixel difference between hybrid_sampling and f32-sampling frames (f16x2 + skip_dead)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
scene = os.path.join(ROOT, "lego_rust")
with N.Renderer(0) as r:
    r.load_scene(scene)
    S = N.api.load_tf_samples(os.path.join(scene, "tf_reference_samples.json"))
    cam = N.camera_from_samples(S, 800, 800, 64)
    base = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", skip_dead=True)
    hyb = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="f16x2", skip_dead=True, hybrid_sampling=True)
    d = np.abs(hyb - base).max(axis=2)
    order = np.argsort(d.reshape(-1))[::-1][:6]
    for idx in order:
        i, j = divmod(int(idx), 800)
        print(f"pixel ({i},{j}) diff {d[i, j]:.3e}  base {base[i, j]}  hyb {hyb[i, j]}")
        dirs = r.stage_ray_dirs(cam, j, i, 1, 1).reshape(1, 3)
        tc = r.stage_stratified(cam, j, i, 1, 1, 64, seed=0).reshape(1, 64)
        o = cam.pos.astype(np.float32)
        pts = (o[None, :] + (dirs * tc.T).astype(np.float32)).astype(np.float32)
        P = np.ascontiguousarray(pts.T); D = np.repeat(dirs, 64, axis=0)
        s32 = r.coarse.forward_batch(P, D)[1].reshape(1, 64)
        s16 = r.coarse.forward_batch(P, D, dtype="f16x2")[1].reshape(1, 64)
        pix = np.array([i * 800 + j], np.uint32)
        a = r.stage_resample(tc, s32, 128, cam.far, seed=0, pixel_index=pix)
        b = r.stage_resample(tc, s16, 128, cam.far, seed=0, pixel_index=pix)
        dt = np.abs(a["t_new"] - b["t_new"])[0]
        dc = np.diff(a["cdf"][0])
        bins = 0.5 * (tc[0, :-1] + tc[0, 1:])
        jj = np.clip((a["t_new"][0][:, None] >= bins[None, :]).sum(-1) - 1, 0, 61)
        mass = dc[jj]
        k = int(np.argmax(dt))
        T = np.cumprod(1 - (1 - np.exp(-s32[0] * np.diff(np.append(tc[0], cam.far)).clip(0))))
        print(f"   max |dt_new| {dt.max():.3e} at draw {k} (bin mass {mass[k]:.3e}); min bin mass over draws {mass.min():.3e}; "
              f"sigma rel diff {np.abs(s32 - s16).max() / (1 + np.abs(s32).max()):.2e}; coarse T min {T.min():.3e}; #draws moved >1e-5: {(dt > 1e-5).sum()}")
