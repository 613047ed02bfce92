#!/usr/bin/env python3
"""Dump what hybrid_sampling's flag decides on, for offline modelling: coarse t, sigma in f32 and f16x2, the f16x2 weights / cdf, and the
draws (t_new) from both density sets (same uniforms) for a window of the C3 frame -> gpurun_out/hyb_case_<name>.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
name = sys.argv[1]; x0, y0, w, h = (int(v) for v in sys.argv[2:6]); deg = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
W = 800; nc, nf = 64, 128
import json
S = json.load(open(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")))
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    if deg:
        c2w = np.array(S["camera_matrix"], np.float64); a = np.deg2rad(deg)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        cam = N.camera_from_pose(np.concatenate([R @ c2w[:, :3], (R @ c2w[:, 3])[:, None]], axis=1), S["hwf"], S["near"], S["far"], W, W, nc)
    else:
        cam = N.camera_from_samples(S, W, W, nc)
    t = r.stage_stratified(cam, x0, y0, w, h, nc, seed=0).reshape(-1, nc)
    dirs = r.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
    o = cam.pos.astype(np.float32)
    pts = (o[None, None, :] + dirs[:, None, :] * t[:, :, None]).astype(np.float32).reshape(-1, 3)
    dd = np.repeat(dirs, nc, axis=0)
    pix = ((y0 + np.arange(h))[:, None] * W + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
    sg = {}
    for dt in ("f32", "f16x2"):
        _, s = r.coarse.forward_batch(np.ascontiguousarray(pts.T), dd, dtype=dt)
        sg[dt] = s.reshape(-1, nc)
    a32 = r.stage_resample(t, sg["f32"], nf, 6.0, seed=0, pixel_index=pix)
    a16 = r.stage_resample(t, sg["f16x2"], nf, 6.0, seed=0, pixel_index=pix)
    flags, _ = r.stage_hybrid_flags(t, sg["f16x2"], nf, 6.0, seed=0, pixel_index=pix)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"hyb_case_{name}.npz"), t=t, s32=sg["f32"], s16=sg["f16x2"], tn32=a32["t_new"], tn16=a16["t_new"], flags=flags)
    print(name, "rays", t.shape[0], "flagged", flags.mean(), "movers", (np.abs(a16["t_new"] - a32["t_new"]).max(axis=1) > 1e-5).mean())
