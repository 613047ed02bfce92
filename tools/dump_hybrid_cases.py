#!/usr/bin/env python3
"""Dump, for offline modelling of hybrid_sampling's flag, several (scene, pose, window, sample-count) cases in ONE process:
per ray the coarse t, the coarse densities in f32 and in the split arithmetics, the present flag, and the per-draw displacement of the
split-arithmetic draws against the f32 draws (same uniforms; f16 is plenty for a displacement)
-> gpurun_out/hyb_cases.npz.  Usage: dump_hybrid_cases.py [out-name]"""
import json
import os
import sys
import tempfile
import pathlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerf_rs_amd as N
from scene_utils import pose as _pose, random_scene as _random_scene

S = json.load(open(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")))
FAR = float(S["far"])
# name, scene, (deg, tilt) or None, frame side, window, pixel stride inside the window, (nc, nf), seed
CASES = [
    ("c3_centre", "lego", None, 800, (200, 200, 400, 400), 4, (64, 128), 0),
    ("c3_left", "lego", None, 800, (40, 300, 300, 300), 4, (64, 128), 0),
    ("c3_top", "lego", None, 800, (250, 60, 300, 240), 4, (64, 128), 0),
    ("pose130", "lego", (130, 0), 800, (150, 150, 500, 500), 6, (64, 128), 7),
    ("pose250_32", "lego", (250, 12), 400, (60, 60, 280, 280), 3, (32, 64), 7),
    ("pose75_20", "lego", (75, 0), 200, (20, 20, 160, 160), 2, (20, 50), 3),
    ("t_json", "lego", None, 128, (16, 16, 96, 96), 1, (64, 128), 7),       # the windows of tests/test_gpu_hybrid_validation.py
    ("t_pose130", "lego", (130, 0), 128, (16, 16, 96, 96), 1, (64, 128), 7),
    ("t_pose250", "lego", (250, 12), 128, (16, 16, 96, 96), 1, (32, 64), 7),
    ("p40_96", "lego", (40, 5), 96, (0, 0, 96, 96), 1, (64, 128), 11),
    ("p300_64", "lego", (300, -8), 64, (0, 0, 64, 64), 1, (48, 96), 13),
    ("fog99", "rnd99", None, 128, (16, 16, 96, 96), 1, (64, 128), 7),
    ("fog321", "rnd321", None, 128, (8, 8, 112, 112), 2, (64, 128), 5),
]


def main():
    out = {}
    tmp = pathlib.Path(tempfile.mkdtemp())
    with N.Renderer(0) as r:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
        for name, scene, pose, W, (x0, y0, w, h), stride, (nc, nf), seed in CASES:
            if scene == "lego":
                net = r.coarse
            else:  # replaces slot 0 (the lego cases come first)
                net = N.load_network_from_dir(r, 0, _random_scene(tmp / scene, int(scene[3:])) / "coarse")
            cam = (N.camera_from_samples(S, W, W, nc) if pose is None else
                   N.camera_from_pose(_pose(S, *pose), S["hwf"], S["near"], S["far"], W, W, nc))
            t = r.stage_stratified(cam, x0, y0, w, h, nc, seed=seed).reshape(-1, nc)
            dirs = r.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
            o = cam.pos.astype(np.float32)
            pix = ((y0 + np.arange(h))[:, None] * W + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
            keep = ((np.arange(h) % stride == 0)[:, None] & (np.arange(w) % stride == 0)[None, :]).reshape(-1)
            t, dirs, pix = np.ascontiguousarray(t[keep]), np.ascontiguousarray(dirs[keep]), np.ascontiguousarray(pix[keep])
            pts = (o[None, None, :] + dirs[:, None, :] * t[:, :, None]).astype(np.float32).reshape(-1, 3)
            dd = np.repeat(dirs, nc, axis=0)
            sg = {}
            for dt in ("f32", "f16x2", "bf16x3"):
                _, s = net.forward_batch(np.ascontiguousarray(pts.T), dd, dtype=dt)
                sg[dt] = s.reshape(-1, nc)
            a32 = r.stage_resample(t, sg["f32"], nf, FAR, seed=seed, pixel_index=pix)
            out[name + "/t"] = t
            out[name + "/s32"] = sg["f32"]
            out[name + "/meta"] = np.array([nc, nf, seed, W, x0, y0, w, h, stride], np.int64)
            out[name + "/pix"] = pix
            for dt in ("f16x2", "bf16x3"):
                flags, tn = r.stage_hybrid_flags(t, sg[dt], nf, FAR, seed=seed, pixel_index=pix)
                mv = np.abs(tn - a32["t_new"])
                if dt == "f16x2" or name in ("c3_centre", "fog99"):   # the 64-MiB cap of gpurun_out
                    out[f"{name}/s_{dt}"] = sg[dt]
                out[f"{name}/move_{dt}"] = mv.astype(np.float16)
                out[f"{name}/flags_{dt}"] = flags
                print(f"{name} {dt}: rays {t.shape[0]} flagged {flags.mean():.4f} movers {(mv.max(axis=1) > 1e-5).mean():.4f} "
                      f"unflagged max {mv.max(axis=1)[~flags].max() if (~flags).any() else 0:.2e}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", (sys.argv[1] if len(sys.argv) > 1 else "hyb_cases") + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 20, "MiB")


if __name__ == "__main__":
    main()
