#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): explain the largest per-pixel difference between a GPU f32 frame and the oracle's frame of a
view that tests/gpu_fuzz_vs_oracle.py reported.  Finds the pixel, then replays its ray: coarse densities on both sides, the CDF, which
fine draws move, how far -- a relocated fine sample (hierarchical sampling is ill-conditioned there) or something else?
Usage: python tests/gpu_diag_oracle_outlier.py W H deg tilt nc nf seed"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import nerf_rs_amd as N
import oracle_py as O
from scene_utils import pose as _pose, oracle_samples as _oracle_samples

W, H = int(sys.argv[1]), int(sys.argv[2]); deg, tilt = float(sys.argv[3]), float(sys.argv[4]); nc, nf, seed = (int(v) for v in sys.argv[5:8])
S = O.load_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"))
O.build()
onets = (O.Net(os.path.join(ROOT, "lego_rust", "coarse")), O.Net(os.path.join(ROOT, "lego_rust", "fine")))
m = _pose(S, deg, tilt)
ocam = O.camera_from_samples(_oracle_samples(S, m), W, H)
opts = O.make_opts(nc, nf, seed=seed)
ref = O.render_image(*onets, ocam, opts)
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_pose(m, S["hwf"], S["near"], S["far"], W, H, nc)
    img = N.render_image(r.coarse, r.fine, cam, nf, seed=seed)
    d = np.abs(img - ref).max(axis=2)
    i, j = np.unravel_index(np.argmax(d), d.shape)
    print(f"largest difference {d[i, j]:.3e} at pixel (row {i}, col {j}); GPU {img[i, j]}, oracle {ref[i, j]}; pixels > 5e-4: {(d > 5e-4).sum()}, > 1e-4: {(d > 1e-4).sum()}, mean {np.abs(img - ref).mean():.2e}")
    dump = O.render_ray_debug(*onets, ocam, opts, int(i), int(j))
    dirs = r.stage_ray_dirs(cam, j, i, 1, 1)[0, 0]
    tc = r.stage_stratified(cam, j, i, 1, 1, nc, seed=seed)[0]
    print(f"ray direction identical: {np.array_equal(dirs, dump['dir_hat'])}; coarse t identical: {np.array_equal(tc[0], dump['t_coarse'])}")
    org = cam.pos.astype(np.float32)
    pts = (org[:, None] + dirs[:, None] * tc[0][None, :]).astype(np.float32)
    _, sg = r.coarse.forward_batch(pts, np.tile(dirs, (nc, 1)))
    so = dump["sigma_coarse"]
    print(f"coarse sigma GPU vs oracle: max abs diff {np.abs(sg - so).max():.3e}, max rel {np.max(np.abs(sg - so) / np.maximum(np.abs(so), 1e-3)):.3e}; zero pattern identical: {np.array_equal(sg == 0, so == 0)}")
    pix = np.array([i * W + j], np.uint32)
    a = r.stage_resample(tc, sg[None, :], nf, cam.far, seed=seed, pixel_index=pix)
    b = r.stage_resample(tc, so[None, :], nf, cam.far, seed=seed, pixel_index=pix)
    mv = np.abs(a["t_new"] - b["t_new"])[0]
    moved = np.nonzero(mv > 1e-4)[0]
    print(f"same resampling kernel on the two density sets: cdf max diff {np.abs(a['cdf'] - b['cdf']).max():.2e}; draws that move by > 1e-4: {moved.tolist()} (by {mv[moved].tolist()}); "
          f"all other draws differ by <= {np.delete(mv, moved).max() if moved.size < mv.size else 0:.2e}")
    print(f"GPU resampling of the ORACLE's densities vs the oracle's own draws: max diff {np.abs(np.sort(b['t_new'][0]) - np.sort(dump['t_new'])).max():.2e}")
    flags, _ = r.stage_hybrid_flags(tc, sg[None, :], nf, cam.far, seed=seed, pixel_index=pix)
    print(f"hybrid_sampling's flag for this ray (a model of a 10x larger density error): {bool(flags[0])}")
    w = dump["w_coarse"]; cdf = dump["cdf"]
    print("oracle coarse weights", np.array2string(w, precision=4, max_line_width=200))
    print("oracle cdf", np.array2string(cdf, precision=6, max_line_width=200))
