#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): bf16x3 vs the f32 MFMA kernel vs the oracle on out-of-distribution inputs
(points far outside the scene, un-normalised and tiny view directions)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_rs_amd as N
import oracle_py as O
O.build()
rng = np.random.default_rng(9)
n = 32768
cases = {"box +-2.2": (rng.uniform(-2.2, 2.2, (3, n)), rng.normal(size=(n, 3))),
         "box +-20": (rng.uniform(-20, 20, (3, n)), rng.normal(size=(n, 3)) * 5),
         "box +-300": (rng.uniform(-300, 300, (3, n)), rng.normal(size=(n, 3)) * 1e-3),
         "near origin 1e-4": (rng.normal(size=(3, n)) * 1e-4, rng.normal(size=(n, 3)))}
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    for sub, net in (("coarse", r.coarse), ("fine", r.fine)):
        onet = O.Net(os.path.join(ROOT, "lego_rust", sub))
        for name, (p, d) in cases.items():
            p = p.astype(np.float32); d = d.astype(np.float32)
            ergb, esg = onet.forward_batch(p, d)
            out = {}
            for dt in ("f32", "bf16x3"):
                rgb, sg = net.forward_batch(p, d, dtype=dt)
                out[dt] = (np.abs(sg - esg) / (1 + np.abs(esg))).max(), np.abs(rgb - ergb).max(), np.isfinite(sg).all() and np.isfinite(rgb).all()
            print(f"{sub:6s} {name:18s} sigma max {esg.max():9.2f} | f32 {out['f32'][0]:.2e} {out['f32'][1]:.2e} | bf16x3 {out['bf16x3'][0]:.2e} {out['bf16x3'][1]:.2e} | finite {out['f32'][2]} {out['bf16x3'][2]}")
