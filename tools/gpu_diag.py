#!/usr/bin/env python3
"""GPU diagnostic: forward_batch error statistics vs the committed oracle fixture, run-to-run determinism."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N

g = np.load(os.path.join(ROOT, "tests/golden/forward_batch_4096.npz"))
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    print(r.device_info())
    for name, net in (("coarse", r.coarse), ("fine", r.fine)):
        rgb, sg = net.forward_batch(g["pts"], g["dirs"])
        rgb2, sg2 = net.forward_batch(g["pts"], g["dirs"])
        es, er = g[f"{name}_sigma"], g[f"{name}_rgb"]
        ds = np.abs(sg - es) / (1 + np.abs(es)); dr = np.abs(rgb - er)
        print(f"{name}: sigma rel err max {ds.max():.3e} mean {ds.mean():.3e} frac>1e-4 {np.mean(ds > 1e-4):.4f} | "
              f"rgb err max {dr.max():.3e} mean {dr.mean():.3e} frac>2e-5 {np.mean(dr > 2e-5):.4f} | "
              f"deterministic {np.array_equal(sg, sg2) and np.array_equal(rgb, rgb2)} finite {np.isfinite(sg).all()}")
        bad = np.where(ds > 1e-4)[0][:8]
        if len(bad):
            print("  first bad points:", bad.tolist(), "lane pos (i%128):", (bad % 128).tolist())
            print("  got", sg[bad], "\n  exp", es[bad])
