#!/usr/bin/env python3
"""The S2 seam (Network::forward_batch, src/network.rs:197-237) on random points against the LIVE oracle, batch after batch: positions
uniform in a box of half-width R (the lego frustum reaches |p| <= 2.42; R is drawn from {1.5, 2.5, 4, 8}), random unit directions,
both networks, in f32, bf16x3 and f16x2 -- each point held to the f32 tolerances of the parity tests, |d sigma| <= 1e-4 (1 + |sigma|),
|d rgb| <= 2e-5.  Reports per arithmetic the points beyond a tolerance and the largest errors relative to it.
Test infrastructure (lives under tests/: it imports oracle/).
Usage: python tests/gpu_fuzz_forward.py [seconds] [rng seed]   (exit code 1 if any point of the scene box |p| <= 2.5 violates a tolerance)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nerf_rs_amd as N
import oracle_py as O


def fuzz(r, onets, budget, rng_seed, batch=1 << 18):
    rng = np.random.default_rng(rng_seed)
    tot = {dt: dict(points=0, sigma_violations=0, rgb_violations=0, worst_sigma=0.0, worst_rgb=0.0, violations_in_scene_box=0) for dt in ("f32", "bf16x3", "f16x2")}
    t_end = time.time() + budget
    batches = 0
    while time.time() < t_end:
        R = float(rng.choice([1.5, 2.5, 4.0, 8.0]))
        pts = rng.uniform(-R, R, size=(3, batch)).astype(np.float32)
        d = rng.normal(size=(batch, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        for net, onet in ((r.coarse, onets[0]), (r.fine, onets[1])):
            ergb, esg = onet.forward_batch(pts, d)
            for dt in tot:
                rgb, sg = net.forward_batch(pts, d, dtype=dt)
                es = np.abs(sg - esg) / (1e-4 * (1 + np.abs(esg)))      # in units of the tolerance
                er = np.abs(rgb - ergb).max(axis=1) / 2e-5
                T = tot[dt]
                T["points"] += batch
                T["sigma_violations"] += int((es > 1).sum()); T["rgb_violations"] += int((er > 1).sum())
                T["worst_sigma"] = max(T["worst_sigma"], float(es.max())); T["worst_rgb"] = max(T["worst_rgb"], float(er.max()))
                if R <= 2.5:
                    T["violations_in_scene_box"] += int(((es > 1) | (er > 1)).sum())
                if not np.isfinite(sg).all() or not np.isfinite(rgb).all():
                    T["violations_in_scene_box"] += 1
                    print(f"NON-FINITE output: {dt} R {R}", flush=True)
        batches += 1
        if batches % 4 == 0:
            print(f"... {batches} batches", {k: (v["points"], v["sigma_violations"], v["rgb_violations"], round(v["worst_sigma"], 3), round(v["worst_rgb"], 3)) for k, v in tot.items()}, flush=True)
    return tot


if __name__ == "__main__":
    O.build()
    onets = (O.Net(os.path.join(ROOT, "lego_rust", "coarse")), O.Net(os.path.join(ROOT, "lego_rust", "fine")))
    with N.Renderer(0) as r:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
        res = fuzz(r, onets, float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(json.dumps(res))
    sys.exit(1 if any(v["violations_in_scene_box"] for v in res.values()) else 0)
