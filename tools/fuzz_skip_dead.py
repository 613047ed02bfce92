#!/usr/bin/env python3
"""Bit-identity fuzz of skip_dead against the fused kernels: random crop windows, sample counts, seeds, SSAA, coarse_only, arithmetics.
The f32 path's in-LDS staging has data-dependent control flow (how many columns each wave deposits, when a colour pass runs, partial
final flushes, idle waves); every combination must still reproduce the non-skipping frame bit for bit.
    python tools/fuzz_skip_dead.py [seconds] [rng seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); n = 0; bad = 0; rays = 0
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
    while time.time() - t0 < budget:
        size = int(rng.choice([64, 200, 400, 800]))
        nc = int(rng.choice([3, 7, 20, 32, 33, 40, 64, 65, 96]))
        cam = N.camera_from_samples(S, size, size, nc)
        w = int(rng.integers(1, min(size, 260) + 1)); h = int(rng.integers(1, min(size, 24) + 1))
        x0 = int(rng.integers(0, size - w + 1)); y0 = int(rng.integers(0, size - h + 1))
        nf = int(rng.choice([0, 5, 50, 64, 95, 128, 160]))
        kw = dict(seed=int(rng.integers(0, 1 << 30)), crop=(x0, y0, w, h))
        if rng.random() < 0.15: kw["ssaa"] = 2
        if rng.random() < 0.15: kw["coarse_only"] = True
        dt = str(rng.choice(["f32", "f32", "f32", "f16x2", "bf16x3", "bf16"]))
        ref = N.render_image(r.coarse, r.fine, cam, nf, dtype=dt, **kw)
        img = N.render_image(r.coarse, r.fine, cam, nf, dtype=dt, skip_dead=True, **kw)
        n += 1; rays += w * h * kw.get("ssaa", 1) ** 2
        if not np.array_equal(img, ref):
            bad += 1
            print("MISMATCH", size, nc, nf, dt, kw, float(np.abs(img - ref).max()), flush=True)
        if n % 50 == 0:
            print(f"{n} cases, {rays} rays, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_skip_dead: {n} cases, {rays} rays, {bad} mismatching in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
