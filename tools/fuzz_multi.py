#!/usr/bin/env python3
"""Fuzz of nerf_render_image_multi (the rayon fan-out + scatter of src/lib.rs:533-557 behind the C ABI): random numbers of contexts
(1..6, sharing the one GPU of the test box), random frame sizes, windows (ragged bands, fewer rows than contexts, one-pixel-wide
frames), SSAA, sample counts, seeds, arithmetics, skip modes and all three gathers -- every frame must be BIT-IDENTICAL to the
single-context render of the same options (per-pixel counter RNG, whole rows per band).  On a shared device the RCCL gather runs its
equal-slot layout + ragged compaction with the collective step rehearsed as device-to-device copies (RCCL proper needs distinct GPUs).
Usage: fuzz_multi.py [seconds] [rng seed]   (exit code 1 on a mismatch; tests/test_gpu_multi.py runs a short one)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N

SAMPLES = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")


def fuzz(renderers, budget, rng_seed):
    """renderers: a list of Renderers with the lego scene loaded (renderers[0] also renders the single-context reference)."""
    rng = np.random.default_rng(rng_seed)
    tot = dict(cases=0, rays=0, mismatching=0, by_gather={"host": 0, "peer": 0, "rccl": 0})
    t_end = time.time() + budget
    one = renderers[0]
    while time.time() < t_end:
        W, H = int(rng.choice([1, 7, 33, 64, 100, 257, 800])), int(rng.choice([1, 5, 32, 101, 333, 800]))
        nc, nf = [(64, 128), (32, 64), (20, 50), (40, 50), (7, 5), (16, 0)][int(rng.integers(6))]
        cam = N.camera_from_samples(SAMPLES, W, H, nc)
        w, h = int(rng.integers(1, min(W, 96) + 1)), int(rng.integers(1, min(H, 48) + 1))
        crop = (int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h)
        ssaa = 2 if rng.integers(5) == 0 else 1
        dtype = ["f32", "f32", "bf16x3", "f16x2", "bf16"][int(rng.integers(5))]
        mode = int(rng.integers(5))
        kw = dict(seed=int(rng.integers(0, 1 << 30)), crop=crop, ssaa=ssaa, dtype=dtype, coarse_only=(nf == 0))
        if mode == 1:
            kw["skip_empty"] = True
        elif mode == 4 and dtype != "bf16":
            kw["certify_zero"] = True     # (with skip_empty / skip_dead: rows round-robin instead of contiguous bands)
        elif mode >= 2:
            kw["skip_dead"] = True
            if mode == 3 and dtype != "bf16" and nf > 0:
                kw["hybrid_sampling"] = True
        n = int(rng.integers(1, len(renderers) + 1))
        gather = ["host", "peer", "rccl"][int(rng.integers(3))]
        ref = N.render_image(one.coarse, one.fine, cam, nf, **kw)
        img = N.render_image_multi(renderers[:n], cam, nf, gather=gather, **kw)
        tot["cases"] += 1; tot["rays"] += w * h * ssaa * ssaa; tot["by_gather"][gather] += 1
        if not np.array_equal(img, ref):
            tot["mismatching"] += 1
            print(f"MISMATCH: frame {W}x{H} crop {crop} ssaa {ssaa} {nc}+{nf} {kw} n {n} gather {gather}: max {np.abs(img - ref).max():.3e}", flush=True)
    return tot


if __name__ == "__main__":
    rs = [N.Renderer(0) for _ in range(6)]
    for r in rs:
        r.load_scene(os.path.join(ROOT, "lego_rust"))
    try:
        res = fuzz(rs, float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    finally:
        for r in rs:
            r.close()
    print(json.dumps(res))
    sys.exit(1 if res["mismatching"] else 0)
