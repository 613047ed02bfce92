#!/usr/bin/env python3
"""Fuzz of zero certification (nerf_render_opts.certify_zero): random poses (any azimuth, +-25 degrees tilt), frame sizes, windows,
sample counts, seeds, SSAA, coarse-only -- in f32 and in the two split arithmetics; the certified frame must be the plain frame of the same arithmetic BIT FOR BIT: one wrong certificate (a sample
the bf16 pass declares a certain zero while the f32 network gives it a density) that reaches a pixel shows up as a mismatch.
Usage: fuzz_certify.py [seconds] [rng seed]   (exit code 1 on a mismatch; tests/test_gpu_certify.py runs a short one)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerf_rs_amd as N
from scene_utils import pose

S = json.load(open(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")))


def fuzz(r, budget, rng_seed):
    rng = np.random.default_rng(rng_seed)
    tot = dict(cases=0, rays=0, mismatching=0, f32_samples_nominal=0, f32_samples_evaluated=0, retries=0, violations=0, audited=0, fallback_rays=0,
               worst_error_over_margin=[0.0, 0.0], least_headroom_over_margin=[9.9, 9.9], margins=None)
    t_end = time.time() + budget
    while time.time() < t_end:
        W = int(rng.choice([64, 128, 200, 400, 800]))
        nc, nf = [(64, 128), (64, 128), (48, 96), (32, 64), (20, 50), (33, 77), (64, 0), (7, 5)][int(rng.integers(8))]
        cam = N.camera_from_pose(pose(S, float(rng.uniform(0, 360)), float(rng.uniform(-25, 25))), S["hwf"], S["near"], S["far"], W, W, nc)
        w, h = int(rng.integers(1, min(W, 200) + 1)), int(rng.integers(1, min(W, 48) + 1))
        kw = dict(seed=int(rng.integers(0, 1 << 30)), crop=(int(rng.integers(0, W - w + 1)), int(rng.integers(0, W - h + 1)), w, h),
                  ssaa=2 if rng.integers(6) == 0 else 1, coarse_only=(nf == 0))
        kw["dtype"] = str(rng.choice(["f32", "f32", "f16x2", "bf16x3"]))  # split arithmetics: f32 certified coarse pass + certified fine pass
        try:
            ref = N.render_image(r.coarse, r.fine, cam, nf, **kw)
        except N.NerfError as e:   # a hot network may leave the f16 range: that arithmetic fails loudly by design, with or without certify_zero
            if e.code == -6 and kw["dtype"] == "f16x2" and "left the range" in e.msg:
                tot["f16_range_errors"] = tot.get("f16_range_errors", 0) + 1
                continue
            raise
        try:
            img, st = N.render_image(r.coarse, r.fine, cam, nf, certify_zero=True, return_stats=True, **kw)
        except N.NerfError as e:   # a network the 16-bit passes cannot certify fails loudly (NERF_ERR_STATE), never silently differently
            if e.code == -6 and "certify_zero" in e.msg:
                tot["failed_loudly"] = tot.get("failed_loudly", 0) + 1
                continue
            raise
        tot["cases"] += 1; tot["rays"] += st.n_rays
        tot["f32_samples_nominal"] += st.n_coarse_points + st.n_fine_points
        tot["f32_samples_evaluated"] += st.n_exec_coarse_trunk + st.n_exec_fine_trunk
        tot["retries"] += st.n_certify_retries; tot["violations"] += st.n_certify_violations; tot["audited"] += st.n_certify_audited
        tot["fallback_rays"] += st.n_certify_fallback_rays; tot["margins"] = list(st.certify_margin)
        for w in range(2):
            tot["worst_error_over_margin"][w] = max(tot["worst_error_over_margin"][w], st.certify_max_error[w] / st.certify_margin[w])
            tot["least_headroom_over_margin"][w] = min(tot["least_headroom_over_margin"][w], st.certify_headroom[w] / st.certify_margin[w])
        if not np.array_equal(img, ref):
            tot["mismatching"] += 1
            d = np.abs(img - ref)
            print(f"MISMATCH: W {W} {nc}+{nf} {kw}: max {d.max():.3e}, {int((d.max(axis=2) > 0).sum())} pixels", flush=True)
    return tot


def scaled_scene(root, scale):
    """lego with dense7 (kernel and bias) scaled: every density pre-activation, and the 16-bit pass's error on it, grows by `scale`.
    scale < 0 ("big" on the command line): dense0 x 2000 and dense1 x 50 instead -- activations beyond the f16 range, so the f16 pre-filter
    must hand the network over to the bf16 one (and that one must calibrate itself on pre-activations of 1e6)."""
    import shutil
    for which in ("coarse", "fine"):
        shutil.copytree(os.path.join(ROOT, "lego_rust", which), os.path.join(root, which))
        for t, k in ((("dense7_kernel", scale), ("dense7_bias", scale)) if scale > 0 else
                     (("dense0_kernel", 2000.0), ("dense0_bias", 2000.0), ("dense1_kernel", 50.0), ("dense1_bias", 50.0))):
            f = os.path.join(root, which, t + ".bin")
            (np.fromfile(f, "<f4") * np.float32(k)).astype("<f4").tofile(f)
    return root


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    scales = [-1.0 if x == "big" else float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1.0]   # e.g. 1,3,10,40,big: hotter networks than lego
    bad = 0
    for k, scale in enumerate(scales):
        import tempfile
        with tempfile.TemporaryDirectory() as tmp, N.Renderer(0) as r:
            r.load_scene(os.path.join(ROOT, "lego_rust") if scale == 1.0 else scaled_scene(os.path.join(tmp, "s"), scale))
            res = fuzz(r, budget / len(scales), seed + k)
        res["pre_activation_scale"] = scale
        print(json.dumps(res), flush=True)
        bad += res["mismatching"]
    sys.exit(1 if bad else 0)
