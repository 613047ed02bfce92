#!/usr/bin/env python3
"""skip_dead in the bf16 arithmetic: bit-identity with the plain bf16 frame on crops / odd sample counts / coarse_only, then frame
times (best of n) at ssaa 1 and 2 (BASELINE config C5)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
    cam = N.camera_from_samples(S, 800, 800, 64)
    ok = True
    for nc, nf, crop, co in ((64, 128, (300, 300, 200, 64), False), (40, 50, (380, 360, 40, 24), False), (4, 0, (150, 150, 100, 40), True),
                             (64, 0, (200, 200, 300, 100), True), (33, 31, (0, 0, 800, 8), False), (64, 128, (395, 400, 1, 1), False)):
        c = N.camera_from_samples(S, 800, 800, nc)
        a = N.render_image(r.coarse, r.fine, c, nf, seed=1, crop=crop, coarse_only=co, dtype="bf16")
        b, st = N.render_image(r.coarse, r.fine, c, nf, seed=1, crop=crop, coarse_only=co, dtype="bf16", skip_dead=True, return_stats=True)
        same = np.array_equal(a, b)
        ok &= same
        print(f"nc {nc} nf {nf} crop {crop} coarse_only {co}: identical={same} passes {st.n_passes} launches {st.n_mlp_launches} exec colour {st.n_exec_colour}", flush=True)
    for ssaa in (1, 2):
        ref = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", ssaa=ssaa)
        for dead in (False, True):
            best = None
            for k in range(n):
                img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", ssaa=ssaa, skip_dead=dead, return_stats=True)
                if best is None or st.ms_total < best.ms_total:
                    best = st
            same = np.array_equal(img, ref)
            ok &= same
            print(f"ssaa {ssaa} skip_dead {dead}: identical={same} passes {best.n_passes} total {best.ms_total:.2f} ms coarse {best.ms_coarse_mlp:.2f} fine {best.ms_fine_mlp:.2f} "
                  f"other {best.ms_other:.2f}; exec coarse {best.n_exec_coarse_trunk / best.n_coarse_points:.4f} fine {best.n_exec_fine_trunk / best.n_fine_points:.4f} "
                  f"colour {best.n_exec_colour / best.n_fine_points:.4f}", flush=True)
    print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
    sys.exit(0 if ok else 1)
