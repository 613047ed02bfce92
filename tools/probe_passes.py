#!/usr/bin/env python3
"""Probe: skip_dead frame time vs export budget (number of passes), f16x2 + skip_dead + hybrid and f32 + skip_dead."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N
scene = os.path.join(ROOT, "lego_rust")
for gib in (16, 48, 64, 128):
    os.environ["NERF_MAX_EXPORT_BYTES"] = str(gib << 30)
    with N.Renderer(0) as r:
        r.load_scene(scene)
        cam = N.camera_from_samples(os.path.join(scene, "tf_reference_samples.json"), 800, 800, 64)
        for kw in (dict(dtype="f16x2", skip_dead=True, hybrid_sampling=True), dict(dtype="f32", skip_dead=True)):
            best = None
            for _ in range(4):
                _, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, return_stats=True, **kw)
                best = st if best is None or st.ms_total < best.ms_total else best
            print(f"budget {gib:4d} GiB  passes {best.n_passes}  {kw['dtype']:6s} hybrid={kw.get('hybrid_sampling', False)!s:5s}  {best.ms_total:7.1f} ms", flush=True)
