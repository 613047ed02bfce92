#!/usr/bin/env python3
"""Diagnostic script (not a pytest module; run by hand on the GPU box).  bf16 MLP kernel: arithmetic check against the oracle's bf16 emulation + frame timing / PSNR gates."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_rs_amd as N
import oracle_py as O

g = np.load(os.path.join(ROOT, "tests/golden/forward_batch_4096.npz"))
def psnr(a, b):
    m = float(np.mean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)); return 99.0 if m == 0 else 10 * np.log10(1 / m)
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    for name, net in (("coarse", r.coarse), ("fine", r.fine)):
        onet = O.Net(os.path.join(ROOT, "lego_rust", name))
        ergb, esg = onet.forward_batch_bf16(g["pts"], g["dirs"])
        rgb, sg = net.forward_batch(g["pts"], g["dirs"], dtype="bf16")
        ds = np.abs(sg - esg) / (1 + np.abs(esg)); dr = np.abs(rgb - ergb)
        d32 = np.abs(sg - g[f"{name}_sigma"]) / (1 + np.abs(g[f"{name}_sigma"]))
        print(f"{name}: vs bf16 emulation sigma rel max {ds.max():.2e} mean {ds.mean():.2e} p99 {np.quantile(ds, .99):.2e} | rgb max {dr.max():.2e} "
              f"mean {dr.mean():.2e} || vs f32 oracle sigma rel max {d32.max():.3f} mean {d32.mean():.4f} rgb max {np.abs(rgb - g[name + '_rgb']).max():.3f}")
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    for dt in ("f32", "bf16", "bf16"):
        img, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype=dt, return_stats=True)
        print(f"{dt}: total {st.ms_total:.1f} ms coarse {st.ms_coarse_mlp:.1f} fine {st.ms_fine_mlp:.1f} other {st.ms_other:.1f} -> {st.n_rays / st.ms_total * 1e3:.0f} rays/s; "
              f"fine {st.n_fine_points * 1186816 / st.ms_fine_mlp / 1e9:.1f} TFLOP/s")
        if dt == "f32": ref = img
    a = np.load(os.path.join(ROOT, "tests/golden/crop_c3_800_64_128.npz")); b = np.load(os.path.join(ROOT, "tests/golden/crop_c3_800_64_128_seed1.npz"))
    x0, y0, w, h = (int(v) for v in a["crop"])
    img1 = N.render_image(r.coarse, r.fine, cam, 128, seed=1, dtype="bf16", crop=(x0, y0, w, h))
    print(f"bf16 vs f32 GPU frame (same seed): PSNR {psnr(img, ref):.2f} dB, max|d| {np.abs(img - ref).max():.3f} mean {np.abs(img - ref).mean():.2e}")
    print(f"Gate 2 on the C3 crop: PSNR(bf16 seed1, CPU seed0) {psnr(img1, a['image']):.2f} dB vs PSNR(CPU seed1, CPU seed0) {psnr(b['image'], a['image']):.2f} dB")
