#!/usr/bin/env python3
"""One-GPU predictor of the strong-scaling curve: render each of the N = 2, 4, 8 bands of the C3 frame SERIALLY on this GPU, per
partition (contiguous bands / single rows round-robin / 8-row stripes -- nerf_render_opts.band_*) and per mode, and report
    balance    = mean band time / slowest band time      (1.0 = every GPU finishes together)
    efficiency = whole-frame time / (N x slowest band)   (predicted strong-scaling efficiency of the render; the gather adds
                                                          0.96 MB per rank, ~0.1 ms)
The reference's counterpart is rayon's work stealing over 8x8 blocks (src/lib.rs:533-550), which balances dynamically; a static
partition has to know where the cost is: plain renders cost the same for every ray, skip_dead / certify_zero follow the scene.
Usage: band_balance.py [repetitions] > profiles/r04_band_balance.jsonl   (table on stderr)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")
MODES = [("f32 (headline)", dict()), ("f32 + skip_dead", dict(skip_dead=True)), ("f32 + certify_zero", dict(certify_zero=True)),
         ("f16x2 + certify_zero", dict(dtype="f16x2", certify_zero=True)), ("bf16 2x2 SSAA (C5)", dict(dtype="bf16", ssaa=2))]
PARTS = [("contiguous", 0), ("rows round-robin", 1), ("8-row stripes", 8)]
rows = []
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(S, 800, 800, 64)

    def ms(kw, band=None):
        best = None
        for _ in range(reps):
            _, st = N.render_image(r.coarse, r.fine, cam, 128, seed=0, band=band, return_stats=True, **kw)
            best = st.ms_total if best is None else min(best, st.ms_total)
        return best

    for name, kw in MODES:
        ms(kw)  # warm (clock, list capacity, margins)
        whole = ms(kw)
        for n in (2, 4, 8):
            for pname, stripe in PARTS:
                t = [ms(kw, (i, n, stripe)) for i in range(n)]
                rec = {"mode": name, "n": n, "partition": pname, "band_stripe_rows": stripe, "whole_frame_ms": whole, "band_ms": t,
                       "balance": sum(t) / n / max(t), "predicted_efficiency": whole / (n * max(t)), "sum_of_bands_over_whole": sum(t) / whole}
                rows.append(rec)
                print(json.dumps(rec), flush=True)
print(f"{'mode':24s} {'N':>2s}  " + "  ".join(f"{p:>22s}" for p, _ in PARTS), file=sys.stderr)
for name, _ in MODES:
    for n in (2, 4, 8):
        cells = []
        for pname, _ in PARTS:
            rec = [x for x in rows if x["mode"] == name and x["n"] == n and x["partition"] == pname][0]
            cells.append(f"bal {rec['balance']:.3f} eff {rec['predicted_efficiency']:.3f}")
        print(f"{name:24s} {n:2d}  " + "  ".join(f"{c:>22s}" for c in cells), file=sys.stderr)
