#!/usr/bin/env python3
"""Whole-frame fixtures from the CPU oracle (oracle/nerf_oracle.c) for the GPU suite, which cannot run the oracle at
this size (the GPU box has no /root/reference and the -m gpu suite must stay short).

    python tests/golden/make_golden_frames.py [threads] [--parts=seed0,seed1,ssaa2]     # ~20 min on 16 fast cores, hours on 8 slow ones; rewrites the two files below

  frame_c3_800_seed0.npz   BASELINE config C3 (lego 800x800, 64 + 128 samples/ray, f32), seed 0: the oracle's whole frame
                           (`image`, 800 x 800 x 3 f32 linear RGB) and its save_ppm quantisation (`rgb8`, src/lib.rs:573-577)
  frame_gates.json         PSNR numbers of the north star's Gate 2, all against A = that seed-0 frame:
                             cpu_seed1_vs_A            the oracle's seed-1 frame (the jitter noise floor, ~40 dB)
                             cpu_ssaa2_seed1_vs_A      the oracle's seed-1 frame at BASELINE config C5's geometry (800x800 output,
                                                       2x2 rays per pixel = 1600x1600 = 2 560 000 rays, box filter)
                           and of the north star's last clause ("PSNR within 0.1 dB of CPU output.ppm"), part `ppm512`:
                             cpu_pi8_512_seed{0,1}_vs_output_ppm_psnr8   8-bit PSNR of the oracle's whole 512x512 frame -- the geometry of the
                                                       reference's own output.ppm: half field of view pi/8 -- against that file
                                                       (tests/golden/reference/output.ppm); the GPU frame must land within 0.1 dB of it
Same chain of trust as make_golden.py: the oracle's MLP is pinned by the reference's 120 golden scalars; the frame is the
oracle's line-by-line restatement of render_image (src/lib.rs:474-565) with the seeded counter RNG.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

SCENE = os.path.join(ROOT, "lego_rust")
OUT = os.environ.get("NERF_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")  # NERF_GOLDEN_OUT: write elsewhere (e.g. gpurun_out/)


def psnr(a, b):
    mse = float(np.mean((np.clip(a, 0, 1).astype(np.float64) - np.clip(b, 0, 1).astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / mse)


def heartbeat(t0):
    """A progress line per minute (a silent 12-minute SSAA frame looks hung to a job runner)."""
    import threading

    def beat():
        while True:
            time.sleep(60)
            print("  ... %.0f s" % (time.time() - t0), flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    threads = int(args[0]) if args else len(os.sched_getaffinity(0))
    parts = "seed0,seed1,ssaa2"
    for a in sys.argv[1:]:
        if a.startswith("--parts="):
            parts = a.split("=", 1)[1]   # e.g. --parts=ssaa2: reuse the committed seed-0 frame, add one number to frame_gates.json
    parts = parts.split(",")
    os.makedirs(OUT, exist_ok=True)
    S = O.load_samples(os.path.join(SCENE, "tf_reference_samples.json"))
    co, fi = O.Net(os.path.join(SCENE, "coarse")), O.Net(os.path.join(SCENE, "fine"))
    cam = O.camera_from_samples(S, 800, 800)
    committed = os.path.join(ROOT, "tests", "golden")
    gates_path = os.path.join(OUT, "frame_gates.json")
    gates = {"config": "lego 800x800, 64 + 128 samples/ray, f32 oracle; A = seed-0 frame (frame_c3_800_seed0.npz)"}
    for cand in (gates_path, os.path.join(committed, "frame_gates.json")):
        if os.path.exists(cand):
            gates.update(json.load(open(cand)))
            break
    t0 = time.time()
    heartbeat(t0)
    if "seed0" in parts:
        a = O.render_image(co, fi, cam, O.make_opts(64, 128, seed=0, threads=threads))
        gates["oracle_seconds_seed0"] = round(time.time() - t0, 1); gates["oracle_threads"] = threads
        np.savez_compressed(os.path.join(OUT, "frame_c3_800_seed0.npz"), image=a, rgb8=O.quantize_rgb8(a), seed=np.uint64(0),
                            n_coarse=64, n_fine=128, width=800, height=800)
        print("seed 0 frame written after %.0f s; white fraction %.4f" % (time.time() - t0, float((a == 1.0).all(axis=2).mean())), flush=True)
    else:
        a = np.load(os.path.join(committed, "frame_c3_800_seed0.npz"))["image"]
    if "seed1" in parts:
        b = O.render_image(co, fi, cam, O.make_opts(64, 128, seed=1, threads=threads))
        gates["cpu_seed1_vs_A"] = psnr(b, a)
        json.dump(gates, open(gates_path, "w"), indent=1)
        print("seed 1:", gates["cpu_seed1_vs_A"], flush=True)
    if "ssaa2" in parts:
        c = O.render_image(co, fi, cam, O.make_opts(64, 128, seed=1, ssaa=2, threads=threads))
        gates["cpu_ssaa2_seed1_vs_A"] = psnr(c, a)
        json.dump(gates, open(gates_path, "w"), indent=1)
        print("ssaa2 seed 1:", gates["cpu_ssaa2_seed1_vs_A"], "total %.0f s" % (time.time() - t0), flush=True)
    if "ppm512" in parts:
        import math
        raw = open(os.path.join(committed, "reference", "output.ppm"), "rb").read()
        magic, dims, maxv, data = raw.split(b"\n", 3)
        assert magic == b"P6" and dims == b"512 512" and maxv == b"255"
        ref = np.frombuffer(data, np.uint8).reshape(512, 512, 3).astype(np.float64)
        cam512 = O.camera_from_samples(S, 512, 512)
        cam512.alpha_width = cam512.alpha_height = math.pi / 8
        for seed in (0, 1):
            img = O.render_image(co, fi, cam512, O.make_opts(64, 128, seed=seed, threads=threads))
            q = O.quantize_rgb8(img).reshape(512, 512, 3).astype(np.float64)
            gates[f"cpu_pi8_512_seed{seed}_vs_output_ppm_psnr8"] = 10.0 * math.log10(255.0 ** 2 / float(np.mean((q - ref) ** 2)))
            json.dump(gates, open(gates_path, "w"), indent=1)
            print(f"512x512 pi/8 seed {seed} vs output.ppm:", gates[f"cpu_pi8_512_seed{seed}_vs_output_ppm_psnr8"], "total %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
