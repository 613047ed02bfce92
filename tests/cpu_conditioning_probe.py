#!/usr/bin/env python3
"""How far apart are TWO CPU evaluations of the reference's algorithm?  The oracle (oracle/nerf_oracle.c, separate multiply and add as
rustc compiles src/network.rs:124-147) against the SAME source compiled with fused multiply-adds (oracle/libnerf_oracle_fma.so: what
f32::mul_add or a BLAS sgemm -- the reference's own macOS Accelerate path, build.rs -- computes): random views as in
tests/gpu_fuzz_vs_oracle.py, same seeds.  Both are f32 evaluations of the same network with differences of ~1e-6 relative in the
densities; hierarchical sampling is ill-conditioned in places, so a few pixels per million differ by far more than the
typical 1e-6 -- in CPU-vs-CPU exactly as in GPU-vs-CPU.  CPU only, test infrastructure.
Usage: python tests/cpu_conditioning_probe.py [seconds] [rng seed]"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O


def _second_oracle(lib_name):
    spec = importlib.util.spec_from_file_location("oracle_py_" + lib_name, os.path.join(ROOT, "oracle", "oracle_py.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m._LIB_PATH = os.path.join(ROOT, "oracle", lib_name)
    assert os.path.exists(m._LIB_PATH), "make -C oracle " + lib_name
    return m


def _pose(samples, deg, tilt):
    c2w = np.array(samples["camera_matrix"], np.float64)
    a, b = np.deg2rad(deg), np.deg2rad(tilt)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ \
        np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    m = np.concatenate([R @ c2w[:, :3], (R @ c2w[:, 3])[:, None]], axis=1)
    return dict(samples, camera_origin=list(m[:, 3]), camera_forward=list(-m[:, 2]), camera_up=list(m[:, 1]))


def probe(budget, rng_seed):
    O.build()
    F = _second_oracle("libnerf_oracle_fma.so")
    S = O.load_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"))
    nets = [(m.Net(os.path.join(ROOT, "lego_rust", "coarse")), m.Net(os.path.join(ROOT, "lego_rust", "fine"))) for m in (O, F)]
    rng = np.random.default_rng(rng_seed)
    tot = dict(frames=0, rays=0, over_5e4=0, over_1e4=0, worst_max=0.0, worst_mean=0.0, frames_failing_gate1=0)
    t_end = time.time() + budget
    while time.time() < t_end:
        W, H = int(rng.choice([24, 32, 40, 48])), int(rng.choice([16, 24, 32]))
        nc, nf = [(64, 128), (48, 96), (32, 64), (20, 50), (64, 64), (33, 77), (16, 0)][int(rng.integers(7))]
        deg, tilt = float(rng.uniform(0, 360)), float(rng.uniform(-25, 25))
        seed = int(rng.integers(0, 1 << 30))
        sm = _pose(S, deg, tilt)
        imgs = [m.render_image(*n, m.camera_from_samples(sm, W, H), m.make_opts(nc, nf, coarse_only=(nf == 0), seed=seed)) for m, n in zip((O, F), nets)]
        d = np.abs(imgs[0] - imgs[1])
        tot["frames"] += 1; tot["rays"] += W * H
        tot["over_5e4"] += int((d.max(axis=2) > 5e-4).sum()); tot["over_1e4"] += int((d.max(axis=2) > 1e-4).sum())
        tot["worst_max"] = max(tot["worst_max"], float(d.max())); tot["worst_mean"] = max(tot["worst_mean"], float(d.mean()))
        if tot["frames"] % 40 == 0:
            print(f"... {tot['frames']} frames, {tot['rays']} rays, {tot['over_5e4']} pixels over 5e-4", flush=True)
        if d.max() > 5e-4 or d.mean() > 1e-5:
            tot["frames_failing_gate1"] += 1
            print(f"CPU vs CPU(fma): {W}x{H} pose {deg:.2f}/{tilt:.2f} {nc}+{nf} seed {seed}: max {d.max():.3e} mean {d.mean():.3e}", flush=True)
    return tot


if __name__ == "__main__":
    print(json.dumps(probe(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
