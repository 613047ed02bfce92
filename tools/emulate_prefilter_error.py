#!/usr/bin/env python3
"""CPU emulation behind certify_zero's margin floors: how far is a 16-bit evaluation of the density pre-activation (weights and layer
inputs rounded to bf16 / f16, f32 accumulation, f32 alpha head -- what mlp_kernel_bf16v2.hip / mlp_kernel_f16v2.hip compute) from the f32 one, on
the samples of real lego rays (sample positions from the CPU oracle: coarse t and merged fine t of N random pixels of the 800 x 800 frame), and
what fraction of the samples lies within a margin of 0 (= has to be evaluated exactly whatever the pre-filter says).  No GPU.
    python tools/emulate_prefilter_error.py [rays=1500]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O

SCENE = os.path.join(ROOT, "lego_rust")


def load(net):
    W = {}
    for line in open(os.path.join(SCENE, net, "shapes.txt")):
        p = line.split()
        W[p[0]] = np.fromfile(os.path.join(SCENE, net, p[0] + ".bin"), np.float32).reshape([int(x) for x in p[1:]])
    return W


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32).view(np.float32)


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def enc(p, octaves):
    out, f = [p], 1.0
    for _ in range(octaves):
        out += [np.sin(p * np.float32(f)), np.cos(p * np.float32(f))]
        f *= 2
    return np.concatenate(out, 1).astype(np.float32)


def pre_activation(W, pts, rnd):
    e = enc(pts, 10); h = e
    for i in range(8):
        x = np.concatenate([e, h], 1) if i == 5 else h   # src/network.rs:209-210
        h = np.maximum(rnd(x) @ rnd(W[f"dense{i}_kernel"]) + W[f"dense{i}_bias"], 0).astype(np.float32)
    return (h @ W["alpha_kernel"] + W["alpha_bias"])[:, 0]   # the alpha head runs in f32 on the f32 accumulators


if __name__ == "__main__":
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    coarse, fine = O.Net(os.path.join(SCENE, "coarse")), O.Net(os.path.join(SCENE, "fine"))
    cam = O.camera_from_samples(O.load_samples(os.path.join(SCENE, "tf_reference_samples.json")), 800, 800)
    opts = O.make_opts(64, 128, seed=0)
    rng = np.random.default_rng(1)
    T = {"coarse": [], "fine": []}; D = []
    for _ in range(n_rays):
        d = O.render_ray_debug(coarse, fine, cam, opts, int(rng.integers(0, 800)), int(rng.integers(0, 800)))
        T["coarse"].append(d["t_coarse"]); T["fine"].append(d["t_merged"]); D.append(d["dir_hat"])
    o = np.array(list(cam.pos), np.float32); D = np.array(D, np.float32)
    for net in ("coarse", "fine"):
        t = np.array(T[net], np.float32)
        pts = (o[None, None, :] + D[:, None, :] * t[:, :, None]).reshape(-1, 3).astype(np.float32)
        W = load(net)
        exact = pre_activation(W, pts, lambda x: x)
        zeros = exact <= 0
        print(f"{net}: {len(exact)} samples, {zeros.mean():.3f} of them zeros of the f32 network")
        for name, rnd in (("bf16", bf16), ("f16", f16)):
            err = np.abs(pre_activation(W, pts, rnd) - exact)
            q = np.quantile(err[zeros], [0.5, 0.99, 0.999, 1.0])
            print(f"  {name}: |16-bit - f32| on the zeros: median {q[0]:.3g}, p99 {q[1]:.3g}, p99.9 {q[2]:.3g}, max {q[3]:.3g}; on all samples max {err.max():.3g}")
        print("  samples with f32 pre-activation > -margin (exact evaluation whatever the pre-filter says): " +
              ", ".join(f"{m}: {(exact > -m).mean():.4f}" for m in (0.1, 0.25, 0.5, 1.0, 2.0, 3.0)) + f"; live: {(exact > 0).mean():.4f}")
