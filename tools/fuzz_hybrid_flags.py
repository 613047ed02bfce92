#!/usr/bin/env python3
"""Fuzz of hybrid_sampling's documented bound ("a ray that is not flagged places every draw within 1e-5 in t of the f32 placement",
include/nerf_mi355x.h): random poses, frame sizes, windows, sample counts and seeds of the lego scene; per case the coarse densities in
f32 and in a split arithmetic, nerf_stage_hybrid_flags on the latter, and the per-draw displacement between the two sets of draws
(same uniforms).  Reports rays, flagged fraction, real movers, MISSES (unflagged rays with a draw beyond 1e-5) and the largest unflagged
displacement.  The bound rests on a statistical model of the density error: expect about one miss per million rays, all below 3e-5.
NERF_FUZZ_SCENE=<dir with coarse/ and fine/> fuzzes another network with the lego camera path (the flag's constants were measured on
the lego networks: run this before enabling hybrid_sampling for a different scene).
Usage: fuzz_hybrid_flags.py [seconds] [rng seed] [seconds of the image-level fuzz]   (exit code 1 if a miss exceeds 5e-5 or the miss rate exceeds 5 per million;
tests/test_gpu_hybrid_validation.py runs a short one)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerf_rs_amd as N
from scene_utils import pose as _pose

S = json.load(open(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json")))
FAR = float(S["far"])


def fuzz(r, budget, rng_seed, missed=None):
    """r: a Renderer with the lego scene loaded.  Returns the totals."""
    rng = np.random.default_rng(rng_seed)
    tot = dict(cases=0, rays=0, flagged=0, movers=0, misses=0, worst=0.0)
    hist = np.zeros(5, np.int64)  # unflagged rays by largest displacement: >1e-6, >3e-6, >5e-6, >7e-6, >1e-5
    t_end = time.time() + budget
    while time.time() < t_end:
        W = int(rng.choice([64, 96, 128, 200, 256, 400, 800]))
        nc, nf = [(64, 128), (64, 128), (48, 96), (32, 64), (20, 50), (64, 64), (33, 77)][int(rng.integers(7))]
        deg, tilt = float(rng.uniform(0, 360)), float(rng.uniform(-25, 25))
        cam = N.camera_from_pose(_pose(S, deg, tilt), S["hwf"], S["near"], S["far"], W, W, nc)
        w, h = int(min(W, rng.integers(16, 161))), int(min(W, rng.integers(16, 129)))
        x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, W - h + 1))
        seed = int(rng.integers(0, 1 << 30))
        t = r.stage_stratified(cam, x0, y0, w, h, nc, seed=seed).reshape(-1, nc)
        dirs = r.stage_ray_dirs(cam, x0, y0, w, h).reshape(-1, 3)
        o = cam.pos.astype(np.float32)
        pts = (o[None, None, :] + dirs[:, None, :] * t[:, :, None]).astype(np.float32).reshape(-1, 3)
        dd = np.repeat(dirs, nc, axis=0)
        pix = ((y0 + np.arange(h))[:, None] * W + (x0 + np.arange(w))[None, :]).reshape(-1).astype(np.uint32)
        soa = np.ascontiguousarray(pts.T)
        _, s32 = r.coarse.forward_batch(soa, dd, dtype="f32")
        ref = r.stage_resample(t, s32.reshape(-1, nc), nf, FAR, seed=seed, pixel_index=pix)["t_new"]
        for dt in ("f16x2", "bf16x3"):
            _, ssp = r.coarse.forward_batch(soa, dd, dtype=dt)
            flags, tn = r.stage_hybrid_flags(t, ssp.reshape(-1, nc), nf, FAR, seed=seed, pixel_index=pix)
            move = np.abs(tn - ref).max(axis=1)
            un = move[~flags]
            tot["rays"] += move.size; tot["flagged"] += int(flags.sum()); tot["movers"] += int((move > 1e-5).sum())
            miss = int((un > 1e-5).sum())
            tot["misses"] += miss
            if un.size:
                tot["worst"] = max(tot["worst"], float(un.max()))
                hist += np.array([(un > v).sum() for v in (1e-6, 3e-6, 5e-6, 7e-6, 1e-5)])
            if miss:
                idx = np.nonzero(~flags & (move > 1e-5))[0]
                for i in idx[:4]:
                    if missed is not None and len(missed) < 400:
                        missed.append(dict(nc=nc, nf=nf, seed=seed, pix=int(pix[i]), dt=dt, t=t[i].copy(), s32=s32.reshape(-1, nc)[i].copy(),
                                           ssp=ssp.reshape(-1, nc)[i].copy(), tn=tn[i].copy(), ref=ref[i].copy()))
                print(f"MISS: W {W} pose {deg:.1f}/{tilt:.1f} window {x0},{y0},{w},{h} {nc}+{nf} seed {seed} {dt}: {miss} rays, max {un.max():.3e}", flush=True)
        tot["cases"] += 1
    return dict(tot, flagged_fraction=tot["flagged"] / max(tot["rays"], 1), mover_fraction=tot["movers"] / max(tot["rays"], 1),
                unflagged_rays_beyond={"1e-6": int(hist[0]), "3e-6": int(hist[1]), "5e-6": int(hist[2]), "7e-6": int(hist[3]), "1e-5": int(hist[4])})


def fuzz_frames(r, budget, rng_seed):
    """The same at image level: random small frames rendered with exact-f32 sampling (skip_dead) and with hybrid_sampling in every
    arithmetic; the hybrid frame must be the f32-sampling frame of the same arithmetic within Gate 1's tolerances (max 5e-4, mean 1e-5)."""
    rng = np.random.default_rng(rng_seed)
    tot = dict(frames=0, rays=0, worst_max=0.0, worst_mean=0.0, over_1e4=0, over_5e4=0)
    t_end = time.time() + budget
    while time.time() < t_end:
        W = int(rng.choice([32, 48, 64, 96, 128]))
        nc, nf = [(64, 128), (64, 128), (48, 96), (32, 64), (20, 50), (64, 64), (33, 77)][int(rng.integers(7))]
        deg, tilt = float(rng.uniform(0, 360)), float(rng.uniform(-25, 25))
        cam = N.camera_from_pose(_pose(S, deg, tilt), S["hwf"], S["near"], S["far"], W, W, nc)
        seed = int(rng.integers(0, 1 << 30))
        for dt in ("f16x2", "bf16x3", "f32"):
            ref = N.render_image(r.coarse, r.fine, cam, nf, seed=seed, dtype=dt, skip_dead=True)
            img = N.render_image(r.coarse, r.fine, cam, nf, seed=seed, dtype=dt, skip_dead=True, hybrid_sampling=True)
            d = np.abs(img - ref)
            tot["worst_max"] = max(tot["worst_max"], float(d.max())); tot["worst_mean"] = max(tot["worst_mean"], float(d.mean()))
            tot["over_1e4"] += int((d.max(axis=2) > 1e-4).sum()); tot["over_5e4"] += int((d.max(axis=2) > 5e-4).sum())
            tot["rays"] += W * W
            if d.max() > 5e-4:
                print(f"FRAME: W {W} pose {deg:.1f}/{tilt:.1f} {nc}+{nf} seed {seed} {dt}: max {d.max():.3e} mean {d.mean():.3e}", flush=True)
        tot["frames"] += 1
    return tot


def acceptable(res):
    return res["worst"] <= 5e-5 and res["misses"] <= max(3, 5e-6 * res["rays"])


if __name__ == "__main__":
    scene = os.environ.get("NERF_FUZZ_SCENE", os.path.join(ROOT, "lego_rust"))  # another network in the reference's directory format
    missed = []  # the missed rays themselves, for offline analysis
    with N.Renderer(0) as r:
        r.load_scene(scene)
        res = fuzz(r, float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, missed)
        print(json.dumps(res))
        if len(sys.argv) > 3:  # third argument: seconds of the image-level fuzz
            print(json.dumps(fuzz_frames(r, float(sys.argv[3]), int(sys.argv[2]))))
    if missed:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", "hyb_fuzz_missed.npy"), np.array(missed, dtype=object), allow_pickle=True)
    sys.exit(0 if acceptable(res) else 1)
