#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/nerf_oracle.c).

Chain of trust: the oracle's MLP is pinned by the reference's own 120 golden scalars
(lego_rust/tf_reference_samples.json, tests/test_oracle_golden.py); camera / sampling / integration have no
known-answer test in the reference, so these fixtures are produced BY the oracle (a line-by-line restatement of
src/lib.rs:176-565) and pin (a) the oracle against regressions and (b) the GPU path on the GPU box, where
/root/reference does not exist.  Deterministic: counter-based RNG, seed recorded in each file.

    python tests/golden/make_golden.py     # rewrites tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

SCENE = os.path.join(ROOT, "lego_rust")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    S = O.load_samples(os.path.join(SCENE, "tf_reference_samples.json"))
    co, fi = O.Net(os.path.join(SCENE, "coarse")), O.Net(os.path.join(SCENE, "fine"))

    # (i) ray directions: corners + interior pixels at 400^2 and 800^2 (un-normalised and normalised)
    rd = {}
    for n in (400, 800):
        cam = O.camera_from_samples(S, n, n)
        pix = [(0, 0), (0, n - 1), (n - 1, 0), (n - 1, n - 1), (n // 2, n // 2), (n // 2 - 1, n // 2), (1, 2), (n - 2, 3),
               (n // 4, n // 3), (n // 3, 3 * n // 4), (17, n - 19), (n - 23, 29), (n // 2, 0), (0, n // 2), (n - 1, n // 2), (n // 2, n - 1)]
        rd[f"pix{n}"] = np.array(pix, np.int32)
        rd[f"dir{n}"] = np.stack([O.get_ray_dir(cam, i, j) for i, j in pix])
        rd[f"dirhat{n}"] = np.stack([O.normalize(O.get_ray_dir(cam, i, j)) for i, j in pix])
    np.savez_compressed(os.path.join(OUT, "ray_dirs.npz"), **rd)

    # (ii) per-ray stage dumps at the C3 geometry (800x800, 64 + 128), seed 0
    seed = 0
    cam = O.camera_from_samples(S, 800, 800)
    opts = O.make_opts(64, 128, seed=seed)
    pixels = [(i, j) for i in range(230, 590, 45) for j in range(250, 610, 90)]  # 8 x 4 = 32 rays through the model
    pixels[0] = (5, 5)  # guaranteed-empty ray (sigma == 0 everywhere -> pure white)
    keys = ["dir_hat", "t_coarse", "sigma_coarse", "w_coarse", "cdf", "u_fine", "t_new", "t_merged", "sigma_fine",
            "rgb_fine", "w_fine", "rgb"]
    dumps = [O.render_ray_debug(co, fi, cam, opts, i, j) for i, j in pixels]
    st = {k: np.stack([d[k] for d in dumps]) for k in keys}
    st["pixels"] = np.array(pixels, np.int32)
    st["pixel_index"] = np.array([i * 800 + j for i, j in pixels], np.uint32)
    st["seed"] = np.uint64(seed); st["near"] = np.float32(cam.near); st["far"] = np.float32(cam.far)
    st["origin"] = np.array(list(cam.pos), np.float32)
    terminated = [(d["w_fine"][-1] == 0) and (d["sigma_fine"] > 0).any() and
                  (np.cumprod(1 - (1 - np.exp(-d["sigma_fine"] * np.diff(np.append(d["t_merged"], cam.far))))).min() < 1e-4)
                  for d in dumps]
    empty = [(d["sigma_fine"] == 0).all() and (d["sigma_coarse"] == 0).all() for d in dumps]
    assert any(terminated), "fixture set must contain a ray cut at T < 1e-4"
    assert any(empty), "fixture set must contain an all-empty ray"
    assert np.allclose(st["rgb"][np.array(empty)], 1.0)
    st["is_terminated"] = np.array(terminated); st["is_empty"] = np.array(empty)
    # count == 0 fallback (src/lib.rs:295-297): fine net on the 64 coarse samples only
    opts0 = O.make_opts(64, 0, seed=seed)
    d0 = [O.render_ray_debug(co, fi, cam, opts0, i, j) for i, j in pixels[:8]]
    st["nofine_rgb"] = np.stack([d["rgb"] for d in d0])
    st["nofine_sigma_fine"] = np.stack([d["sigma_fine"] for d in d0])
    np.savez_compressed(os.path.join(OUT, "ray_stages_800.npz"), **st)
    print("stage fixtures:", sum(terminated), "terminated,", sum(empty), "empty of", len(pixels))

    # (iii) deterministic crop images
    cam400 = O.camera_from_samples(S, 400, 400)
    c1 = O.render_image(co, fi, cam400, O.make_opts(64, 0, coarse_only=True, crop=(150, 150, 100, 100), seed=0))
    np.savez_compressed(os.path.join(OUT, "crop_c1_400_coarse_only.npz"), image=c1, crop=np.array((150, 150, 100, 100)),
                        seed=np.uint64(0), n_coarse=64, n_fine=0, width=400, height=400)
    crop = (368, 352, 64, 64)
    c3 = O.render_image(co, fi, cam, O.make_opts(64, 128, crop=crop, seed=0))
    np.savez_compressed(os.path.join(OUT, "crop_c3_800_64_128.npz"), image=c3, crop=np.array(crop), seed=np.uint64(0),
                        n_coarse=64, n_fine=128, width=800, height=800)
    c3s1 = O.render_image(co, fi, cam, O.make_opts(64, 128, crop=crop, seed=1))
    np.savez_compressed(os.path.join(OUT, "crop_c3_800_64_128_seed1.npz"), image=c3s1, crop=np.array(crop),
                        seed=np.uint64(1), n_coarse=64, n_fine=128, width=800, height=800)
    # SSAA 2x of a 16x16 window at 400^2 output (= 32x32 rays of the 800^2 ray grid)
    cs = O.render_image(co, fi, cam400, O.make_opts(64, 128, crop=(192, 184, 16, 16), ssaa=2, seed=0))
    np.savez_compressed(os.path.join(OUT, "crop_ssaa2_400.npz"), image=cs, crop=np.array((192, 184, 16, 16)),
                        seed=np.uint64(0), n_coarse=64, n_fine=128, width=400, height=400, ssaa=2)
    mse = float(np.mean((np.clip(c3, 0, 1) - np.clip(c3s1, 0, 1)) ** 2))
    print("crops written; seed0-vs-seed1 PSNR on the C3 crop: %.2f dB" % (10 * np.log10(1.0 / mse)))

    # forward_batch vectors: 4096 random scene points (SURVEY 8d micro-bench distribution), both networks
    rng = np.random.default_rng(0)
    pts = rng.uniform(-2.2, 2.2, size=(3, 4096)).astype(np.float32)
    v = rng.normal(size=(4096, 3)); dirs = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    fb = {"pts": pts, "dirs": dirs}
    for name, net in (("coarse", co), ("fine", fi)):
        rgb, sig = net.forward_batch(pts, dirs)
        fb[f"{name}_rgb"], fb[f"{name}_sigma"] = rgb, sig
    np.savez_compressed(os.path.join(OUT, "forward_batch_4096.npz"), **fb)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
