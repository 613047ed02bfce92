"""Scene helpers shared by the GPU tests and the fuzz / dump tools: rotated camera poses of the lego JSON camera and random-weight
networks in the reference's directory format (src/lib.rs:108-174)."""
import numpy as np


def pose(samples, deg, tilt=0.0):
    """The JSON camera rotated about the scene's up axis (and optionally tilted about x)."""
    c2w = np.array(samples["camera_matrix"], np.float64)
    a, b = np.deg2rad(deg), np.deg2rad(tilt)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ \
        np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    return np.concatenate([R @ c2w[:, :3], (R @ c2w[:, 3])[:, None]], axis=1)


def oracle_samples(samples, m):
    return dict(samples, camera_origin=list(m[:, 3]), camera_forward=list(-m[:, 2]), camera_up=list(m[:, 1]))


def random_scene(root, seed, alpha_bias=(0.02, 0.03), alpha_scale=0.25):
    """Two random networks of the reference's architecture in its directory format (He-scaled weights; alpha bias > 0 so that
    the density field is alive: a fog of varying density instead of a surface -- very different weight statistics from lego)."""
    rng = np.random.default_rng(seed)
    shapes = [("dense0", 63, 256)] + [(f"dense{i}", 256, 256) for i in range(1, 5)] + [("dense5", 319, 256), ("dense6", 256, 256),
              ("dense7", 256, 256), ("bottleneck", 256, 256), ("viewdirs", 283, 128), ("rgb", 128, 3), ("alpha", 256, 1)]
    for which, a_bias in (("coarse", alpha_bias[0]), ("fine", alpha_bias[1])):
        d = root / which
        d.mkdir(parents=True)
        lines = []
        for name, k, n in shapes:
            w = (rng.normal(size=(k, n)) * np.sqrt(2.0 / k)).astype("<f4")
            b = (rng.normal(size=(n,)) * 0.1).astype("<f4")
            if name == "alpha":
                w *= alpha_scale
                b[:] = a_bias
            w.tofile(d / f"{name}_kernel.bin"); b.tofile(d / f"{name}_bias.bin")
            lines += [f"{name}_kernel {k} {n}", f"{name}_bias {n}"]
        (d / "shapes.txt").write_text("\n".join(lines) + "\n")
    return root
