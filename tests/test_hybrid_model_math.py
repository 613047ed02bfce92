"""The mathematics behind hybrid_sampling's flag (sampling_kernels.hip k_resample), on the CPU: the telescoped first-order
bound of |d cdf_j| per bin edge must dominate what a density perturbation within its per-sample limits actually does to the CDF of
sample_importance (src/lib.rs:289-351) -- checked by brute force in float64 (no rounding in the way) on random rays: surfaces, fogs,
rays cut at T < 1e-4, rays whose surface sits in their last samples (the case the first version of the model got wrong)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import hybrid_model as F


def _rays(rng, R, nc):
    edges = np.linspace(2.0, 6.0, nc + 1)
    t = edges[:-1] + (edges[1:] - edges[:-1]) * rng.uniform(size=(R, nc))
    s = np.zeros((R, nc))
    kind = rng.integers(0, 4, size=R)
    for r in range(R):
        if kind[r] == 0:      # an opaque surface somewhere, soft leading edge
            k = rng.integers(1, nc)
            s[r, k:] = rng.uniform(5, 150, size=nc - k)
            s[r, max(k - 2, 0):k] = rng.uniform(0, 2, size=k - max(k - 2, 0))
        elif kind[r] == 1:    # a thin fog
            s[r] = rng.uniform(0, 0.5, size=nc) * (rng.uniform(size=nc) < 0.5)
        elif kind[r] == 2:    # surface in the last one to three samples
            k = nc - rng.integers(1, 4)
            s[r, k:] = rng.uniform(1, 30, size=nc - k)
        else:                 # a few isolated blobs
            for k in rng.integers(0, nc, size=3):
                s[r, k] = rng.uniform(0.1, 8)
    return t, s


def test_first_order_bound_dominates_brute_force_perturbations():
    rng = np.random.default_rng(5)
    for nc in (64, 32, 20):
        t, s = _rays(rng, 400, nc)
        q = F.resample_quantities(t, s, 6.0, dt=np.float64)
        e = np.where(s > 0, 1e-5 + 2e-6 * s, 0.0)            # per-sample limits: far above float64 rounding, small enough for first order
        b = F.model_bound(q, s, 0, 0, 0, 0.0, e=e, round_t=0.0)
        worst = 0.0
        for trial in range(12):
            sign = rng.choice([-1.0, 1.0], size=s.shape) if trial else np.ones_like(s)
            mag = rng.uniform(0, 1, size=s.shape) if trial > 2 else 1.0
            q2 = F.resample_quantities(t, np.maximum(s + sign * mag * e * (1 if trial != 1 else -1), 0), 6.0, dt=np.float64)
            same_cut = (q2["w"] > 0).sum(axis=1) == (q["w"] > 0).sum(axis=1)   # a cut that flips is the near-cut rule's business
            d = np.abs(q2["cdf"] - q["cdf"])[same_cut]
            d = np.where(d < 1e-11, 0.0, d)                   # float64 rounding of the sums themselves (the flag cares about >= 1e-9)
            ratio = np.where(d > 0, d / np.maximum(b[same_cut], 1e-300), 0.0)
            worst = max(worst, float(ratio.max()))
        assert worst <= 1.01, (nc, worst)                     # first order: 1 % for the second-order terms
        # and it is not vacuous: with all errors at their limits and one sign the bound is attained within a factor of a few somewhere
        q3 = F.resample_quantities(t, s + e, 6.0, dt=np.float64)
        att = (np.abs(q3["cdf"] - q["cdf"]) / np.maximum(b, 1e-300))[:, 1:-1]
        assert np.nanmax(att) > 0.5


def test_emulation_matches_the_kernels_documented_constants():
    """The constants quoted in DESIGN 4.8 / sampling_kernels.hip are the ones the offline evaluation used."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-rs_amd", "csrc", "sampling_kernels.hip")).read()
    assert "kEpsAbs = 2e-5f, kEpsRel = 6e-6f, kEpsCap = 2e-4f, kRound = 6.0f * 5.9604645e-8f" in src
    assert "1.2e-7f * al / fmaxf(1.0f - al, 6e-8f)" in src
