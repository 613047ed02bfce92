#!/usr/bin/env python3
"""Offline (CPU, numpy) study of hybrid_sampling's flag on the cases dumped by tools/dump_hybrid_cases.py: emulates k_resample's
quantities from the split arithmetic's densities and compares candidate per-draw displacement bounds with the measured per-draw
displacements (GPU, split-arithmetic draws vs f32 draws, same uniforms).
The kernel applies two rules on top of what is emulated here (both found later by tools/fuzz_hybrid_flags.py, neither changes the
statistics below by more than 0.1 %): a draw within 4 b of a bin edge is also tested against the neighbouring bin's width / mass,
and a density of 0 that the split kernels marked as uncertain (-0.0f) carries an error bound like a positive one.
Usage: fit_hybrid_model.py gpurun_out/hyb_cases.npz [arith]            (exploration)
       fit_hybrid_model.py gpurun_out/hyb_cases.npz arith eval eps_abs eps_rel eps_cap kappa   (flagged fraction, misses, calibration)"""
import sys

import numpy as np

f32 = np.float32
ROUND_T = 1.0


def philox_u(seed, pix, stream, n):
    """u01 of Philox-4x32-10, key = seed, counter = (pix, stream, k/4, 0) -- sampling_kernels.hip / oracle"""
    k = np.arange(n)
    c0 = np.repeat(pix.astype(np.uint64)[:, None], (n + 3) // 4, axis=1)
    c1 = np.full_like(c0, stream)
    c2 = np.broadcast_to((np.arange((n + 3) // 4)).astype(np.uint64)[None, :], c0.shape).copy()
    c3 = np.zeros_like(c0)
    k0 = np.uint64(seed & 0xFFFFFFFF); k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0; p1 = np.uint64(0xCD9E8D57) * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & M, p1 >> np.uint64(32), p1 & M
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & M; k1 = (k1 + np.uint64(0xBB67AE85)) & M
    x = np.stack([c0, c1, c2, c3], axis=2).reshape(pix.shape[0], -1)[:, :n]
    return ((x >> np.uint64(9)).astype(np.float32) * f32(1.0 / 8388608.0)).astype(f32)


def resample_quantities(t, s, far, dt=np.float32):
    """k_resample's arithmetic (reference order; f32 as in the kernel, or float64 for the math test) vectorised over rays"""
    t = t.astype(dt); s = s.astype(dt)
    R, nc = t.shape
    delta = np.concatenate([t[:, 1:] - t[:, :-1], (dt(far) - t[:, -1:])], axis=1).astype(dt)
    delta = np.maximum(delta, dt(0))
    alpha = (dt(1) - np.exp(-(s * delta).astype(dt)).astype(dt)).astype(dt)
    w = np.zeros_like(t); T = np.ones(R, dt); cut = np.zeros(R, bool); near = np.zeros(R, bool)
    Ts = np.zeros_like(t)
    for i in range(nc):
        Ts[:, i] = T
        w[:, i] = np.where(cut, dt(0), T * alpha[:, i])
        T = np.where(cut, T, (T * (dt(1) - alpha[:, i])).astype(dt))
        near |= np.abs(T - dt(1e-4)) < dt(1e-7)
        cut |= T < dt(1e-4)
    m = nc - 2
    adj = (np.maximum(w[:, 1:nc - 1], dt(0)) + dt(1e-5)).astype(dt)
    S = np.zeros(R, dt)
    for i in range(m):
        S = (S + adj[:, i]).astype(dt)
    pdf = (adj / S[:, None]).astype(dt)
    cdf = np.zeros((R, m + 1), dt)
    c = np.zeros(R, dt)
    for i in range(m):
        c = (c + pdf[:, i]).astype(dt)
        cdf[:, i + 1] = c
    cdf[:, m] = dt(1)
    bins = (dt(0.5) * (t[:, :-1] + t[:, 1:])).astype(dt)
    return dict(delta=delta, alpha=alpha, w=w, T=Ts, near=near, adj=adj, S=S, cdf=cdf, bins=bins, cutT=T)


def draws(q, u):
    cdf, bins = q["cdf"], q["bins"]
    R, m1 = cdf.shape; m = m1 - 1
    j = np.empty(u.shape, np.int64)
    for r in range(R):
        j[r] = np.clip(np.searchsorted(cdf[r], u[r], side="right") - 1, 0, m - 1)
    rr = np.arange(R)[:, None]
    cl, cu = cdf[rr, j], cdf[rr, j + 1]
    bl, bu = bins[rr, j], bins[rr, j + 1]
    den = np.maximum(cu - cl, f32(1e-6))
    return j, (bl + (bu - bl) * ((u - cl) / den)).astype(f32), (bu - bl), (cu - cl)


def main():
    Z = np.load(sys.argv[1])
    arith = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
    names = sorted({k.split("/")[0] for k in Z.files})
    for name in names:
        if f"{name}/s_{arith}" not in Z.files:
            continue
        nc, nf, seed = (int(v) for v in Z[name + "/meta"][:3])
        t, s32, s16 = Z[name + "/t"], Z[name + "/s32"], Z[f"{name}/s_{arith}"]
        mv = Z[f"{name}/move_{arith}"].astype(np.float32); flags = Z[f"{name}/flags_{arith}"].astype(bool)
        pix = Z[name + "/pix"]
        u = philox_u(seed, pix, 1, nf)
        q = resample_quantities(t, s16, 6.0)
        q32 = resample_quantities(t, s32, 6.0)
        j, tn, width, mass = draws(q, u)
        j32, tn32, _, _ = draws(q32, u)
        emu = np.abs(tn - tn32)
        ds = np.abs(s16 - s32)
        print(f"== {name} ({arith}): rays {t.shape[0]}, |ds| max {ds.max():.2e}, rel {np.max(ds / (1e-30 + np.maximum(np.abs(s32), 1e-3))):.2e}; "
              f"one zero / other not: {((s16 == 0) != (s32 == 0)).mean():.2e}, max sigma there {np.max(np.where((s16 == 0) != (s32 == 0), np.maximum(s16, s32), 0)):.2e}")
        big = mv > 1e-5
        print(f"   GPU movers (ray) {big.any(axis=1).mean():.4f}, draws moving {big.mean():.2e}; emulated movers (ray) {(emu > 1e-5).any(axis=1).mean():.4f}; "
              f"flagged {flags.mean():.4f}, near-cut {q['near'].mean():.4f}")
        rr = np.arange(t.shape[0])[:, None]
        cj = q["cdf"][rr, j]
        # where do moving draws sit?
        if big.any():
            print(f"   moving draws: cdf_j quantiles {np.quantile(cj[big], [0.01, 0.1, 0.5, 0.9])}, mass quantiles {np.quantile(mass[big], [0.1, 0.5, 0.9, 0.99])}, "
                  f"S quantiles {np.quantile(np.broadcast_to(q['S'][:, None], mv.shape)[big], [0.01, 0.1, 0.5])}")
            dcdf_needed = mv * np.maximum(mass, 1e-6) / np.maximum(width, 1e-9)  # the |dcdf| that explains each displacement
            print(f"   implied |dcdf| of moving draws: quantiles {np.quantile(dcdf_needed[big], [0.1, 0.5, 0.9, 0.99, 1.0])}")
            allq = np.quantile(dcdf_needed[mv > 0], [0.5, 0.9, 0.99, 0.999, 1.0]) if (mv > 0).any() else None
            print(f"   implied |dcdf| of all draws with a displacement: {allq}")


if __name__ == "__main__" and (len(sys.argv) <= 3 or sys.argv[3] != "eval"):
    main()


def model_bound(q, s, eps_a, eps_r, eps_cap, kappa, l2=0.0, e=None, round_t=None):
    """Per-bin-edge bound of |d cdf_j|.  The weights telescope: sum_{i<=j} w_i = 1 - T_(j+1), so with the interior samples 1..j in front
    of edge j, P_j = T_1 - T_(j+1) + j 1e-5, S = T_1 - T_end + m 1e-5, cdf_j = P_j / S and, to first order in d sigma,
        dT_i = -T_i sum_{k<i} delta_k dsigma_k
        d cdf_j = ((T_(j+1) - cdf_j T_end) X_j - cdf_j T_end (X_end - X_j) - (1 - cdf_j) T_1 X_0') / S,  X_j = sum_{k<=j} delta_k dsigma_k
    (samples behind the T < 1e-4 cut have no influence: T is frozen there).  |dsigma_k| <= e_k = min(eps_a + eps_r sigma_k, eps_cap) for
    sigma_k > 0, 0 for an exact zero.  l2 = 0: X bounded by the L1 sum; l2 > 0: by l2 x the root of the sum of squares (independent errors).
    Plus the rounding noise of the sequential f32 sums: kappa x 6e-8 x cdf_j."""
    delta, w, T, S, cdf = q["delta"], q["w"], q["T"], q["S"], q["cdf"]
    R, nc = s.shape
    if e is None:
        e = np.where(s > 0, np.minimum(eps_a + eps_r * s, eps_cap), 0.0)
    Tn = np.concatenate([T[:, 1:], q["cutT"][:, None]], axis=1)      # T_(i+1)
    frozen = np.concatenate([np.zeros((R, 1), bool), Tn[:, :-1] == Tn[:, 1:]], axis=1) & (Tn < 1e-4)  # behind the cut
    # T *= (1 - alpha) in f32: alpha is rounded to its own ulp, so the factor (1 - alpha) carries an ABSOLUTE error of up to ulp(alpha)
    # -- a relative error ulp(alpha) / (1 - alpha) of T, large when a sample is nearly opaque (found by tools/fuzz_hybrid_flags.py:
    # rays that start inside matter, T_1 ~ 1e-4 quantised to 6e-8)
    al = q["alpha"].astype(np.float64)
    rho = np.where(e > 0, 1.2e-7 * al / np.maximum(1 - al, 6e-8), 0.0) * (ROUND_T if round_t is None else round_t)
    de = np.where(frozen, 0.0, delta * e + rho)
    if l2 > 0:
        X = l2 * np.sqrt(np.cumsum(de * de, axis=1)); Xend = X[:, -1:]
        rest = l2 * np.sqrt(np.maximum(Xend ** 2 - X ** 2, 0)) / l2 * 1.0
        rest = np.sqrt(np.maximum((Xend / l2) ** 2 - (X / l2) ** 2, 0)) * l2
    else:
        X = np.cumsum(de, axis=1); Xend = X[:, -1:]; rest = Xend - X
    m = nc - 2
    Tend = Tn[:, m:m + 1]                                # behind the last INTERIOR sample: the ray's last sample is in no bin
    Xend = X[:, m:m + 1]
    rest = (np.sqrt(np.maximum((Xend / l2) ** 2 - (X / l2) ** 2, 0)) * l2) if l2 > 0 else np.maximum(Xend - X, 0)
    # edge j (0..m) sits behind interior samples 1..j: X_j, T_(j+1)
    Xj = X[:, 0:m + 1]; Tj1 = Tn[:, 0:m + 1]; restj = rest[:, 0:m + 1]
    X0 = de[:, 0:1]; T1 = Tn[:, 0:1]
    b = (np.abs(Tj1 - cdf * Tend) * Xj + cdf * Tend * restj + (1 - cdf) * T1 * X0) / S[:, None] + kappa * 6e-8 * cdf
    b[:, -1] = 0.0                                       # cdf[m] is forced to 1
    b[:, 0] = 0.0
    return b


def evaluate(path, arith, params, tau=1e-5, verbose=True):
    Z = np.load(path)
    names = sorted({k.split("/")[0] for k in Z.files})
    tot = []
    for name in names:
        if f"{name}/s_{arith}" not in Z.files:
            continue
        nc, nf, seed = (int(v) for v in Z[name + "/meta"][:3])
        t, s32, s16 = Z[name + "/t"], Z[name + "/s32"], Z[f"{name}/s_{arith}"]
        mv = Z[f"{name}/move_{arith}"].astype(np.float32); flags_old = Z[f"{name}/flags_{arith}"].astype(bool)
        u = philox_u(seed, Z[name + "/pix"], 1, nf)
        q = resample_quantities(t, s16, 6.0); q32 = resample_quantities(t, s32, 6.0)
        j, tn, width, mass = draws(q, u)
        b = model_bound(q, s16.astype(np.float64), *params)
        rr = np.arange(t.shape[0])[:, None]
        bj = np.maximum(b[rr, j], b[rr, j + 1])
        pred = width * bj / np.maximum(mass, 1e-30)
        flag = (pred > tau).any(axis=1) | q["near"]
        rmv = mv.max(axis=1)
        # calibration: actual |dcdf| (emulated) against the bound
        dc = np.abs(q["cdf"].astype(np.float64) - q32["cdf"])
        ratio = dc / np.maximum(b, 1e-30)
        ratio[:, 0] = 0; ratio[:, -1] = 0
        miss = (~flag) & (rmv > tau)
        if verbose:
            print(f"{name:12s} old flagged {flags_old.mean():.4f} new {flag.mean():.4f}  movers {(rmv > tau).mean():.4f}  missed {miss.sum():3d}  unflagged max {rmv[~flag].max() if (~flag).any() else 0:.2e}"
                  f"  |dcdf|/bound: p99 {np.quantile(ratio, 0.99):.2f} p99.99 {np.quantile(ratio, 0.9999):.2f} max {ratio.max():.2f}  (near {q['near'].mean():.4f})")
        tot.append((name, flags_old.mean(), flag.mean(), miss.sum(), rmv[~flag].max() if (~flag).any() else 0))
    return tot


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "eval":
    params = tuple(float(v) for v in sys.argv[4:9])
    evaluate(sys.argv[1], sys.argv[2], params)
