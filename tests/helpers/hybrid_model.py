"""numpy emulation of k_resample's arithmetic and of hybrid_sampling's per-bin-edge bound (sampling_kernels.hip) for
tests/test_hybrid_model_math.py.  (Round 3's offline fitting tools that used these functions were removed in round 4: hybrid_sampling is
frozen as a documented approximation.)"""
import numpy as np

f32 = np.float32
ROUND_T = 1.0

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
