// sampling_kernels.hip -- ray generation, stratified + hierarchical sampling, transmittance integration.
//
// gfx950 restatement of the non-MLP part of render_block (reference src/lib.rs:353-472):
//   k_ray_dirs         Camera::get_ray_dir + normalize            (src/lib.rs:213-231, 367-373; src/vec3.rs:27-34)
//   k_stratified       stratified_samples                          (src/lib.rs:233-248)
//   k_resample         compute_weights + sample_importance + sort  (src/lib.rs:250-351, 414-420)
//   k_composite        integrate_ray                               (src/lib.rs:176-195)
//   k_box_downsample   SSAA box filter (extension)
// All arithmetic is IEEE f32 without contraction (the file is built with -ffp-contract=off and HIP's default
// correctly-rounded fp32 divide/sqrt), so ray directions and coarse sample positions are BIT-IDENTICAL to the
// CPU path; the sequential f32 sums of the reference (pdf sum, cdf, transmittance, colour accumulation) are kept
// in the reference's order.  These kernels are HBM/latency-bound bookkeeping (<1 % of an f32 frame): resampling runs one
// wave per ray (lanes = samples), compositing one ray per lane with LDS-staged samples.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sampling_kernels.h"

namespace {

// Philox-4x32-10, key = seed, counter = (pixel_index, stream, k/4, 0); same stream as the CPU oracle.
__device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                           uint32_t c3, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 9) * (1.0f / 8388608.0f); }

// Intra-wave LDS hand-off: LDS instructions of one wave execute in issue order, so a compiler-level fence
// (no instruction at wavefront scope) plus a scheduling barrier is all that is needed.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

} // namespace

// Row of the (rny x rnx) ray grid that holds ray r of the pass: the pass's rows are consecutive rows of the window, or -- when the window's
// rows are dealt out in stripes over several bands (nerf_render_opts.band_*; multi-GPU partitions that balance when the cost per ray is
// not uniform) -- consecutive rows of THIS band: band-local row j is row stripe_y0 + ((j / stripe) * stripe_n + stripe_i) * stripe + j % stripe.
__device__ __forceinline__ int ray_row(const RayGenArgs &a, int r) {
    const int j = a.ry0 + r / a.rw;
    return a.stripe > 0 ? a.stripe_y0 + ((j / a.stripe) * a.stripe_n + a.stripe_i) * a.stripe + j % a.stripe : j;
}

// ---- ray directions: one thread per ray of the pass rectangle ---------------------------------------
__global__ void k_ray_dirs(RayGenArgs a, float *__restrict__ dirs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rays) return;
    const int i = ray_row(a, r), j = a.rx0 + r % a.rw;
    const float x = (((float)j + a.half) / (float)a.rnx) * 2.0f - 1.0f;
    const float y = 1.0f - (((float)i + a.half) / (float)a.rny) * 2.0f;
    const float xs = x * a.sx, ys = y * a.sy;
    const float dx = (a.r[0] * xs + a.u[0] * ys) + a.f[0];
    const float dy = (a.r[1] * xs + a.u[1] * ys) + a.f[1];
    const float dz = (a.r[2] * xs + a.u[2] * ys) + a.f[2];
    if (a.normalize) {
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dirs[3 * (size_t)r + 0] = dx / len; dirs[3 * (size_t)r + 1] = dy / len; dirs[3 * (size_t)r + 2] = dz / len;
    } else {
        dirs[3 * (size_t)r + 0] = dx; dirs[3 * (size_t)r + 1] = dy; dirs[3 * (size_t)r + 2] = dz;
    }
}

// ---- stratified samples: one thread per (ray, group of 4 samples) -----------------------------------
__global__ void k_stratified(RayGenArgs a, int count, float near_, float far_, uint32_t seed_lo, uint32_t seed_hi,
                             float *__restrict__ t) {
    const int quads = (count + 3) >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)a.n_rays * quads) return;
    const int r = (int)(gid / quads), q = (int)(gid % quads);
    const uint32_t pix = (uint32_t)(ray_row(a, r) * a.rnx + (a.rx0 + r % a.rw));
    uint32_t rnd[4];
    philox4x32(seed_lo, seed_hi, pix, 0u, (uint32_t)q, 0u, rnd);
    const float interval = (far_ - near_) / (float)count;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        if (k < count) {
            const float lower = near_ + (float)k * interval;
            const float upper = lower + interval;
            t[(size_t)r * count + k] = lower + (upper - lower) * u01(rnd[e]);
        }
    }
}

// ---- transmittance weights, shared by resample and composite ----------------------------------------
// alpha[] in LDS -> w[] in LDS, sequential in sample order exactly as compute_weights (src/lib.rs:261-280).
// Executed redundantly by every lane of the wave (wave-uniform control flow, LDS broadcast reads).
// Returns true if, before the cut, the transmittance passed within 0.1 % of the 1e-4 threshold: there a 1e-5 relative density
// difference can flip the cut by one sample (hybrid sampling redoes such rays in f32; nobody else looks at the result).
// `Tn` (optional, LDS): Tn[i] = transmittance BEHIND sample i (frozen once the ray is cut) -- hybrid sampling's error model reads it.
__device__ __forceinline__ bool weights_scan(const float *alpha, float *w, int n, int lane, float *Tn = nullptr) {
    // branch-free form of the early break (src/lib.rs:273-279): once T < 1e-4 every later weight is 0 and T is not touched
    // again -- identical values, but the loop has no loop-carried branch, so the LDS reads pipeline
    float T = 1.0f;
    bool cut = false, near = false;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
        const float al = alpha[i];
        const float wi = cut ? 0.0f : T * al;
        if (lane == 0) w[i] = wi;
        T = cut ? T : T * (1.0f - al);
        if (Tn && lane == 0) Tn[i] = T;
        near = near || fabsf(T - 1e-4f) < 1e-7f;
        cut = cut || T < 1e-4f;
    }
    return near;
}

__device__ __forceinline__ float sample_alpha(const float *t, const float *sigma, int i, int n, float far_) {
    float delta = (i + 1 < n) ? t[i + 1] - t[i] : far_ - t[i];
    if (delta < 0.0f) delta = 0.0f;
    return 1.0f - expf(-sigma[i] * delta);
}

// ---- hierarchical resampling: one wave per ray --------------------------------------------------------
// LDS per wave (floats): t[nc] sigma[nc] alpha[nc] w[nc] cdf[nc] bins[nc] Tn[nc] merged[pow2 >= nc+nf]
__global__ __launch_bounds__(256) void k_resample(ResampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int ray = blockIdx.x * 4 + wv;
    if (ray >= a.n_rays) return; // whole wave exits together; no block-level barrier below
    if (a.ray_list) {            // second launch of the hybrid sampling pass: only the listed rays
        if ((unsigned)ray >= *a.ray_list_count) return;
        ray = (int)a.ray_list[ray];
    }
    const int nc = a.nc, nf = a.nf, M = nc + nf;
    float *t = lds_f + (size_t)wv * (7 * nc + a.sort_pow2);
    float *sg = t + nc, *alpha = sg + nc, *w = alpha + nc, *cdf = w + nc, *bins = cdf + nc, *Tn = bins + nc, *mg = Tn + nc;
    const bool flagging = a.flag_list || a.flag_out; // hybrid sampling's first launch (and its stage hook): wave-uniform

    for (int i = lane; i < nc; i += 64) { t[i] = a.t_coarse[(size_t)ray * nc + i]; sg[i] = a.sigma_coarse[(size_t)ray * nc + i]; }
    wave_sync();
    for (int i = lane; i < nc; i += 64) alpha[i] = sample_alpha(t, sg, i, nc, a.far_);
    wave_sync();
    const bool near_cut = weights_scan(alpha, w, nc, lane, flagging ? Tn : nullptr);
    wave_sync();
    if (a.w_out) for (int i = lane; i < nc; i += 64) a.w_out[(size_t)ray * nc + i] = w[i];

    // sample_importance (src/lib.rs:289-351); nc >= 3 and nf > 0 guaranteed by the host
    const int m = nc - 2;
    for (int i = lane; i < nc - 1; i += 64) bins[i] = 0.5f * (t[i] + t[i + 1]);
    for (int i = lane; i < m; i += 64) {
        const float x = w[i + 1];
        alpha[i] = (x > 0.0f ? x : 0.0f) + 1e-5f; // adjusted
    }
    wave_sync();
    float sum = 0.0f;
    for (int i = 0; i < m; ++i) sum += alpha[i];                 // iter().sum(), sequential
    wave_sync();
    for (int i = lane; i < m; i += 64) alpha[i] = alpha[i] / sum;
    wave_sync();
    float cumulative = 0.0f;
    for (int i = 0; i < m; ++i) { cumulative += alpha[i]; if (lane == 0) cdf[i + 1] = cumulative; }
    if (lane == 0) { cdf[0] = 0.0f; cdf[m] = 1.0f; }             // :320, :326-328
    wave_sync();
    if (a.cdf_out) for (int i = lane; i <= m; i += 64) a.cdf_out[(size_t)ray * (nc - 1) + i] = cdf[i];

    const uint32_t pix = a.pixel_index ? a.pixel_index[ray]
                                       : (uint32_t)(ray_row(a.g, ray) * a.g.rnx + (a.g.rx0 + ray % a.g.rw));
    // Hybrid sampling: a bound of |d cdf_j| at every bin edge j if this ray's densities carry the split arithmetics' error against the
    // f32 kernel (fitted and checked offline on dumped cases: tools/dump_hybrid_cases.py, tools/fit_hybrid_model.py).
    // The weights telescope -- sum_{i<=j} w_i = 1 - T_(j+1) -- so with the interior samples 1..j in front of edge j
    //     cdf_j = (T_1 - T_(j+1) + j 1e-5) / S,   S = T_1 - T_end + m 1e-5 (end = m + 1),   dT_i = -T_i sum_{k<i} delta_k dsigma_k   (first order)
    //     d cdf_j = ((T_(j+1) - cdf_j T_end) X_j - cdf_j T_end (X_end - X_j) - (1 - cdf_j) T_1 X_0) / S,   X_j = sum_{k<=j} delta_k dsigma_k
    // (a sample behind the T < 1e-4 cut has no influence: T is frozen there).  With |dsigma_k| <= e_k every X is bounded by the running
    // sum of delta_k e_k (L1: no assumption on the signs; each term also carries the quantisation of T, see below).  e_k = min(kEpsAbs + kEpsRel sigma_k, kEpsCap) for sigma_k > 0 -- the measured
    // f16x2 / bf16x3-vs-f32 density differences of the lego networks (between their p99 and their maximum at every magnitude; an exact
    // zero is exact in every arithmetic unless its pre-activation is within that error of 0: the split kernels mark those as -0.0f).
    // On top, the sequential f32 sums of the CDF round differently as soon as ANY input differs: kRound ulps of 1.0 relative to cdf_j
    // (measured: <= 3 between the arithmetics).  A draw in bin [j, j+1) moves by at most width x max(b_j, b_j+1) / mass; above flag_tau the ray is flagged.
    // Against round 2's one |dCDF| per ray: 0.6 x as many flagged rays on the lego views, and over 17 dumped cases (112 000 rays, 5 of
    // them not used for the fit) no unflagged draw moves by more than 4.8e-6 (round 2's model: 8.1e-6 on the same rays).
    constexpr float kEpsAbs = 2e-5f, kEpsRel = 6e-6f, kEpsCap = 2e-4f, kRound = 6.0f * 5.9604645e-8f;
    float *bnd = mg; // edge bounds b_0..b_m (mg[0..nc) is free until the coarse samples are copied in below)
    if (flagging) {  // wave-uniform
        float carry = 0.0f;
        for (int base = 0; base < nc; base += 64) { // inclusive running sum of delta_k e_k (any order: it is a bound)
            const int i = base + lane;
            float v = 0.0f;
            if (i < nc) {
                float delta = (i + 1 < nc) ? t[i + 1] - t[i] : a.far_ - t[i];
                if (delta < 0.0f) delta = 0.0f;
                const float s_i = sg[i];
                const bool behind_cut = i > 0 && Tn[i - 1] < 1e-4f;
                // ... plus the quantisation of T: `T *= 1 - alpha` multiplies by a factor whose ABSOLUTE error is ulp(alpha) (alpha is
                // rounded on its own grid), a relative error ulp(alpha) / (1 - alpha) of every later T -- 6e-4 behind a sample that
                // leaves T ~ 1e-4 (rays that start inside matter: 0.1 per million rays of the fuzz moved by 3e-4 before this term)
                const float al = 1.0f - expf(-s_i * delta);
                // an exact zero carries no error -- unless the split kernels marked it as uncertain (-0.0f: its pre-activation is within
                // their own error of 0, mlp_split_kernels.hip.h alpha_head)
                const bool uncertain_zero = __float_as_uint(s_i) == 0x80000000u;
                v = ((s_i > 0.0f || uncertain_zero) && !behind_cut) ? delta * fminf(kEpsAbs + kEpsRel * s_i, kEpsCap) + 1.2e-7f * al / fmaxf(1.0f - al, 6e-8f) : 0.0f;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
            v += carry;
            if (i < nc) bnd[i] = v;
            carry = __shfl(v, 63, 64);
        }
        wave_sync();
        // "end" = the last INTERIOR sample m = nc - 2: the last sample of the ray is in no bin, so neither S nor any cdf_j sees it
        const float x_end = bnd[m], x_0 = bnd[0], t_end = Tn[m], t_1 = Tn[0];
        wave_sync();
        for (int j = lane; j <= m; j += 64) { // X_j is replaced by b_j in place (a lane reads only its own entry)
            const float c = cdf[j], xj = bnd[j];
            const float b = (fabsf(Tn[j] - c * t_end) * xj + c * t_end * (x_end - xj) + (1.0f - c) * t_1 * x_0) / sum + kRound * c;
            bnd[j] = (j == 0 || j == m) ? 0.0f : b; // cdf[0] = 0 and cdf[m] = 1 are constants
        }
        wave_sync();
    }
    bool light = false;
    for (int s = lane; s < nf; s += 64) {
        float u;
        if (a.u_in) u = a.u_in[(size_t)ray * nf + s];
        else { uint32_t rnd[4]; philox4x32(a.seed_lo, a.seed_hi, pix, 1u, (uint32_t)(s >> 2), 0u, rnd); u = u01(rnd[s & 3]); }
        // first j with cdf[j] <= u < cdf[j+1] == largest j in [0,m-1] with cdf[j] <= u for a non-decreasing cdf
        int lo = 0, hi = m - 1;
        if (!(u >= cdf[0] && u < cdf[m])) lo = hi;                // no match: idx = adjusted.len() - 1 (:333)
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cdf[mid] <= u) lo = mid; else hi = mid - 1; }
        const float cl = cdf[lo], cu = cdf[lo + 1];
        float denom = cu - cl;
        const float bl = bins[lo], bu = bins[lo + 1];
        if (flagging) {
            const float b_lo = bnd[lo], b_hi = bnd[lo + 1];
            light = light || !((bu - bl) * fmaxf(b_lo, b_hi) <= a.flag_tau * denom);
            // A draw close to an edge may land in the NEIGHBOURING bin under the other arithmetic: if that bin is light, the part of
            // the edge's shift spent in it is stretched by its width / mass (tools/fuzz_hybrid_flags.py: 0.7 rays per million sat next
            // to an empty bin and moved by up to 3.6e-2).  "Close" = within four times the edge's bound: one density error decides here
            // and a single error reaches 3 x its e_k (a 255 M-ray fuzz found a draw 3.1 b from an edge, 2.8e-2 away afterwards).
            if (lo > 0 && u - cl <= 4.0f * b_lo) light = light || !((bl - bins[lo - 1]) * b_lo <= a.flag_tau * (cl - cdf[lo - 1]));
            if (lo + 1 < m && cu - u <= 4.0f * b_hi) light = light || !((bins[lo + 2] - bu) * b_hi <= a.flag_tau * (cdf[lo + 2] - cu));
        }
        if (!(denom > 1e-6f)) denom = 1e-6f;
        const float tt = (u - cl) / denom;
        mg[nc + s] = bl + (bu - bl) * tt;
    }
    if (flagging) { // wave-uniform branch; every lane votes
        const bool any_light = __any(light) || near_cut; // an ill-conditioned draw, or a transmittance within 0.1 % of the cut
        if (a.flag_list && any_light && lane == 0) a.flag_list[atomicAdd(a.flag_count, 1u)] = (unsigned)ray;
        if (a.flag_out && lane == 0) a.flag_out[ray] = any_light ? 1 : 0;
    }
    for (int i = lane; i < nc; i += 64) mg[i] = t[i];
    wave_sync();
    if (a.t_new_out) for (int s = lane; s < nf; s += 64) a.t_new_out[(size_t)ray * nf + s] = mg[nc + s];

    // merged.sort_by(partial_cmp) (:419).  Only the sorted VALUES leave this kernel and equal floats are interchangeable, so any
    // correct sort reproduces the reference's stable sort bit for bit: a bitonic network over the next power of two (pad =
    // +inf).  Up to 256 elements (the shipped 64 + 128 configuration) it runs in registers: a lane owns four consecutive
    // elements, compare-exchange distances 1 and 2 stay inside the lane, larger ones exchange whole lanes by __shfl_xor --
    // no LDS traffic, no bank conflicts, no wave barriers.  Wider sorts fall back to the same network in LDS.
    const int P = a.sort_pow2;
    if (P <= 256) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int i = 4 * lane + e; v[e] = i < M ? mg[i] : __builtin_inff(); }
        auto cx = [](float &x, float &y, bool up) { const float lo = fminf(x, y), hi = fmaxf(x, y); x = up ? lo : hi; y = up ? hi : lo; };
        for (int k = 2; k <= 256; k <<= 1) {
            if (k > P) break;                                    // elements beyond P are +inf padding in sorted position
            for (int j = k >> 1; j >= 4; j >>= 1) {              // partner element 4 (lane ^ (j / 4)) + e
                const bool up = ((4 * lane) & k) == 0, lower = ((4 * lane) & j) == 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = __shfl_xor(v[e], j >> 2, 64);
                    v[e] = (lower == up) ? fminf(v[e], o) : fmaxf(v[e], o);
                }
            }
            if (k >= 4) { const bool up = ((4 * lane) & k) == 0; cx(v[0], v[2], up); cx(v[1], v[3], up); cx(v[0], v[1], up); cx(v[2], v[3], up); }
            else        { cx(v[0], v[1], true); cx(v[2], v[3], false); }  // k = 2: direction = bit 1 of the element index
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int i = 4 * lane + e; if (i < M) a.t_fine[(size_t)ray * M + i] = v[e]; }
        return;
    }
    for (int e = M + lane; e < P; e += 64) mg[e] = __builtin_inff();
    wave_sync();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = lane; p < (P >> 1); p += 64) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const float x = mg[i], y = mg[l];
                const bool up = (i & k) == 0;
                const float lo = fminf(x, y), hi = fmaxf(x, y);
                mg[i] = up ? lo : hi; mg[l] = up ? hi : lo;
            }
            wave_sync();
        }
    }
    for (int e = lane; e < M; e += 64) a.t_fine[(size_t)ray * M + e] = mg[e];
}

// ---- compositing: one ray per lane, samples staged through LDS ------------------------------------------
// integrate_ray (src/lib.rs:176-195) is a sequential recurrence per ray (transmittance product with the T < 1e-4 cut, colour
// sums in sample order), so a lane owns a ray and walks its samples in order -- the arithmetic and its order are exactly
// the reference's.  A wave = 64 consecutive rays.  Global reads stay line-granular: the wave stages kCompChunk (16)
// samples of its 64 rays at a time in LDS (16 consecutive floats of t / sigma = one 64-B line per ray, 48 floats of rgb =
// three lines), rows padded to odd strides so that the per-lane walk is bank-conflict free.  One wave per workgroup:
// 21 KiB of LDS each, seven workgroups per CU.
constexpr int kCompChunk = 16;
constexpr int kCompTS = kCompChunk + 1;     // row strides (floats)
constexpr int kCompCS = 3 * kCompChunk + 1;

__global__ __launch_bounds__(64) void k_composite(CompositeArgs a) {
    __shared__ float s_t[64 * kCompTS], s_sg[64 * kCompTS], s_col[64 * kCompCS];
    const int lane = threadIdx.x;
    const int ray0 = blockIdx.x * 64;
    const int n = a.n;
    const int my_ray = ray0 + lane;
    const bool live = my_ray < a.n_rays;
    const int sub = lane & 15, rq = lane >> 4; // staging: 4 rays x 16 samples per wave-instruction
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, acc = 0.0f;
    bool cut = false;                           // compute_weights' early break (src/lib.rs:273-279): later weights are 0
    float t_cur = 0.0f;
    for (int c0 = 0; c0 < n; c0 += kCompChunk) {
        const int cs = n - c0 < kCompChunk ? n - c0 : kCompChunk;
        // stage: t[c0 .. c0+cs] (one extra: the next sample's t closes the last interval), sigma, rgb.  All loads are issued
        // before the first LDS store and none is predicated (indices are clamped instead; a clamped slot is never read):
        // a guarded load would sit in its own basic block behind its own s_waitcnt and serialise the memory latency.
        float vt[16], vs[16], vc[16][3];
        const int se = sub < cs ? sub : cs - 1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int gr = ray0 + 4 * k + rq; gr = gr < a.n_rays ? gr : a.n_rays - 1;
            const size_t base = (size_t)gr * n + c0;
            vt[k] = a.t[base + se]; vs[k] = a.sigma[base + se];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = 16 * q + sub;
                vc[k][q] = a.rgb[3 * base + (e < 3 * cs ? e : 3 * cs - 1)];
            }
        }
        int er = my_ray < a.n_rays ? my_ray : a.n_rays - 1;
        const float t_ext = (c0 + cs < n) ? a.t[(size_t)er * n + c0 + cs] : a.far_; // last interval ends at far (:180)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rr = 4 * k + rq;
            s_t[rr * kCompTS + se] = vt[k]; s_sg[rr * kCompTS + se] = vs[k];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = 16 * q + sub;
                s_col[rr * kCompCS + (e < 3 * cs ? e : 3 * cs - 1)] = vc[k][q];
            }
        }
        s_t[lane * kCompTS + cs] = t_ext; // slot cs is written by nobody else (clamped duplicates land on slots < cs)
        __syncthreads();
        if (c0 == 0) t_cur = s_t[lane * kCompTS];
        for (int i = 0; i < cs; ++i) {
            const float t_next = s_t[lane * kCompTS + i + 1];
            float delta = t_next - t_cur;              // sample_alpha
            if (delta < 0.0f) delta = 0.0f;
            const float al = 1.0f - expf(-s_sg[lane * kCompTS + i] * delta);
            t_cur = t_next;
            float wi = 0.0f;
            if (!cut) {
                wi = T * al;
                T *= 1.0f - al;
                cut = T < 1e-4f;
            }
            if (a.w_out && live) a.w_out[(size_t)my_ray * n + c0 + i] = wi;
            const float *col = &s_col[lane * kCompCS + 3 * i];
            r += col[0] * wi; g += col[1] * wi; b += col[2] * wi; // integrate_ray :185-194, sample order
            acc += wi;
        }
        __syncthreads();
    }
    if (live) {
        const float bg = 1.0f * (1.0f - acc);
        float *o = a.out + 3 * (size_t)my_ray;
        o[0] = r + bg; o[1] = g + bg; o[2] = b + bg;
    }
}

// ---- SSAA box filter: out[i][j][c] = (sum over s x s sub-rays, row-major) * (1/(s*s)) -------------------
__global__ void k_box_downsample(const float *__restrict__ rays, float *__restrict__ out, int w, int h, int s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h * 3) return;
    const int c = idx % 3, j = (idx / 3) % w, i = idx / (3 * w);
    const int RW = w * s;
    float acc = 0.0f;
    for (int di = 0; di < s; ++di)
        for (int dj = 0; dj < s; ++dj) acc += rays[3 * ((size_t)(i * s + di) * RW + (j * s + dj)) + c];
    out[idx] = acc * (1.0f / (float)(s * s));
}

// ---- zero certification (nerf_render_opts.certify_zero): which samples does the exact kernel have to look at? ------------
// `pre` holds the bf16 kernel's density PRE-activations of all samples of a pass (rays x spr).  Two exact facts of the reference make a
// sample's exact evaluation unnecessary:
//   (Z) its exact density is 0 (weight T * (1 - exp(-0 * delta)) = 0, src/lib.rs:271-272) -- CERTIFIED when the 16-bit (pre-filter: f16 or bf16 operands) pre-activation is
//       below -margin (margin = several times the bf16-vs-exact difference ever seen near 0; audited, see k_cert_audit);
//   (C) it lies behind the ray's T < 1e-4 cut (src/lib.rs:276-279: every later weight is zero-filled whatever its density) -- PREDICTED
//       from the bf16 densities, VERIFIED with the exact ones (k_cert_verify), so nothing rests on the prediction but the amount of work.
// k_cert_plan walks every ray front to back (lanes = samples, optical depth by a wave prefix sum: a prediction needs no particular
// summation order) and finds j* = the first sample whose bf16 transmittance exp(-depth) is below the prediction threshold (depth_limit =
// -ln of it; kept well below the exact cut's 1e-4 so that the exact cut almost always falls inside [0, j*)).  Then
//   * samples [0, j*): the UNCERTAIN ones (not certified: positive, near zero, NaN) go on the phase-1 list; of the certified ones a
//     deterministic 1 / (audit_mask_near + 1) of those certified by less than twice the margin and 1 / (audit_mask + 1) of the rest go on the list too, with bit 31 set: the exact kernel evaluates them like any other listed
//     sample but stores the raw pre-activation, which k_cert_audit compares with 0 and with the bf16 value (remembered in `aux`);
//   * samples [j*, spr): uncertain ones are only MARKED (kCertMarker in the buffer); k_cert_verify either clears the marks (the exact
//     transmittance over [0, j*) did fall below 1e-4: the ray is decided) or lists them for a second exact launch (rare);
//   * the buffer is zeroed otherwise: it becomes the exact pass's density buffer, whose listed entries the exact kernel overwrites.
// List order is arbitrary (one atomic per workgroup and list part, entries staged in LDS); results do not depend on it.  The list has a
// CAPACITY: entries beyond it are counted, not stored -- the host sees count > capacity after the frame, grows the list and renders the
// frame again (nerf_api.cpp).
constexpr float kCertMarker = -1.0f;
constexpr int kCertPlanMaxLds = 159 * 1024; // dynamic LDS of k_cert_plan (its reservation counters are static LDS on top)
constexpr int kPlanRays = 8; // rays per wave between two flushes of its LDS staging area

__device__ __forceinline__ bool cert_audit_pick(unsigned idx, unsigned salt, unsigned mask) {
    return (((idx ^ salt) * 0x9E3779B1u) >> 7 & mask) == 0u; // a fixed pseudo-random 1 / (mask + 1) of the sample indices
}

// dynamic LDS: 4 waves x (rays_per_wave x spr entries + kAuxStage {stage position, pre-filter pre-activation} pairs)
constexpr int kAuxStage = 256; // audited certificates a wave can stage between two flushes (at most rays_per_wave x spr / 16 = 96 for 8 rays of 192 samples); beyond: not audited
__global__ __launch_bounds__(256) void k_cert_plan(CertPlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int spr = a.spr;
    const size_t per_wave = (size_t)a.rays_per_wave * spr + 2 * kAuxStage;
    unsigned *stage = lds_u + (size_t)wv * per_wave;
    unsigned *aux_stage = stage + (size_t)a.rays_per_wave * spr;
    const int ray0 = (blockIdx.x * 4 + wv) * a.rays_per_wave;
    unsigned staged = 0, staged_back = 0, n_aux = 0; // wave-uniform
    const unsigned stage_cap = (unsigned)a.rays_per_wave * (unsigned)spr;
    for (int rr = 0; rr < a.rays_per_wave; ++rr) {
        const int ray = ray0 + rr;
        if (ray >= a.n_rays) break; // wave-uniform
        const size_t base = (size_t)ray * spr;
        float carry = 0.0f; // optical depth in front of this group (bf16 densities)
        int jstar = spr;
        for (int gb = 0; gb < spr; gb += 256) { // four groups of 64 samples per batch: their loads are issued together (unpredicated, clamped
          float pre_q[4], od_q[4];             // indices) -- one group at a time left this kernel latency-bound at 2 ms per 123 M samples
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = gb + 64 * q + lane, ic = i < spr ? i : spr - 1;
            const float pr = a.pre[base + ic], t0 = a.t[base + ic], tn = a.t[base + (ic + 1 < spr ? ic + 1 : ic)];
            float delta = (ic + 1 < spr ? tn : a.far_) - t0;
            if (delta < 0.0f) delta = 0.0f;
            pre_q[q] = pr;
            od_q[q] = fmaxf(pr, 0.0f) * delta; // NaN pre-activation -> 0 here, and "uncertain" below
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int g0 = gb + 64 * q;
            if (g0 >= spr) break; // wave-uniform
            const int i = g0 + lane;
            const bool in = i < spr;
            const float pre = in ? pre_q[q] : 0.0f, od = in ? od_q[q] : 0.0f;
            const bool unc = in && !(pre < -a.margin && pre >= -3.0e38f); // NaN, +-inf (an f16 pre-filter that left its range) are never certificates
            bool dead = jstar < spr; // a previous group already reached the predicted cut
            if (!dead) {
                float v = od;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
                const float before = carry + (v - od); // depth in front of sample i
                const unsigned long long dm = __ballot(in && before > a.depth_limit);
                if (dm) jstar = g0 + (int)__builtin_ctzll(dm);
                dead = i >= jstar;
                carry += __shfl(v, 63, 64);
            }
            // denser where a certificate is at risk: 1 / (audit_mask_near + 1) of the samples certified by less than twice the margin,
            // 1 / (audit_mask + 1) of the others (whose bf16 error still feeds the any-depth error statistic)
            bool audit = in && !dead && !unc && cert_audit_pick((unsigned)(base + i), a.audit_salt, pre > -2.0f * a.margin ? a.audit_mask_near : a.audit_mask);
            const unsigned long long am = __ballot(audit);
            const unsigned aux_at = n_aux + (unsigned)__popcll(am & ((1ull << lane) - 1ull));
            audit = audit && aux_at < (unsigned)kAuxStage; // no room to remember its bf16 value: certified without audit, like its 63 siblings
            const bool listed = (in && !dead && unc) || audit;
            // probable zeros (and the audited certificates, which are zeros unless the certificate is wrong) are staged from the END of the
            // wave's area downwards and leave for the back part of the list: tiles of zeros skip their colour heads (skip_empty)
            const bool to_back = listed && a.count_back && (audit || pre < -a.zero_threshold);
            if (in) a.pre[base + i] = (dead && unc) ? kCertMarker : 0.0f;
            const unsigned long long lm = __ballot(listed && !to_back), bm = __ballot(to_back);
            const unsigned at = to_back ? stage_cap - 1u - (staged_back + (unsigned)__popcll(bm & ((1ull << lane) - 1ull)))
                                        : staged + (unsigned)__popcll(lm & ((1ull << lane) - 1ull));
            if (listed) stage[at] = (unsigned)(base + i) | (audit ? 0x80000000u : 0u);
            if (audit) { aux_stage[2 * aux_at] = at; aux_stage[2 * aux_at + 1] = __float_as_uint(pre); }
            staged += (unsigned)__popcll(lm);
            staged_back += (unsigned)__popcll(bm);
            n_aux = n_aux + (unsigned)__popcll(am);
            n_aux = n_aux < (unsigned)kAuxStage ? n_aux : (unsigned)kAuxStage;
          }
        }
        if (lane == 0) a.jstar[ray] = jstar;
    }
    // One reservation per WORKGROUP and list part: a same-address atomic costs ~7 ns at the L2 and they serialise -- one per wave was half of
    // this kernel's 2 ms per 123 M samples.  Every wave of the block gets here (no early return above).
    __shared__ unsigned s_cnt[4][3], s_base[3];
    if (lane == 0) { s_cnt[wv][0] = staged; s_cnt[wv][1] = staged_back; s_cnt[wv][2] = n_aux; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned total = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        unsigned *ctr = threadIdx.x == 0 ? a.count : threadIdx.x == 1 ? a.count_back : a.aux_count;
        s_base[threadIdx.x] = (total && ctr) ? atomicAdd(ctr, total) : 0u;
    }
    __syncthreads();
    unsigned at_front = s_base[0], at_back = s_base[1], at_aux = s_base[2];
    for (int w = 0; w < wv; ++w) { at_front += s_cnt[w][0]; at_back += s_cnt[w][1]; at_aux += s_cnt[w][2]; }
    if (staged + staged_back == 0) return; // wave-uniform
    if (n_aux) { // remember {sample, pre-filter pre-activation} of the audited certificates; one that does not fit is not audited (flag cleared)
        for (unsigned k = lane; k < n_aux; k += 64) {
            const unsigned sp = aux_stage[2 * k];
            if (at_aux + k < a.aux_capacity) { a.aux[2 * (size_t)(at_aux + k)] = stage[sp] & 0x7fffffffu; a.aux[2 * (size_t)(at_aux + k) + 1] = aux_stage[2 * k + 1]; }
            else stage[sp] &= 0x7fffffffu;
        }
        wave_sync();
    }
    for (unsigned k = lane; k < staged; k += 64)
        if (at_front + k < a.capacity) a.list[at_front + k] = stage[k];
    // (front and back parts meet only in a frame whose list is too short: the host renders that frame again)
    for (unsigned k = lane; k < staged_back; k += 64)
        if (at_back + k < a.capacity) a.list[a.capacity - 1u - (at_back + k)] = stage[stage_cap - 1u - k];
}

// Exact transmittance over [0, j*) of every ray with a predicted cut (j* < spr): one wave per ray, alpha in parallel, the recurrence as
// compute_weights has it (src/lib.rs:261-280; the same operations in the same order as weights_scan / k_composite above, on the same
// buffer contents: exact densities of the listed samples, exact zeros of the certified ones).  If it falls below 1e-4 inside the prefix,
// every sample from j* on has weight 0 whatever its density: the marks are cleared and the ray is done.  Otherwise (the 16-bit prediction
// was too optimistic) the marked samples are listed for the second exact launch.  LDS: 4 waves x spr floats.
__global__ __launch_bounds__(256) void k_cert_verify(CertVerifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= a.n_rays) return;
    const int spr = a.spr, j = a.jstar[ray];
    if (j >= spr) return; // no predicted cut: everything uncertain was on the first list
    float *alpha = lds_f + (size_t)wv * spr;
    const float *t = a.t + (size_t)ray * spr;
    float *sg = a.sigma + (size_t)ray * spr;
    for (int i = lane; i < j; i += 64) alpha[i] = sample_alpha(t, sg, i, spr, a.far_);
    wave_sync();
    float T = 1.0f;
    bool cut = false;
#pragma unroll 8
    for (int i = 0; i < j; ++i) { // wave-uniform, branch-free form of the early break (see weights_scan)
        const float al = alpha[i];
        T = cut ? T : T * (1.0f - al);
        cut = cut || T < 1e-4f;
    }
    unsigned n_more = 0;
    for (int g0 = j; g0 < spr; g0 += 64) {
        const int i = g0 + lane;
        const bool marked = i < spr && sg[i] == kCertMarker;
        if (marked) sg[i] = 0.0f;
        if (cut) continue; // wave-uniform
        const unsigned long long mm = __ballot(marked);
        if (!mm) continue;
        unsigned at = 0;
        if (lane == 0) at = atomicAdd(a.count, (unsigned)__popcll(mm)); // rare path: one atomic per 64 samples is fine
        at = (unsigned)__builtin_amdgcn_readfirstlane((int)at) + (unsigned)__popcll(mm & ((1ull << lane) - 1ull));
        if (marked && at < a.capacity) a.list[at] = (unsigned)((size_t)ray * spr + i);
        n_more += (unsigned)__popcll(mm);
    }
    if (n_more && lane == 0 && a.fallback_rays) atomicAdd(a.fallback_rays, 1u);
}

// The audit of (Z): `aux` holds {sample, pre-filter pre-activation} of the certified samples that went on the list with bit 31 set -- the exact
// kernel evaluated them all the same and left its RAW pre-activation in the density buffer.  A positive one is a VIOLATION (the
// certificate was wrong; this sample now holds its true density, its unaudited siblings do not).  Otherwise -pre is how far the sample
// stood from a positive density (HEADROOM = the minimum over the audited samples), and |pre_16bit - pre_exact| is what the pre-filter got
// wrong on a sample it certified (MAX ERROR, at any depth below 0 -- samples right at the margin are rare in a network with large
// pre-activations, its errors are not).  The host widens the margin and renders again when either uses up more than half the margin.
// The buffer entry is reset to the exact kernel's value for such a sample: relu(pre) = 0.
// audit[0] += audited, audit[1] += violations, audit[2] = max(0x7f800000 - bits(headroom)) (zero-initialised = +inf), audit[3] = max bits(|error|)
__global__ __launch_bounds__(256) void k_cert_audit(const unsigned *__restrict__ aux, const unsigned *__restrict__ aux_count, unsigned aux_capacity,
                                                    float *__restrict__ sigma, unsigned *__restrict__ audit) {
    unsigned n = *aux_count;
    n = n < aux_capacity ? n : aux_capacity;
    const int lane = threadIdx.x & 63;
    unsigned n_aud = 0, n_bad = 0, best = 0, worst = 0;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const unsigned idx = aux[2 * (size_t)k];
        const float pre_b = __uint_as_float(aux[2 * (size_t)k + 1]);
        const float v = sigma[idx];
        ++n_aud;
        if (v <= 0.0f) {
            const unsigned x = 0x7f800000u - __float_as_uint(-v); // -v >= 0 (or +0 for -0): bits ordered like the floats
            best = x > best ? x : best;
        } else ++n_bad; // positive or NaN
        const float err = fabsf(pre_b - v);
        if (err <= 3.0e38f) { const unsigned e = __float_as_uint(err); worst = e > worst ? e : worst; }
        if (!(v > 0.0f)) sigma[idx] = 0.0f; // relu of a non-positive (or NaN: fmaxf(NaN, 0) = 0) pre-activation
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_aud += __shfl_xor(n_aud, off, 64); n_bad += __shfl_xor(n_bad, off, 64);
        unsigned o = __shfl_xor(best, off, 64); best = o > best ? o : best;
        o = __shfl_xor(worst, off, 64); worst = o > worst ? o : worst;
    }
    if (lane == 0 && n_aud) { atomicAdd(audit, n_aud); if (n_bad) atomicAdd(audit + 1, n_bad); atomicMax(audit + 2, best); atomicMax(audit + 3, worst); }
}

size_t cert_plan_lds_bytes(int spr, int *rays_per_wave) {
    int r = (int)((size_t)(96 * 1024) / ((size_t)4 * spr * sizeof(unsigned)));
    r = r > kPlanRays ? kPlanRays : r;
    if (r < 1) r = 1;
    if (rays_per_wave) *rays_per_wave = r;
    return (size_t)4 * ((size_t)r * spr + 2 * kAuxStage) * sizeof(unsigned);
}

hipError_t launch_cert_plan(const CertPlanArgs &a, hipStream_t st) {
    if (a.n_rays <= 0 || a.spr <= 0) return hipSuccess;
    CertPlanArgs b = a;
    const size_t lds = cert_plan_lds_bytes(a.spr, &b.rays_per_wave);
    if (lds > (size_t)kCertPlanMaxLds) return hipErrorInvalidValue;
    const int rays_per_block = 4 * b.rays_per_wave;
    hipLaunchKernelGGL(k_cert_plan, dim3((a.n_rays + rays_per_block - 1) / rays_per_block), dim3(256), lds, st, b);
    return hipGetLastError();
}

hipError_t launch_cert_verify(const CertVerifyArgs &a, hipStream_t st) {
    if (a.n_rays <= 0 || a.spr <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * a.spr * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_cert_verify, dim3((a.n_rays + 3) / 4), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_cert_audit(const unsigned *aux, const unsigned *aux_count, unsigned aux_capacity, float *sigma, unsigned *audit, int n_cus, hipStream_t st) {
    if (aux_capacity == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cert_audit, dim3(n_cus > 0 ? 2 * n_cus : 512), dim3(256), 0, st, aux, aux_count, aux_capacity, sigma, audit);
    return hipGetLastError();
}

// ---- multi-GPU: bands -> frame ---------------------------------------------------------------------------------------------------
// `slots` = n bands of slot_floats floats each (band b's rows packed at its start); band b holds the rows of the stripes b, b + n, ... of
// the frame (stripe = `stripe` rows; stripe == 0: contiguous bands, the first h % n bands one row longer).  One thread per float4-less float:
// the frame is 7.7 MB -- a copy kernel, HBM-bound.
__global__ void k_bands_to_frame(const float *__restrict__ slots, float *__restrict__ frame, int w, int h, int n, int stripe, size_t slot_floats) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)w * 3;
    if (idx >= row_floats * h) return;
    const int y = (int)(idx / row_floats);
    const size_t x = idx % row_floats;
    int b, j;
    if (stripe > 0) { const int k = y / stripe; b = k % n; j = (k / n) * stripe + y % stripe; }
    else {
        const int base = h / n, rem = h % n, split = rem * (base + 1);
        if (y < split) { b = y / (base + 1); j = y % (base + 1); } else { b = rem + (y - split) / base; j = (y - split) % base; }
    }
    frame[idx] = slots[(size_t)b * slot_floats + (size_t)j * row_floats + x];
}

hipError_t launch_bands_to_frame(const float *slots, float *frame, int w, int h, int n, int stripe, size_t slot_floats, hipStream_t st) {
    const size_t total = (size_t)w * 3 * h;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bands_to_frame, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slots, frame, w, h, n, stripe, slot_floats);
    return hipGetLastError();
}

// ---- launchers -----------------------------------------------------------------------------------------
hipError_t launch_ray_dirs(const RayGenArgs &a, float *dirs, hipStream_t st) {
    if (a.n_rays <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ray_dirs, dim3((a.n_rays + 255) / 256), dim3(256), 0, st, a, dirs);
    return hipGetLastError();
}

hipError_t launch_stratified(const RayGenArgs &a, int count, float near_, float far_, uint64_t seed, float *t,
                             hipStream_t st) {
    if (a.n_rays <= 0 || count <= 0) return hipSuccess;
    const long long total = (long long)a.n_rays * ((count + 3) / 4);
    hipLaunchKernelGGL(k_stratified, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, count, near_, far_,
                       (uint32_t)seed, (uint32_t)(seed >> 32), t);
    return hipGetLastError();
}

static int pow2_at_least(int v) { int p = 2; while (p < v) p <<= 1; return p; }
size_t resample_lds_bytes(int nc, int nf) { return (size_t)4 * (7 * nc + pow2_at_least(nc + nf)) * sizeof(float); }
size_t composite_lds_bytes(int) { return sizeof(float) * 64 * (2 * kCompTS + kCompCS); } // static, independent of n

hipError_t sampling_init(void) {
    hipError_t e = hipFuncSetAttribute((const void *)k_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cert_plan, hipFuncAttributeMaxDynamicSharedMemorySize, kCertPlanMaxLds); // (+ 60 B static)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cert_verify, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

hipError_t launch_resample(const ResampleArgs &a, hipStream_t st) {
    if (a.n_rays <= 0) return hipSuccess;
    ResampleArgs b = a;
    b.sort_pow2 = pow2_at_least(a.nc + a.nf);
    hipLaunchKernelGGL(k_resample, dim3((a.n_rays + 3) / 4), dim3(256), resample_lds_bytes(a.nc, a.nf), st, b);
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs &a, hipStream_t st) {
    if (a.n_rays <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_composite, dim3((a.n_rays + 63) / 64), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_box_downsample(const float *rays, float *out, int w, int h, int s, hipStream_t st) {
    const int total = w * h * 3;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_box_downsample, dim3((total + 255) / 256), dim3(256), 0, st, rays, out, w, h, s);
    return hipGetLastError();
}
