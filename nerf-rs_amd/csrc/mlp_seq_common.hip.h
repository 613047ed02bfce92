// mlp_seq_common.hip.h -- the arithmetic-independent half of exact dead-sample skipping (skip_dead), shared by the f32
// kernels (mlp_kernel_seq.hip) and the bf16x3 kernels (mlp_kernel_bf16x3.hip): the device-side ray queue, the workgroup vote,
// the inputs of a 32-sample chunk, the reference's transmittance recurrence with its T < 1e-4 cut, and the compacted export /
// import of the trunk outputs of the live samples.  A kernel supplies only its layers between chunk_inputs() and chunk_finish().
//
// Reference semantics (src/lib.rs:261-280, compute_weights): T = 1; for each sample front to back: delta (clamped at 0, last one
// far - t), alpha = 1 - exp(-sigma delta), w = T alpha, T = T (1 - alpha); once T < 1e-4 every later weight is 0.  The operations
// and their order below are k_composite's (sampling_kernels.hip), so "retire" and "live" decide exactly what k_composite will use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"

namespace mlpseq {

constexpr int kH8TileFloats = 32 * 64 * 4; // one 32-sample export tile: [t * 4 + q][lane][4 floats] = 32 KiB

// What a wave is working on (wave-uniform): one ray at a time, its chunks of 32 samples front to back.
struct RayWork {
    int ray, chunk, n_chunks;
    float T;                         // transmittance in front of the current chunk
    unsigned long long samples_done; // statistics: samples this wave evaluated (a ray's last chunk may be partial)
    unsigned long long live_done;    // statistics: samples with weight > 0 this wave found (in-kernel colour passes only)
};

__device__ __forceinline__ void work_init(RayWork &W, const SeqArgs &A) {
    W.n_chunks = (A.samples_per_ray + 31) >> 5;
    W.ray = A.n_rays; W.chunk = W.n_chunks; // no ray yet
    W.T = 1.0f;
    W.samples_done = 0;
    W.live_done = 0;
}

// Take the next ray from the device-side queue if the current one is finished, then vote: the four waves of a workgroup walk the
// weight stream in lockstep, so they leave together once nobody has a ray.  Contains one workgroup barrier.  `vote` = 4 ints of LDS.
__device__ __forceinline__ void work_take(RayWork &W, const SeqArgs &A, int lane) {
    if (W.chunk >= W.n_chunks) {
        unsigned r = 0;
        if (lane == 0) r = atomicAdd(A.ray_counter, 1u);
        r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
        if (A.ray_list) { // queue entries are positions in a device-side ray list (hybrid sampling)
            const unsigned v = r < *A.ray_list_count ? A.ray_list[r] : (unsigned)A.n_rays;
            W.ray = __builtin_amdgcn_readfirstlane((int)v);
        } else W.ray = r < (unsigned)A.n_rays ? (int)r : A.n_rays;
        W.chunk = 0;
        W.T = 1.0f;
    }
}

__device__ __forceinline__ bool work_vote(bool has_work, LDS_AS int *vote, int wave, int lane) {
    if (lane == 0) vote[wave] = has_work ? 1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return (vote[0] | vote[1] | vote[2] | vote[3]) != 0;
}

__device__ __forceinline__ bool work_acquire(RayWork &W, const SeqArgs &A, LDS_AS int *vote, int wave, int lane) {
    work_take(W, A, lane);
    return work_vote(W.ray < A.n_rays, vote, wave, lane);
}

// Lane p's sample of the chunk: position p = origin + d_hat * t with the multiply and the add rounded separately (src/lib.rs:436).
struct ChunkIn {
    size_t base;  // index of the ray's first sample
    int s;        // sample index within the ray
    bool has, valid;
    float t, t_next, dx, dy, dz, px, py, pz;
};

__device__ __forceinline__ ChunkIn chunk_inputs(const RayWork &W, const SeqArgs &A, int p) {
    ChunkIn c;
    const int M = A.samples_per_ray;
    c.has = W.ray < A.n_rays;
    c.s = W.chunk * 32 + p;
    c.valid = c.has && c.s < M;
    c.base = (size_t)(c.has ? W.ray : 0) * M;
    c.t = A.t[c.base + (c.s < M ? c.s : M - 1)];
    c.t_next = A.t[c.base + (c.s + 1 < M ? c.s + 1 : M - 1)];
    const float *dv = A.ray_dirs + 3 * (size_t)(c.has ? W.ray : 0);
    c.dx = dv[0]; c.dy = dv[1]; c.dz = dv[2];
    c.px = __fadd_rn(A.origin[0], __fmul_rn(c.dx, c.t));
    c.py = __fadd_rn(A.origin[1], __fmul_rn(c.dy, c.t));
    c.pz = __fadd_rn(A.origin[2], __fmul_rn(c.dz, c.t));
    return c;
}

__device__ __forceinline__ float lane_value(float v, int k) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

// After the trunk, part 1: store sigma and continue the transmittance recurrence through the chunk.  Returns which samples are LIVE
// (weight > 0: exactly the samples whose colour reaches the pixel (B)) and whether the ray is cut inside this chunk (A).
struct LiveInfo {
    bool live;               // this lane's sample (both lane-halves of a column agree)
    unsigned long long mask; // live columns (bit p)
    int n_live;
    bool cut;
};

__device__ __forceinline__ LiveInfo chunk_scan(RayWork &W, const SeqArgs &A, const ChunkIn &c, float sigma, int p, int h) {
    const int M = A.samples_per_ray;
    if (c.valid && h == 0) A.sigma_out[c.base + c.s] = sigma;
    float delta = (c.s + 1 < M) ? c.t_next - c.t : A.far_ - c.t;
    if (delta < 0.0f) delta = 0.0f;
    const float alpha = c.valid ? 1.0f - expf(-sigma * delta) : 0.0f;
    float my_w = 0.0f, T = W.T;
    bool cut = false;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float al = lane_value(alpha, k);
        const float wk = cut ? 0.0f : T * al;
        if (p == k) my_w = wk;
        T = cut ? T : T * (1.0f - al);
        cut = cut || T < 1e-4f;
    }
    W.T = T;
    if (c.has) { const int left = M - 32 * W.chunk; W.samples_done += (unsigned)(left < 32 ? left : 32); }
    LiveInfo li;
    li.live = my_w > 0.0f;
    li.mask = __ballot(li.live) & 0xffffffffull;
    li.n_live = __popcll(li.mask);
    li.cut = cut;
    return li;
}

// Part 2: advance to the next chunk -- or retire the ray at the cut (A): behind it nothing of this ray is evaluated.
__device__ __forceinline__ void chunk_advance(RayWork &W, const SeqArgs &A, const ChunkIn &c, bool cut, int p, int h) {
    const int M = A.samples_per_ray;
    if (A.zero_fill_after_cut && cut && c.has && h == 0)
        for (int cc = W.chunk + 1; cc < W.n_chunks; ++cc)
            if (cc * 32 + p < M) A.sigma_out[c.base + cc * 32 + p] = 0.0f;
    W.chunk = (cut || !c.has) ? W.n_chunks : W.chunk + 1;
}

// The two-launch form (split arithmetics): export the trunk outputs `Y` (eight accumulator tiles in the C/D layout of the 32x32
// MFMAs) of the live samples to the compacted HBM buffer of the colour kernel.
template <bool EXPORT, class Tiles>
__device__ __forceinline__ void chunk_finish(RayWork &W, const SeqArgs &A, const ChunkIn &c, float sigma, const Tiles &Y, int lane, int p, int h) {
    const LiveInfo li = chunk_scan(W, A, c, sigma, p, h);
    if (EXPORT) {
        if (li.n_live) {
            unsigned b = 0;
            if (lane == 0) b = atomicAdd(A.live_count, (unsigned)li.n_live);
            b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
            if (li.live) {
                const unsigned slot = b + (unsigned)__popcll(li.mask & ((1ull << p) - 1ull));
                float *dst = A.h8 + (size_t)(slot >> 5) * kH8TileFloats + ((slot & 31) + 32 * h) * 4;
#pragma unroll
                for (int tt = 0; tt < 8; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
                        v[0] = Y[tt][4 * q + 0]; v[1] = Y[tt][4 * q + 1]; v[2] = Y[tt][4 * q + 2]; v[3] = Y[tt][4 * q + 3];
                        *(f32x4 *)(dst + (tt * 4 + q) * 256) = v;
                    }
                if (h == 0) A.slot_point[slot] = (unsigned)(c.base + c.s);
            }
        }
    }
    chunk_advance(W, A, c, li.cut, p, h);
}

__device__ __forceinline__ void work_done(const RayWork &W, const SeqArgs &A, int lane) {
    if (A.stats && lane == 0 && W.samples_done) atomicAdd(A.stats, W.samples_done);
    if (A.live_count && lane == 0 && W.live_done) atomicAdd(A.live_count, (unsigned)W.live_done); // in-kernel colour passes: a statistic only
}

// ---- colour kernel side: one compacted slot per MFMA column ------------------------------------------------------------
struct ColourIn {
    unsigned i; // sample index the colour goes to
    bool valid;
    float dx, dy, dz;
};

__device__ __forceinline__ int colour_tiles(const ColourArgs &A, unsigned *n_live) {
    *n_live = *A.live_count;
    return (int)((*n_live + (unsigned)nerfmlp::kPointsPerBlock - 1u) / (unsigned)nerfmlp::kPointsPerBlock);
}

template <class Tiles>
__device__ __forceinline__ ColourIn colour_inputs(const ColourArgs &A, unsigned n_live, int tile, int wave, int lane, Tiles &Y) {
    using namespace nerfmlp;
    ColourIn c;
    const unsigned slot = (unsigned)tile * kPointsPerBlock + wave * kPointsPerWave + (lane & 31);
    c.valid = slot < n_live;
    c.i = A.slot_point[c.valid ? slot : n_live - 1];
    const float *dv = A.ray_dirs + 3 * (size_t)(c.i / (unsigned)A.samples_per_ray);
    c.dx = dv[0]; c.dy = dv[1]; c.dz = dv[2];
    const float *src = A.h8 + (size_t)(tile * kWavesPerBlock + wave) * kH8TileFloats + lane * 4;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4 *)(src + (tt * 4 + q) * 256);
            Y[tt][4 * q + 0] = v[0]; Y[tt][4 * q + 1] = v[1]; Y[tt][4 * q + 2] = v[2]; Y[tt][4 * q + 3] = v[3];
        }
    return c;
}

__device__ __forceinline__ void colour_store(const ColourArgs &A, const ColourIn &c, const float (&rgb)[3], int h) {
    if (c.valid && h == 0) {
        A.rgb_out[3 * (size_t)c.i + 0] = rgb[0];
        A.rgb_out[3 * (size_t)c.i + 1] = rgb[1];
        A.rgb_out[3 * (size_t)c.i + 2] = rgb[2];
    }
}

// launch geometry shared by both arithmetics
inline int trunk_blocks(const SeqArgs &a, int n_blocks) {
    const long long wave_rays = ((long long)a.n_rays + nerfmlp::kWavesPerBlock - 1) / nerfmlp::kWavesPerBlock;
    if (n_blocks > wave_rays) n_blocks = (int)wave_rays;
    return n_blocks < 1 ? 1 : n_blocks;
}

} // namespace mlpseq
