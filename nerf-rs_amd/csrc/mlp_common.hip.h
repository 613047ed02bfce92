// mlp_common.hip.h -- device helpers shared by the f32 (mlp_kernel.hip) and bf16 (mlp_kernel_bf16.hip) MLP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LDS_AS __attribute__((address_space(3)))

namespace mlpdev {

// One 1-KiB LDS-DMA piece (64 lanes x 16 B, lane-linear in LDS).  Hidden from the compiler's waitcnt bookkeeping on
// purpose (it would drain vmcnt(0) in front of every later ds_read); completion is enforced by the caller's explicit
// vmcnt + s_barrier.  No instruction offset: the immediate of global_load_lds applies to the global AND the LDS address.
__device__ __forceinline__ void glds_piece(uint32_t lane16, const char *gsrc, uint32_t dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane16), "s"(gsrc), "s"(dst)
                 : "memory");
}

// The same with M0 -- the LDS destination -- written once per chunk quarter (dma_set_dst, where the chunk is selected) and the quarter's
// four pieces addressed by the instruction offset, which advances the global AND the LDS address.  Beside MFMAs a piece costs its wave 25
// cycles with M0 saved / written / restored around it, 17 with M0 written, 0 with M0 left alone (tools/probes/lds_dma_stagger_probe.hip).
// hipcc uses M0 for nothing of its own in these kernels (no LDS-direct, no s_movrel, no sendmsg): tests/test_host_logic.py checks the ISA.
#ifndef NERF_M0_PER_CHUNK
#define NERF_M0_PER_CHUNK 1
#endif
__device__ __forceinline__ void dma_set_dst(uint32_t dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory"); }
__device__ __forceinline__ void glds_piece_m0(uint32_t lane16, const char *gsrc, int i) {
    switch (i) { // i is a constant after unrolling
    case 0:  asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gsrc) : "memory"); break;
    case 1:  asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(gsrc) : "memory"); break;
    case 2:  asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane16), "s"(gsrc) : "memory"); break;
    default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(lane16), "s"(gsrc) : "memory"); break;
    }
}

// ReLU as a signed-integer max on the bit pattern: one v_max_i32 (fmaxf costs an extra canonicalising v_max), and --
// unlike inline asm -- visible to hipcc's hazard recogniser, which must pad the VALU-write -> MFMA-operand-read
// wait states.  Negative floats and -0.0 are negative integers -> +0.0; positives are unchanged.
__device__ __forceinline__ float relu(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// sin and cos of x (the encodings of the lego frustum reach ~1.25e3 rad): k = rint(x * 2/pi), three-constant Cody-Waite
// reduction with FMA (x - k*pi/2 is exact in the first step: a multiple of 2^-23 below 1), Cephes sinf/cosf minimax
// polynomials on [-pi/4, pi/4], quadrant fix-up by sign-bit arithmetic.  Branch-free; max error 9.2e-8 abs for |x| <= 2^18,
// 1.2e-7 for |x| <= 2^20 (the three constants carry 72 bits of pi/2), 3e-6 at 2^22: the documented input domain of
// nerf_forward_batch is |p| <= 2048 (include/nerf_mi355x.h).
__device__ __forceinline__ void fast_sincos(float x, float *s_out, float *c_out) {
    const float k = __builtin_rintf(x * 0.636619772f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.371138828673793e-08f, r);
    r = fmaf(k, 1.7763568394002505e-15f, r);
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    const float s = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    const float c = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const uint32_t q = (uint32_t)(int)k;
    const bool swap = (q & 1u) != 0;
    const uint32_t sb = __builtin_bit_cast(uint32_t, swap ? c : s) ^ ((q & 2u) << 30);
    const uint32_t cb = __builtin_bit_cast(uint32_t, swap ? s : c) ^ (((q + 1u) & 2u) << 30);
    *s_out = __builtin_bit_cast(float, sb);
    *c_out = __builtin_bit_cast(float, cb);
}

template <bool FAST>
__device__ __forceinline__ void sincos_sel(float x, float *s, float *c) {
    if constexpr (FAST) fast_sincos(x, s, c);
    else sincosf(x, s, c);
}

// Positional encoding of a point: this lane-half's 32 slots (mlp_layout.h posSlotFeature; src/network.rs:263-292):
// octaves 5h..5h+4 as idx 6*o + {sin xyz, cos xyz}, idx 30/31 = raw x,y (h = 0) or raw z, zero pad (h = 1).
template <bool FAST>
__device__ __forceinline__ void encode_point(float px, float py, float pz, int h, f32x16 (&E)[2]) {
    float f = h ? 32.0f : 1.0f;
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        float s, c;
        sincos_sel<FAST>(f * px, &s, &c); E[(6 * o + 0) >> 4][(6 * o + 0) & 15] = s; E[(6 * o + 3) >> 4][(6 * o + 3) & 15] = c;
        sincos_sel<FAST>(f * py, &s, &c); E[(6 * o + 1) >> 4][(6 * o + 1) & 15] = s; E[(6 * o + 4) >> 4][(6 * o + 4) & 15] = c;
        sincos_sel<FAST>(f * pz, &s, &c); E[(6 * o + 2) >> 4][(6 * o + 2) & 15] = s; E[(6 * o + 5) >> 4][(6 * o + 5) & 15] = c;
        f *= 2.0f;
    }
    E[1][14] = h ? pz : px;
    E[1][15] = h ? 0.0f : py;
}

// Direction encoding: 16 slots per lane-half (mlp_layout.h dirSlotFeature; src/network.rs:294-330): octaves 2h, 2h+1,
// raw x,y,z on h = 0, zero pads.
template <bool FAST>
__device__ __forceinline__ void encode_dir(float dx, float dy, float dz, int h, f32x16 &D) {
    float f = h ? 4.0f : 1.0f;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float s, c;
        sincos_sel<FAST>(f * dx, &s, &c); D[6 * o + 0] = s; D[6 * o + 3] = c;
        sincos_sel<FAST>(f * dy, &s, &c); D[6 * o + 1] = s; D[6 * o + 4] = c;
        sincos_sel<FAST>(f * dz, &s, &c); D[6 * o + 2] = s; D[6 * o + 5] = c;
        f *= 2.0f;
    }
    D[12] = h ? 0.f : dx; D[13] = h ? 0.f : dy; D[14] = h ? 0.f : dz; D[15] = 0.f;
}

// bf16 kernels only: the same slots by angle doubling.  One accurate sincos per coordinate at the lane-half's base octave,
// then sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a for the next octaves: ~5 VALU per (coordinate, octave) instead of ~23.
// The error roughly doubles per step (4 steps: < 2e-6 absolute), two orders below the bf16 rounding (2^-9 relative) that follows.
__device__ __forceinline__ void encode_point_doubling(float px, float py, float pz, int h, f32x16 (&E)[2]) {
    const float f = h ? 32.0f : 1.0f;
    float s[3], c[3];
    fast_sincos(f * px, &s[0], &c[0]);
    fast_sincos(f * py, &s[1], &c[1]);
    fast_sincos(f * pz, &s[2], &c[2]);
#pragma unroll
    for (int o = 0; o < 5; ++o) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            E[(6 * o + k) >> 4][(6 * o + k) & 15] = s[k];
            E[(6 * o + 3 + k) >> 4][(6 * o + 3 + k) & 15] = c[k];
            if (o < 4) {
                const float s2 = (s[k] + s[k]) * c[k];
                c[k] = fmaf(-2.0f * s[k], s[k], 1.0f);
                s[k] = s2;
            }
        }
    }
    E[1][14] = h ? pz : px;
    E[1][15] = h ? 0.0f : py;
}

__device__ __forceinline__ void encode_dir_doubling(float dx, float dy, float dz, int h, f32x16 &D) {
    const float f = h ? 4.0f : 1.0f;
    float s[3], c[3];
    fast_sincos(f * dx, &s[0], &c[0]);
    fast_sincos(f * dy, &s[1], &c[1]);
    fast_sincos(f * dz, &s[2], &c[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        D[k] = s[k]; D[3 + k] = c[k];
        D[6 + k] = (s[k] + s[k]) * c[k];
        D[9 + k] = fmaf(-2.0f * s[k], s[k], 1.0f);
    }
    D[12] = h ? 0.f : dx; D[13] = h ? 0.f : dy; D[14] = h ? 0.f : dz; D[15] = 0.f;
}

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// rgb head on the VALU + sigmoid (src/network.rs:223, :165) from the four f32 accumulator tiles of the viewdirs layer: one channel.
template <class Tiles>
__device__ __forceinline__ float rgb_channel(const Tiles &V, const LDS_AS float *small, int h, int ch) {
    using namespace nerfmlp;
    const LDS_AS f32x4 *w = (const LDS_AS f32x4 *)(small + kRgbWOff + (h * 3 + ch) * 64);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = w[t * 4 + q];
            a0 = fmaf(wv[0], relu(V[t][4 * q + 0]), a0);
            a1 = fmaf(wv[1], relu(V[t][4 * q + 1]), a1);
            a2 = fmaf(wv[2], relu(V[t][4 * q + 2]), a2);
            a3 = fmaf(wv[3], relu(V[t][4 * q + 3]), a3);
        }
    }
    const float v = xhalf_sum((a0 + a1) + (a2 + a3)) + small[kMiscOff + 1 + ch];
    return 1.0f / (1.0f + expf(-v));
}

template <class Tiles>
__device__ __forceinline__ void rgb_head(const Tiles &V, const LDS_AS float *small, int h, float (&c)[3]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) c[ch] = rgb_channel(V, small, h, ch);
}

// Raw per-point inputs: 6 floats.  MODE_POINTS: position + direction as given (src/network.rs:197).  MODE_RAYS:
// (t, unused, unused) + the ray's unit direction; p = origin + dir_hat * t is formed at use with the multiply and the
// add rounded separately (src/lib.rs:396 / :436).  The index is clamped: padding lanes of the last tile and the
// look-ahead tile read the last point.
struct RawIn { float a, b, c, dx, dy, dz; };

// MLP_MODE_LIST: the list's length lives on the device; n_points is its CAPACITY (entries beyond it were counted, not stored: the host
// renders such a frame again with a larger list, nerf_api.cpp).  The list has a front part growing up from entry 0 and an optional back
// part growing down from entry n_points - 1 (MlpArgs.point_list_count_back).
template <class Args>
__device__ __forceinline__ void list_parts(const Args &A, unsigned &front, unsigned &back) {
    const unsigned cap = (unsigned)A.n_points, nf = *A.point_list_count;
    front = nf < cap ? nf : cap;
    back = 0;
    if (A.point_list_count_back) { const unsigned nb = *A.point_list_count_back; back = nb < cap - front ? nb : cap - front; }
}
template <class Args>
__device__ __forceinline__ int list_length(const Args &A) {
    unsigned f, b;
    list_parts(A, f, b);
    return (int)(f + b);
}

template <int MODE, class Args>
__device__ __forceinline__ RawIn load_raw(const Args &A, int tile_idx, int wave, int p) {
    using namespace nerfmlp;
    RawIn r;
    int i = tile_idx * kPointsPerBlock + wave * kPointsPerWave + p;
    if (MODE == 2) { // slot i of a device-side sample list: entry = sample index | (audited certificate ? 1 << 31 : 0)
        unsigned nfront, nback;
        list_parts(A, nfront, nback);
        const int n = (int)(nfront + nback);
        r.a = 0.f; r.b = 0.f; r.c = 0.f; r.dx = 0.f; r.dy = 0.f; r.dz = 1.f;
        if (n <= 0) return r;
        const unsigned slot = (unsigned)(i < n ? i : n - 1);
        const unsigned entry = A.point_list[slot < nfront ? slot : (unsigned)A.n_points - nback + (slot - nfront)];
        const unsigned idx = entry & 0x7fffffffu;
        const unsigned ray = idx / (unsigned)A.samples_per_ray;
        r.a = A.t[idx]; r.b = __builtin_bit_cast(float, entry);
        r.dx = A.ray_dirs[3 * (size_t)ray]; r.dy = A.ray_dirs[3 * (size_t)ray + 1]; r.dz = A.ray_dirs[3 * (size_t)ray + 2];
        return r;
    }
    i = i < A.n_points ? i : A.n_points - 1;
    if (MODE == 0) {
        r.a = A.pts_soa[i]; r.b = A.pts_soa[(size_t)A.n_points + i]; r.c = A.pts_soa[2 * (size_t)A.n_points + i];
        r.dx = A.dirs_aos[3 * (size_t)i]; r.dy = A.dirs_aos[3 * (size_t)i + 1]; r.dz = A.dirs_aos[3 * (size_t)i + 2];
    } else {
        const int ray = i / A.samples_per_ray;
        r.a = A.t[i]; r.b = 0.f; r.c = 0.f;
        r.dx = A.ray_dirs[3 * (size_t)ray]; r.dy = A.ray_dirs[3 * (size_t)ray + 1]; r.dz = A.ray_dirs[3 * (size_t)ray + 2];
    }
    return r;
}

template <int MODE, class Args>
__device__ __forceinline__ void point_of(const Args &A, const RawIn &in, float &px, float &py, float &pz) {
    if (MODE == 0) {
        px = in.a; py = in.b; pz = in.c;
    } else {
        px = __fadd_rn(A.origin[0], __fmul_rn(in.dx, in.a));
        py = __fadd_rn(A.origin[1], __fmul_rn(in.dy, in.a));
        pz = __fadd_rn(A.origin[2], __fmul_rn(in.dz, in.a));
    }
}

// Empty-tile vote (skip_empty): true iff some point of the workgroup's 128-point tile has sigma > 0.  All four waves call
// it at the same program point; contains one workgroup barrier.  `vote` = 4 ints of LDS scratch.
__device__ __forceinline__ bool tile_has_density(LDS_AS int *vote, bool lane_has, int wave, int lane) {
    const bool any_wave = __any(lane_has);
    if (lane == 0) vote[wave] = any_wave ? 1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return (vote[0] | vote[1] | vote[2] | vote[3]) != 0;
}

} // namespace mlpdev
