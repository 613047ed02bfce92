// mlp_kernel_bf16x3b.hip -- the bf16x3 arithmetic (mlp_kernel_bf16x3.hip: f32 by three-way bf16 split, six products per f32
// product) on v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why: both bf16-family kernels are power-limited (1.9-2.0 GHz at 70-83 % matrix-pipe occupancy), and the 16x16x32
// instruction draws less power per FLOP (tools/probes/mfma_power_probe.hip: the same FLOPs at 2.26 GHz instead of 1.92).
//
// Layout.  A wave owns 32 points = two point blocks pb of 16; lane l = (p = l & 15, g = l >> 4).  A 32-feature activation
// tile T is four 4-register blocks [fb][pb]: register r of block (fb, pb) on lane (p, g) holds feature 32 T + 16 fb + 4 g + r
// of point 16 pb + p -- the C/D layout of the 16x16 MFMA (row = 4 g + r, column = p).  The B operand of the k-step that
// consumes tile T for point block pb is the 8 values [fb = 0: r = 0..3, fb = 1: r = 0..3] (k = 8 g + j <-> feature
// 32 T + 16 (j >> 2) + 4 g + (j & 3)), so a finished accumulator tile is again the next layer's operand without leaving
// registers.  One unit = (input tile, 32-feature output tile nt, 16-feature half ob): three 1-KiB A pieces (w1, w2, w3 of
// the same 16 x 32 block) and 2 x 6 MFMAs (both point blocks).  The weight stream is a pure permutation of the first
// bf16 design's piece array (nerf_api.cpp), chunk counts and pipeline as in mlp_kernel_bf16x3.hip; small parameters are in
// natural feature order.  Encoding slots follow from the same permutation (every lane evaluates the sincos of its own 16
// position slots and 8 direction slots per point).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"

using namespace nerfmlp;
using namespace mlpdev;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCB = kChunkBytesX3, kRS = kRingSlotsX3;
constexpr int kX3Ahead = 1; // operand prefetch distance in units

#include "mlp_x3_pipe.hip.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define XB_PIN() __builtin_amdgcn_sched_barrier(0)

typedef f32x4 Tile[2][2];  // [fb][pb]
struct B3 { u32x4 h, m, l; };
struct PrepState { f32x2 x; uint32_t h, m; };

// pair PAIR (0..7) of an input tile: point block PAIR >> 2, feature block (PAIR >> 1) & 1, registers 2 (PAIR & 1), + 1
template <bool RELU, int PAIR, int STAGE>
__device__ __forceinline__ void prep_stage(const Tile &in, B3 (&b)[2], PrepState &st) {
    constexpr int pb = PAIR >> 2, fb = (PAIR >> 1) & 1, hf = PAIR & 1, word = 2 * fb + hf;
    if constexpr (STAGE == 0) {
        float x0 = in[fb][pb][2 * hf], x1 = in[fb][pb][2 * hf + 1];
        asm volatile("" : "+v"(x0), "+v"(x1));
        if (RELU) { x0 = relu(x0); x1 = relu(x1); }
        st.x = f32x2{x0, x1};
    } else if constexpr (STAGE == 1) {
        st.h = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
        const f32x2 hf2 = {__builtin_bit_cast(float, st.h << 16), __builtin_bit_cast(float, st.h & 0xffff0000u)};
        st.x = st.x - hf2;
    } else if constexpr (STAGE == 2) {
        st.m = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
        const f32x2 mf = {__builtin_bit_cast(float, st.m << 16), __builtin_bit_cast(float, st.m & 0xffff0000u)};
        st.x = st.x - mf;
    } else {
        b[pb].h[word] = st.h; b[pb].m[word] = st.m;
        b[pb].l[word] = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
    }
}

template <bool RELU>
__device__ __forceinline__ void prep_all(const Tile &in, B3 (&b)[2]) {
    static_for<0, 8>([&](auto pc) {
        constexpr int pr = decltype(pc)::value;
        PrepState st;
        prep_stage<RELU, pr, 0>(in, b, st); prep_stage<RELU, pr, 1>(in, b, st);
        prep_stage<RELU, pr, 2>(in, b, st); prep_stage<RELU, pr, 3>(in, b, st);
    });
}

__device__ __forceinline__ void pin_tile(Tile &t) { asm volatile("" : "+a"(t[0][0]), "+a"(t[0][1]), "+a"(t[1][0]), "+a"(t[1][1])); }

// One input tile = one k-step (K = 32): 2 NT units with the prepared operands `bc`; the next tile's operands are split into
// `bn` in the MFMA gaps (pair u / 2, two stages per unit, in the 8-tile layers; pair u, four stages, in the viewdirs layer).
template <int NT, bool HAS_NEXT, bool NRELU, bool NACC>
__device__ __forceinline__ void tile_step(Tile (&out)[8], const B3 (&bc)[2], Tile &nin, B3 (&bn)[2], PipeX &P) {
    if constexpr (HAS_NEXT && NACC) pin_tile(nin);
    const bf16x8 b1[2] = {__builtin_bit_cast(bf16x8, bc[0].h), __builtin_bit_cast(bf16x8, bc[1].h)};
    const bf16x8 b2[2] = {__builtin_bit_cast(bf16x8, bc[0].m), __builtin_bit_cast(bf16x8, bc[1].m)};
    const bf16x8 b3[2] = {__builtin_bit_cast(bf16x8, bc[0].l), __builtin_bit_cast(bf16x8, bc[1].l)};
    PrepState st;
    static_for<0, 2 * NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int nt = u >> 1, ob = u & 1, U = u & 7;
        constexpr int PAIR = NT == 8 ? (u >> 1) : u;
        constexpr int S0 = NT == 8 ? 2 * (u & 1) : 0; // first stage handled in this unit
        bf16x8 a1, a2, a3;
        f32x4 &c0 = out[nt][ob][0], &c1 = out[nt][ob][1];
        pipe_take<U>(P, a1, a2, a3);
        XB_PIN();
        c0 = MFMA32(a3, b1[0], c0); c1 = MFMA32(a3, b1[1], c1);
        XB_PIN();
        pipe_prefetch<U>(P);
        XB_PIN();
        c0 = MFMA32(a2, b2[0], c0); c1 = MFMA32(a2, b2[1], c1);
        XB_PIN();
        pipe_dma<U>(P);
        XB_PIN();
        c0 = MFMA32(a1, b3[0], c0);
        XB_PIN();
        if constexpr (HAS_NEXT) prep_stage<NRELU, PAIR, S0>(nin, bn, st);
        XB_PIN();
        c1 = MFMA32(a1, b3[1], c1);
        c0 = MFMA32(a2, b1[0], c0);
        XB_PIN();
        if constexpr (HAS_NEXT) prep_stage<NRELU, PAIR, S0 + 1>(nin, bn, st);
        XB_PIN();
        c1 = MFMA32(a2, b1[1], c1);
        c0 = MFMA32(a1, b2[0], c0);
        XB_PIN();
        if constexpr (HAS_NEXT && NT == 4) prep_stage<NRELU, PAIR, 2>(nin, bn, st);
        XB_PIN();
        c1 = MFMA32(a1, b2[1], c1);
        c0 = MFMA32(a1, b1[0], c0);
        XB_PIN();
        if constexpr (HAS_NEXT && NT == 4) prep_stage<NRELU, PAIR, 3>(nin, bn, st);
        XB_PIN();
        c1 = MFMA32(a1, b1[1], c1);
        XB_PIN();
    });
    // keep every accumulation chain in program order (see mlp_kernel_bf16.hip)
    static_for<0, NT>([&](auto nc) { constexpr int nt = decltype(nc)::value; pin_tile(out[nt]); });
}

template <int NT>
__device__ __forceinline__ void load_bias(Tile (&out)[8], const LDS_AS float *bias, int g) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const f32x4 v = *(const LDS_AS f32x4 *)(bias + 32 * nt + 16 * ob + 4 * g); // natural feature order
            out[nt][ob][0] = v; out[nt][ob][1] = v;
        }
}

// `bc` = operands of in[0] on entry; ping-pongs bc / bn through the tiles; NEXT describes the tile after in[7] (none if !HAS_LAST_NEXT)
template <int NT, bool RELU>
__device__ __forceinline__ void eight_tiles(Tile (&in)[8], Tile (&out)[8], B3 (&ba)[2], B3 (&bb)[2], PipeX &P) {
    tile_step<NT, true, RELU, true>(out, ba, in[1], bb, P);
    tile_step<NT, true, RELU, true>(out, bb, in[2], ba, P);
    tile_step<NT, true, RELU, true>(out, ba, in[3], bb, P);
    tile_step<NT, true, RELU, true>(out, bb, in[4], ba, P);
    tile_step<NT, true, RELU, true>(out, ba, in[5], bb, P);
    tile_step<NT, true, RELU, true>(out, bb, in[6], ba, P);
    tile_step<NT, true, RELU, true>(out, ba, in[7], bb, P);
    // operands of in[7] are in bb
}

template <bool RELU>
__device__ __forceinline__ void hidden_layer(Tile (&in)[8], Tile (&out)[8], const LDS_AS float *bias, PipeX &P, int g) {
    load_bias<8>(out, bias, g);
    B3 ba[2], bb[2];
    pin_tile(in[0]);
    prep_all<RELU>(in[0], ba);
    eight_tiles<8, RELU>(in, out, ba, bb, P);
    tile_step<8, false, false, false>(out, bb, in[7], ba, P);
}

__device__ __forceinline__ float group_sum(float v) { // sum over the four lane groups of a point
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Position / direction encodings in the slot order the weight permutation induces: block (fb) register r of tile e on lane
// group g is row 16 fb + 4 g + r of the tile = slot 16 e + 4 (2 fb + (g >> 1)) + r of lane-half (g & 1) in the 32x32 layouts
// (mlp_layout.h posSlotFeature / dirSlotFeature).  Every slot is evaluated on its own (one fast_sincos per slot).
__device__ __forceinline__ float pos_slot(int slot, int hh, float x, float y, float z) {
    if (slot >= 30) return hh == 0 ? (slot == 30 ? x : y) : (slot == 30 ? z : 0.0f);
    const int o = 5 * hh + slot / 6, comp = slot % 6, c = comp % 3;
    const float v = c == 0 ? x : (c == 1 ? y : z);
    float s, co;
    fast_sincos(ldexpf(v, o), &s, &co);
    return comp < 3 ? s : co;
}

__device__ __forceinline__ float dir_slot(int slot, int hh, float x, float y, float z) {
    if (slot >= 12) return (hh == 0 && slot < 15) ? (slot == 12 ? x : (slot == 13 ? y : z)) : 0.0f;
    const int o = 2 * hh + slot / 6, comp = slot % 6, c = comp % 3;
    const float v = c == 0 ? x : (c == 1 ? y : z);
    float s, co;
    fast_sincos(ldexpf(v, o), &s, &co);
    return comp < 3 ? s : co;
}

} // namespace

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void nerf_mlp_kernel_bf16x3b(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15;
    const int g = lane >> 4;

    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    PipeX P;
    P.lane16 = lane * 16;
    P.ring_lane = lds + P.lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 6144;
    P.stream_bytes = (FULL ? kChunksFullX3 : kChunksSigmaX3) * kCB;
    P.gbase = (const char *)A.wstream + wave * 6144;
    __syncthreads();
    pipe_start(P);

    const int n_tiles = (A.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    auto raw = [&](int tile_idx, int pb) -> RawIn { // clamped: padding lanes and the look-ahead tile read the last point
        RawIn r;
        int i = tile_idx * kPointsPerBlock + wave * kPointsPerWave + 16 * pb + p;
        i = i < A.n_points ? i : A.n_points - 1;
        if (MODE == MLP_MODE_POINTS) {
            r.a = A.pts_soa[i]; r.b = A.pts_soa[(size_t)A.n_points + i]; r.c = A.pts_soa[2 * (size_t)A.n_points + i];
            r.dx = A.dirs_aos[3 * (size_t)i]; r.dy = A.dirs_aos[3 * (size_t)i + 1]; r.dz = A.dirs_aos[3 * (size_t)i + 2];
        } else {
            const int ray = i / A.samples_per_ray;
            r.a = A.t[i]; r.b = 0.f; r.c = 0.f;
            r.dx = A.ray_dirs[3 * (size_t)ray]; r.dy = A.ray_dirs[3 * (size_t)ray + 1]; r.dz = A.ray_dirs[3 * (size_t)ray + 2];
        }
        return r;
    };
    const int hh = g & 1, gq = g >> 1;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int i0 = tile * kPointsPerBlock + wave * kPointsPerWave + p, i1 = i0 + 16;
        const bool v0 = i0 < A.n_points, v1 = i1 < A.n_points;
        const RawIn in0 = raw(tile, 0), in1 = raw(tile, 1);

        Tile E[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            float px, py, pz;
            point_of<MODE>(A, pb ? in1 : in0, px, py, pz);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) E[e][fb][pb][r] = pos_slot(16 * e + 4 * (2 * fb + gq) + r, hh, px, py, pz);
        }

        Tile X[8], Y[8];
        B3 ba[2], bb[2];
        load_bias<8>(X, small + kBiasOff + 0 * 256, g);            // dense0 (src/network.rs:204)
        prep_all<false>(E[0], ba);
        tile_step<8, true, false, false>(X, ba, E[1], bb, P);
        tile_step<8, false, false, false>(X, bb, E[1], ba, P);
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, g);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, g);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, g);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, g);
        load_bias<8>(Y, small + kBiasOff + 5 * 256, g);            // dense5 on [encoding ; h4] (:209-210)
        prep_all<false>(E[0], ba);
        tile_step<8, true, false, false>(Y, ba, E[1], bb, P);
        tile_step<8, true, true, true>(Y, bb, X[0], ba, P);
        eight_tiles<8, true>(X, Y, ba, bb, P);
        tile_step<8, false, false, false>(Y, bb, X[7], ba, P);
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, g);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, g);

        // alpha head (f32 VALU): sigma = relu(b + sum_F w[F] relu(h8[F])) (src/network.rs:216)
        float al[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                const f32x4 w = *(const LDS_AS f32x4 *)(small + kAlphaWOff + 32 * t + 16 * fb + 4 * g);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    f32x4 x = Y[t][fb][pb];
                    asm volatile("" : "+v"(x));
#pragma unroll
                    for (int r = 0; r < 4; ++r) al[pb] = fmaf(w[r], relu(x[r]), al[pb]);
                }
            }
        const float s0 = fmaxf(group_sum(al[0]) + small[kMiscOff + 0], 0.f);
        const float s1 = fmaxf(group_sum(al[1]) + small[kMiscOff + 0], 0.f);
        if (g == 0 && v0) A.sigma_out[i0] = s0;
        if (g == 1 && v1) A.sigma_out[i1] = s1;

        if (FULL && A.skip_empty) { // exact empty-tile skip, see mlp_kernel.hip
            LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
            const bool any_wg = tile_has_density(vote, (v0 && s0 > 0.0f) || (v1 && s1 > 0.0f), wave, lane);
            if (!any_wg) {
                if (g == 0 && v0) { A.rgb_out[3 * (size_t)i0] = 0.f; A.rgb_out[3 * (size_t)i0 + 1] = 0.f; A.rgb_out[3 * (size_t)i0 + 2] = 0.f; }
                if (g == 1 && v1) { A.rgb_out[3 * (size_t)i1] = 0.f; A.rgb_out[3 * (size_t)i1 + 1] = 0.f; A.rgb_out[3 * (size_t)i1 + 2] = 0.f; }
                if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 1ull);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                pipe_start(P);
                continue;
            }
        }

        if (FULL) {
            hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, g); // bottleneck: no activation on its output (:218)
            Tile D;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const RawIn &ri = pb ? in1 : in0;
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[fb][pb][r] = dir_slot(4 * (2 * fb + gq) + r, hh, ri.dx, ri.dy, ri.dz);
            }
            Tile (&V)[8] = Y;                                            // Y is dead after the bottleneck
            load_bias<4>(V, small + kBiasViewOff, g);
            pin_tile(X[0]);
            prep_all<false>(X[0], ba);
            eight_tiles<4, false>(X, V, ba, bb, P);
            tile_step<4, true, false, false>(V, bb, D, ba, P);
            tile_step<4, false, false, false>(V, ba, D, bb, P);
            // rgb head (f32 VALU) + sigmoid (src/network.rs:222-223, :165)
            float c[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        f32x4 x = V[t][fb][pb];
                        asm volatile("" : "+v"(x));
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const f32x4 w = *(const LDS_AS f32x4 *)(small + kRgbWOff + ch * 128 + 32 * t + 16 * fb + 4 * g);
#pragma unroll
                            for (int r = 0; r < 4; ++r) c[pb][ch] = fmaf(w[r], relu(x[r]), c[pb][ch]);
                        }
                    }
            float o0[3], o1[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                o0[ch] = 1.0f / (1.0f + expf(-(group_sum(c[0][ch]) + small[kMiscOff + 1 + ch])));
                o1[ch] = 1.0f / (1.0f + expf(-(group_sum(c[1][ch]) + small[kMiscOff + 1 + ch])));
            }
            if (g == 0 && v0) { A.rgb_out[3 * (size_t)i0] = o0[0]; A.rgb_out[3 * (size_t)i0 + 1] = o0[1]; A.rgb_out[3 * (size_t)i0 + 2] = o0[2]; }
            if (g == 1 && v1) { A.rgb_out[3 * (size_t)i1] = o1[0]; A.rgb_out[3 * (size_t)i1 + 1] = o1[1]; A.rgb_out[3 * (size_t)i1 + 2] = o1[2]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((nerf_mlp_kernel_bf16x3b<FULL, MODE>), dim3(n_blocks), dim3(256), kLdsBytesX3, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_bf16x3b_init() {
    const void *ks[4] = {(const void *)nerf_mlp_kernel_bf16x3b<true, MLP_MODE_POINTS>, (const void *)nerf_mlp_kernel_bf16x3b<false, MLP_MODE_POINTS>,
                         (const void *)nerf_mlp_kernel_bf16x3b<true, MLP_MODE_RAYS>, (const void *)nerf_mlp_kernel_bf16x3b<false, MLP_MODE_RAYS>};
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesX3);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_mlp_bf16x3b_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_POINTS>(a, n_blocks, stream);
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
