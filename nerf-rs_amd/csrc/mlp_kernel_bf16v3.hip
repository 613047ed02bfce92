// mlp_kernel_bf16v3.hip -- bf16 MLP on v_mfma_f32_16x16x32_bf16 (BASELINE config C5 study, third design).
//
// Why another shape.  The 16-bit matrix cores of this chip are POWER-limited: a bare v_mfma_f32_32x32x16_bf16 loop on random operands
// (operands in registers, nothing else in the loop) delivers 0.72-0.73 of the nominal 2.5 PFLOP/s at a 1.80 GHz clock, the
// 16x16x32 shape 0.82 at 2.08 GHz (tools/probes/mfma_power_probe.hip, profiles/r03_power_probe.jsonl; MI355X_MICROARCH.md "DVFS
// give-back" item 7).  mlp_kernel_bf16v2.hip (32x32x16) sits at 0.60-0.61 = 83 % of what its shape can deliver at all; the only
// lever left is the shape.  This kernel is v2's design -- output-tile-major, 64 points per wave, packed bf16 activations in
// registers, f32 accumulate, f32 heads, the same LDS-DMA weight ring -- re-tiled for the 16x16x32 instruction:
//   * a wave's 64 points are four COLUMN GROUPS of 16 (the MFMA's N); lane l = (j = l & 15, g = l >> 4) serves point 16 cg + j of
//     every group and holds the k-slice 8 g .. 8 g + 7 of each 32-wide k-step (B operand) / features 4 g .. 4 g + 3 of each
//     16-feature output half (C/D);
//   * an output tile is 32 features = two 16-feature halves x four column groups = eight f32x4 accumulators; per k-step (K = 32)
//     two 1-KiB A pieces (one per half) feed eight MFMAs -- the same bytes per FLOP through LDS-DMA and ds_read as v2;
//   * a finished output tile, ReLU'd and packed pairwise (v_cvt_pk_bf16_f32 + v_pk_max_i16), IS k-step nt of the next layer: k-slot
//     (g, e) of a k-step holds feature 4 g + e (e < 4) or 16 + 4 g + e - 4 (e >= 4) of that tile -- the host permutes the weight
//     rows accordingly (nerf_api.cpp bf16_v3_from_v1), so activations never leave registers;
//   * encodings: lane L computes ALL features of point L (angle doubling from three base octaves), packs them to bf16 in the
//     reference's feature order and transposes them into B fragments through 8 KiB of LDS per wave (wave-private, no barrier).
// Stream: per layer, per output tile, per k-step, per half one 1-KiB piece; 16-KiB chunks; exactly v2's piece counts (dense0 32,
// hidden 128, dense5 160, viewdirs 72 + 8 pad), so the chunk constants of mlp_layout.h (kChunks*Bf16V2) are shared.
// Arithmetic = the oracle's bf16 emulation (bf16 weights and layer inputs, RNE; f32 accumulate; f32 biases, heads, sigmoid).
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
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef NERF_BV3_AHEAD
#define NERF_BV3_AHEAD 8 // A-operand prefetch distance in pieces (even, < NERF_BV3_RING - 1, and <= 8: a piece of the NEXT chunk may only be read
                         // behind the mid-chunk sync that proves it landed).  Measured, full kernel: 2 -> 103.1 ms, 4 -> 104.2, 6 -> 100.0, 8 -> 96.9
#endif
#ifndef NERF_BV3_RING
#define NERF_BV3_RING 16 // register ring of prefetched A operands (8 or 16: must divide the 16 pieces of a chunk)
#endif
#ifndef NERF_BV3_SCHED_BARRIER
#define NERF_BV3_SCHED_BARRIER 1
#endif
// timing-only diagnostics (results are garbage): which resource bounds the kernel
#ifndef NERF_BV3_DIAG_NO_DMA
#define NERF_BV3_DIAG_NO_DMA 0
#endif
#ifndef NERF_BV3_DIAG_NO_BARRIER
#define NERF_BV3_DIAG_NO_BARRIER 0
#endif
#ifndef NERF_BV3_DIAG_NO_EPILOGUE
#define NERF_BV3_DIAG_NO_EPILOGUE 0
#endif
#ifndef NERF_BV3_DIAG_NO_LDS
#define NERF_BV3_DIAG_NO_LDS 0
#endif

namespace {

constexpr int kCB = kChunkBytesBf16V2, kRS = kRingSlotsBf16V2;
constexpr int kAhead = NERF_BV3_AHEAD;
constexpr int kEncBytesPerWave = 8192;                       // transposition area of the encodings: 64 points x 64 slots x bf16
constexpr int kEncOff = kRS * kCB + kSmallBytes;
constexpr int kLdsBytesV3 = kEncOff + 4 * kEncBytesPerWave;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- weight ring: v2's pipeline (16-KiB chunks of 16 pieces; sync at piece 8; one DMA piece per wave at pieces 9, 11, 13, 15) ----
struct PipeV {
    const LDS_AS char *rd_base;
    const LDS_AS char *ring_lane;
    uint32_t rd_slot_off;
    u32x4 a[NERF_BV3_RING];
    uint32_t ring_addr, wr_slot_off, next_off, stream_bytes;
    const char *gbase, *cur_src;
    uint32_t cur_dst, lane16;
};

__device__ __forceinline__ void pipe_next_chunk(PipeV &P) {
    uint32_t off = P.next_off, slot = P.wr_slot_off;
    asm volatile("" : "+s"(off), "+s"(slot));
    P.cur_src = P.gbase + off;
    P.cur_dst = P.ring_addr + slot;
    off += kCB;
    P.next_off = (off == P.stream_bytes) ? 0u : off;
    slot += kCB;
    P.wr_slot_off = (slot == kRS * kCB) ? 0u : slot;
#if NERF_M0_PER_CHUNK
    dma_set_dst(P.cur_dst);
#endif
}

__device__ __forceinline__ void pipe_start(PipeV &P) {
    P.next_off = 0;
    P.wr_slot_off = 0;
#pragma unroll
    for (int c = 0; c < kRS - 1; ++c) {
        pipe_next_chunk(P);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if NERF_M0_PER_CHUNK
            glds_piece_m0(P.lane16, P.cur_src, i);
#else
            glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
#endif
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int j = 0; j < kAhead; ++j) P.a[j] = *(const LDS_AS u32x4 *)(P.rd_base + j * 1024);
}

template <int PH>
__device__ __forceinline__ bf16x8 pipe_take(PipeV &P) {
    if constexpr (PH == 8) {
#if NERF_BV3_DIAG_NO_BARRIER
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRS - 3)) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kRS - 3)) : "memory");
#endif
        pipe_next_chunk(P);
    }
#if NERF_BV3_DIAG_NO_LDS
    if constexpr (PH != 0) { u32x4 a = P.a[0]; asm volatile("" : "+v"(a)); return __builtin_bit_cast(bf16x8, a); }
#endif
    static_assert(kAhead % 2 == 0, "paired reads need an even prefetch distance");
    if constexpr ((PH & 1) == 0) {
        constexpr int R = NERF_BV3_RING;
        static_assert((R == 8 || R == 16) && kAhead + 1 < R && kAhead <= 8, "ring");
        asm volatile("" : "+v"(P.a[PH % R]), "+v"(P.a[(PH + 1) % R])); // one s_waitcnt for the pair
        const u32x4 a = P.a[PH % R];
        if constexpr (PH + kAhead == 16) { // the pair to prefetch opens the next chunk (kAhead <= 14: exactly one even phase per chunk does)
            uint32_t off = P.rd_slot_off + kCB;
            off = (off == kRS * kCB) ? 0u : off;
            P.rd_slot_off = off;
            P.rd_base = P.ring_lane + off;
        }
        P.a[(PH + kAhead) % R] = *(const LDS_AS u32x4 *)(P.rd_base + ((PH + kAhead) % 16) * 1024);
        P.a[(PH + kAhead + 1) % R] = *(const LDS_AS u32x4 *)(P.rd_base + ((PH + kAhead + 1) % 16) * 1024);
        return __builtin_bit_cast(bf16x8, a);
    } else {
        return __builtin_bit_cast(bf16x8, P.a[PH % NERF_BV3_RING]);
    }
}

template <int OFF>
__device__ __forceinline__ void glds_piece_off(uint32_t lane16, const char *gsrc, uint32_t dst) {
#if NERF_M0_PER_CHUNK
    (void)dst; // M0 holds it since pipe_next_chunk
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane16), "s"(gsrc), "n"(OFF) : "memory");
    return;
#endif
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane16), "s"(gsrc), "s"(dst), "n"(OFF)
                 : "memory");
}

template <int PH>
__device__ __forceinline__ void pipe_dma(PipeV &P) {
#if NERF_BV3_DIAG_NO_DMA
    return;
#endif
    if constexpr (PH >= 9 && (PH & 1) == 1) glds_piece_off<((PH - 9) / 2) * 1024>(P.lane16, P.cur_src, P.cur_dst);
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <bool RELU>
__device__ __forceinline__ uint32_t pack2(float x, float y) {
    const f32x2 p = {x, y};
    bf16x2 c = __builtin_convertvector(p, bf16x2);
    if (RELU) {
        s16x2 i = __builtin_bit_cast(s16x2, c);
        i = __builtin_elementwise_max(i, (s16x2){0, 0});
        return __builtin_bit_cast(uint32_t, i);
    }
    return __builtin_bit_cast(uint32_t, c);
}

struct Bk { u32x4 c[4]; };      // one k-step (32 features) of the packed activations: this lane's 8 k-values for each column group
struct AccT { f32x4 v[2][4]; }; // one output tile: [16-feature half][column group]: features 16 fh + 4 g + r of point 16 cg + j

struct Heads {
    float alpha[4];   // per column group: partial sums over this lane's features
    float rgb[4][3];
};

// per-lane offsets into the small-parameter block: it is laid out for the 32x32 kernels ([nt][h][16] with feature
// 32 nt + (r & 3) + 8 (r >> 2) + 4 h); this lane's features 16 fh + 4 g + (0..3) of tile nt are the four consecutive floats at
// h' = g & 1, r' = 4 (2 fh + (g >> 1)): one f32x4 / two f32x2
struct LaneOfs { int bias, alpha, rgb; };

// Epilogue of pair PR (0..15) of a finished output tile: PR = 4 cg + q; q < 2: half 0, registers 2 q, 2 q + 1; q >= 2: half 1.
// HEAD 0: convert only; 1: convert + alpha partial sums; 2: alpha only (sigma kernels); 3: rgb partial sums only.
template <int PR, bool RELU, int HEAD, int NTI>
__device__ __forceinline__ void convert_pair(const AccT &acc, Bk &dst, Heads &H, const LDS_AS float *small, const LaneOfs &L) {
#if NERF_BV3_DIAG_NO_EPILOGUE
    if constexpr (PR != 0) return;
#endif
    constexpr int cg = PR >> 2, q = PR & 3, fh = q >> 1, r0 = 2 * (q & 1);
    const float x0 = acc.v[fh][cg][r0], x1 = acc.v[fh][cg][r0 + 1];
    if constexpr (HEAD == 1 || HEAD == 2) { // alpha = sum_F w[F] relu(h8[F]) in f32 (src/network.rs:216)
        const f32x2 w = *(const LDS_AS f32x2 *)(small + kAlphaWOff + L.alpha + NTI * 16 + 8 * fh + r0);
        H.alpha[cg] = fmaf(w[1], relu(x1), fmaf(w[0], relu(x0), H.alpha[cg]));
    }
    if constexpr (HEAD == 3) { // rgb pre-activations from relu(viewdirs) in f32 (src/network.rs:222-223)
        const float a0 = relu(x0), a1 = relu(x1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x2 w = *(const LDS_AS f32x2 *)(small + kRgbWOff + L.rgb + c * 64 + NTI * 16 + 8 * fh + r0);
            H.rgb[cg][c] = fmaf(w[1], a1, fmaf(w[0], a0, H.rgb[cg][c]));
        }
    }
    if constexpr (HEAD == 0 || HEAD == 1) dst.c[cg][q] = pack2<RELU>(x0, x1);
}

__device__ __forceinline__ void load_bias(f32x4 (&b)[2], const LDS_AS float *bias, int nt, const LaneOfs &L) {
    b[0] = *(const LDS_AS f32x4 *)(bias + nt * 32 + L.bias);
    b[1] = *(const LDS_AS f32x4 *)(bias + nt * 32 + L.bias + 8);
}

#if NERF_BV3_SCHED_BARRIER
#define BV3_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define BV3_PIN() ((void)0)
#endif

// MFMA slot (0 .. 8 N_STEPS - 1) of the NEXT tile behind which pair pr of a finished tile is converted: spread evenly, not before
// slot 2 (reading an accumulator right behind the MFMA that finished it costs the MFMA-write -> VALU-read wait states)
template <int N_STEPS>
constexpr int pair_slot(int pr) { return 2 + pr * (8 * N_STEPS - 3) / 16; }

// One output tile NTI of a layer: KS k-steps (two pieces each) from stream phase PH, 8 MFMAs per k-step.  `bv` holds this tile's
// bias (loaded one tile ahead) and is the C operand of the first MFMA of every chain.  Under the MFMAs runs the epilogue of the
// previously finished tile `p` into `e`, its 16 pairs spread over the first EP_STEPS k-steps:
//   EP = 1: the previous tile of this layer -> out[NTI - 1];
//   EP = 2: the LAST tile of the previous layer, whose packed result is k-step PIDX of this layer's own input: written into
//           in[PIDX] before the MFMAs of step PIDX read it (EP_STEPS <= PIDX).
template <int KS, int NT, int NTI, int PH, int EP, bool EP_RELU, int EP_HEAD, int EP_NTI, int EP_STEPS>
__device__ __forceinline__ void out_tile(const Bk (&in)[KS], Bk &e, AccT &c, const AccT &p, f32x4 (&bv)[2], const LDS_AS float *bias,
                                         const LDS_AS float *small, Heads &H, PipeV &P, const LaneOfs &L) {
    static_for<0, KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        static_for<0, 2>([&](auto fh_c) {
            constexpr int fh = decltype(fh_c)::value;
            constexpr int ph = (PH + 2 * ks + fh) % 16;
            const bf16x8 a = pipe_take<ph>(P);
            static_for<0, 4>([&](auto cg_c) {
                constexpr int cg = decltype(cg_c)::value;
                constexpr int slot = 8 * ks + 4 * fh + cg;
                if constexpr (ks == 0) c.v[fh][cg] = MFMA32(a, __builtin_bit_cast(bf16x8, in[ks].c[cg]), bv[fh]);
                else                   c.v[fh][cg] = MFMA32(a, __builtin_bit_cast(bf16x8, in[ks].c[cg]), c.v[fh][cg]);
                BV3_PIN();
                if constexpr (EP != 0)
                    static_for<0, 16>([&](auto pr_c) {
                        constexpr int pr = decltype(pr_c)::value;
                        if constexpr (pair_slot<EP_STEPS>(pr) == slot) convert_pair<pr, EP_RELU, EP_HEAD, EP_NTI>(p, e, H, small, L);
                    });
                if constexpr (cg == 1) pipe_dma<ph>(P);
                if constexpr (ks == 0 && fh == 1 && cg == 3 && NTI + 1 < NT) load_bias(bv, bias, NTI + 1, L); // every chain has consumed bv
                BV3_PIN();
            });
        });
    });
}

// One layer: NT (even) output tiles alternating two accumulator sets (even tiles set a, odd tiles set b).
//   PEND:  the previous layer deferred its last tile's epilogue (in set b): it runs under this layer's tile 0 and lands in in[PIDX].
//   DEFER: leave this layer's last tile to the next layer in the same way instead of a tail burst (not for the head layers).
template <int KS, int NT, bool RELU_OUT, int HEAD, bool PEND, bool PEND_RELU, int PIDX, bool DEFER>
__device__ __forceinline__ void layer(Bk (&in)[KS], Bk (&out)[8], AccT &Ca, AccT &Cb, const LDS_AS float *bias, const LDS_AS float *small,
                                      Heads &H, PipeV &P, const LaneOfs &L) {
    static_assert(NT % 2 == 0, "layers alternate two accumulator sets and must end on set b");
    static_assert(!DEFER || HEAD == 0, "a deferred epilogue only converts");
    f32x4 bv[2];
    load_bias(bv, bias, 0, L);
    static_for<0, NT>([&](auto nt_c) {
        constexpr int nt = decltype(nt_c)::value;
        constexpr int ph = (nt * KS * 2) % 16;
        constexpr int o = nt > 0 ? nt - 1 : 0;
        if constexpr (nt == 0) {
            if constexpr (PEND) out_tile<KS, NT, 0, ph, 2, PEND_RELU, 0, 0, PIDX>(in, in[PIDX], Ca, Cb, bv, bias, small, H, P, L);
            else                out_tile<KS, NT, 0, ph, 0, false, 0, 0, 1>(in, out[0], Ca, Cb, bv, bias, small, H, P, L);
        } else if constexpr ((nt & 1) == 0) {
            out_tile<KS, NT, nt, ph, 1, RELU_OUT, HEAD, nt - 1, KS>(in, out[o], Ca, Cb, bv, bias, small, H, P, L);
        } else {
            out_tile<KS, NT, nt, ph, 1, RELU_OUT, HEAD, nt - 1, KS>(in, out[o], Cb, Ca, bv, bias, small, H, P, L);
        }
    });
    if constexpr (!DEFER)
        static_for<0, 16>([&](auto pr_c) { convert_pair<decltype(pr_c)::value, RELU_OUT, HEAD, NT - 1>(Cb, out[NT - 1], H, small, L); });
}

// ---- encodings: lane L computes every feature of point L in the reference's order (src/network.rs:263-330), by angle doubling
// from an accurate sincos every four octaves (error <= 2e-6, far below the bf16 rounding that follows), and the wave transposes the
// packed values into B fragments through its private LDS area: fragment (ks, cg) of lane (j, g) = slots 32 ks + 8 g .. + 7 of
// point 16 cg + j, stored at [(ks * 4 + cg) * 64 + g * 16 + j] x 16 B (lane-linear, conflict-free on the read side).
template <int N_OCT, int N_SLOTS>
__device__ __forceinline__ void encode_ref_order(float px, float py, float pz, float (&E)[N_SLOTS]) {
    E[0] = px; E[1] = py; E[2] = pz;
    const float pc[3] = {px, py, pz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f, c = 0.f;
#pragma unroll
        for (int o = 0; o < N_OCT; ++o) {
            if ((o & 3) == 0) fast_sincos((float)(1 << o) * pc[k], &s, &c);
            else { const float s2 = (s + s) * c; c = fmaf(-2.0f * s, s, 1.0f); s = s2; }
            E[3 + 6 * o + k] = s;
            E[3 + 6 * o + 3 + k] = c;
        }
    }
#pragma unroll
    for (int i = 3 + 6 * N_OCT; i < N_SLOTS; ++i) E[i] = 0.f;
}

template <int KS>
__device__ __forceinline__ void transpose_to_fragments(const float (&E)[32 * KS], Bk (&out)[KS], LDS_AS char *area, int lane) {
    const int cg = lane >> 4, j = lane & 15;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // earlier reads of the area (previous encoding) have returned
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = pack2<false>(E[32 * ks + 8 * g + 2 * q], E[32 * ks + 8 * g + 2 * q + 1]);
            *(LDS_AS u32x4 *)(area + ((ks * 4 + cg) * 64 + g * 16 + j) * 16) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private area: the wave's own writes are done (no barrier needed)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[ks].c[c] = *(const LDS_AS u32x4 *)(area + ((ks * 4 + c) * 64 + lane) * 16);
}

} // namespace

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void nerf_mlp_kernel_bf16v3(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    LaneOfs L;
    L.bias = (g & 1) * 16 + (g >> 1) * 4;
    L.alpha = (g & 1) * 128 + (g >> 1) * 4;
    L.rgb = (g & 1) * 192 + (g >> 1) * 4;
    LDS_AS char *enc_area = (LDS_AS char *)smem + kEncOff + wave * kEncBytesPerWave;

    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    PipeV P;
    P.lane16 = lane * 16;
    P.ring_lane = lds + P.lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 4096;
    P.stream_bytes = (FULL ? kChunksFullBf16V2 : kChunksSigmaBf16V2) * kCB;
    P.gbase = (const char *)A.wstream + wave * 4096;
    __syncthreads();
    pipe_start(P);
    uint64_t clk0 = 0, rt0 = 0;
    if (A.clock_out) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    const int n_tiles = (A.n_points + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2;
    auto raw = [&](int tile_idx) -> RawIn { // lane L <-> point L of the wave's 64; clamped: padding lanes and the look-ahead tile read the last point
        RawIn r;
        int i = tile_idx * kPointsPerBlockBf16V2 + wave * 64 + lane;
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
    RawIn nxt = raw(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * kPointsPerBlockBf16V2 + wave * 64; // this wave's first point
        const RawIn in = nxt;
        nxt = raw(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);

        Bk E[2]; // position encoding: 64 slots (63 features) = 2 k-steps
        {
            float px, py, pz;
            point_of<MODE>(A, in, px, py, pz);
            float e[64];
            encode_ref_order<10, 64>(px, py, pz, e);
            transpose_to_fragments<2>(e, E, enc_area, lane);
        }
        Bk X[8], Y[8];
        AccT Ca, Cb;
        Heads H;
#pragma unroll
        for (int c = 0; c < 4; ++c) { H.alpha[c] = 0.f; H.rgb[c][0] = H.rgb[c][1] = H.rgb[c][2] = 0.f; }

        layer<2, 8, true, 0, false, false, 0, true>(E, X, Ca, Cb, small + kBiasOff + 0 * 256, small, H, P, L);      // dense0 (src/network.rs:204)
        layer<8, 8, true, 0, true, true, 7, true>(X, Y, Ca, Cb, small + kBiasOff + 1 * 256, small, H, P, L);
        layer<8, 8, true, 0, true, true, 7, true>(Y, X, Ca, Cb, small + kBiasOff + 2 * 256, small, H, P, L);
        layer<8, 8, true, 0, true, true, 7, true>(X, Y, Ca, Cb, small + kBiasOff + 3 * 256, small, H, P, L);
        layer<8, 8, true, 0, true, true, 7, true>(Y, X, Ca, Cb, small + kBiasOff + 4 * 256, small, H, P, L);
        {   // dense5 on [encoding (2 k-steps) ; h4 (8 k-steps)] (src/network.rs:209-210)
            Bk C5[10];
            C5[0] = E[0]; C5[1] = E[1];
#pragma unroll
            for (int k = 0; k < 8; ++k) C5[2 + k] = X[k];
            layer<10, 8, true, 0, true, true, 9, true>(C5, Y, Ca, Cb, small + kBiasOff + 5 * 256, small, H, P, L); // h4's last tile lands in C5[9]
        }
        layer<8, 8, true, 0, true, true, 7, true>(Y, X, Ca, Cb, small + kBiasOff + 6 * 256, small, H, P, L);
        // dense7: alpha head from the f32 accumulators; the packed h8 is only needed when the colour branch follows
        layer<8, 8, true, FULL ? 1 : 2, true, true, 7, false>(X, Y, Ca, Cb, small + kBiasOff + 7 * 256, small, H, P, L);
        float sg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { // sum over the four lane groups, + bias, ReLU (src/network.rs:216)
            float v = H.alpha[c];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            sg[c] = fmaxf(v + small[kMiscOff + 0], 0.f);
        }
        bool any_density = false;
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = base + 16 * c + j;
                if (i < A.n_points) { A.sigma_out[i] = sg[c]; any_density = any_density || sg[c] > 0.0f; }
            }
        }
        if constexpr (FULL) {
            if (A.skip_empty) { // exact empty-tile skip, see mlp_kernel.hip: all 256 sigmas are 0 -> colours are never used
                LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
                const bool any_wg = tile_has_density(vote, any_density, wave, lane);
                if (!any_wg) {
                    if (g == 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int i = base + 16 * c + j;
                            if (i < A.n_points) { A.rgb_out[3 * (size_t)i] = 0.f; A.rgb_out[3 * (size_t)i + 1] = 0.f; A.rgb_out[3 * (size_t)i + 2] = 0.f; }
                        }
                    }
                    if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 2ull); // counter unit = 128 points
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // in-flight chunks landed; ring idle
                    pipe_start(P);
                    continue;
                }
            }
            layer<8, 8, false, 0, false, false, 0, true>(Y, X, Ca, Cb, small + kBiasOff + 8 * 256, small, H, P, L); // bottleneck: no activation (:218)
            Bk V[9];
#pragma unroll
            for (int k = 0; k < 8; ++k) V[k] = X[k];
            {
                float d[32];
                encode_ref_order<4, 32>(in.dx, in.dy, in.dz, d);
                Bk D[1];
                transpose_to_fragments<1>(d, D, enc_area, lane);
                V[8] = D[0];
            }
            layer<9, 4, true, 3, true, false, 7, false>(V, Y, Ca, Cb, small + kBiasViewOff, small, H, P, L); // viewdirs + rgb partial sums (:220-223)
            {   // viewdirs' 72 pieces end at phase 8; the stream carries 8 zero pieces up to the chunk end: step over them
                bf16x8 d;
                d = pipe_take<8>(P);                   asm volatile("" ::"v"(d));
                d = pipe_take<9>(P);  pipe_dma<9>(P);  asm volatile("" ::"v"(d));
                d = pipe_take<10>(P);                  asm volatile("" ::"v"(d));
                d = pipe_take<11>(P); pipe_dma<11>(P); asm volatile("" ::"v"(d));
                d = pipe_take<12>(P);                  asm volatile("" ::"v"(d));
                d = pipe_take<13>(P); pipe_dma<13>(P); asm volatile("" ::"v"(d));
                d = pipe_take<14>(P);                  asm volatile("" ::"v"(d));
                d = pipe_take<15>(P); pipe_dma<15>(P); asm volatile("" ::"v"(d));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float col[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { // sigmoid (src/network.rs:165)
                    float v = H.rgb[c][ch];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    col[ch] = 1.0f / (1.0f + expf(-(v + small[kMiscOff + 1 + ch])));
                }
                const int i = base + 16 * c + j;
                if (g == 0 && i < A.n_points) { A.rgb_out[3 * (size_t)i] = col[0]; A.rgb_out[3 * (size_t)i + 1] = col[1]; A.rgb_out[3 * (size_t)i + 2] = col[2]; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA may be in flight when the workgroup's LDS is released
    if (A.clock_out && tid == 0) { // diagnostic: shader clock = d(memtime) / d(memrealtime) x 100 MHz
        A.clock_out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        A.clock_out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((nerf_mlp_kernel_bf16v3<FULL, MODE>), dim3(n_blocks), dim3(256), kLdsBytesV3, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_bf16v3_init() {
    const void *ks[3] = {(const void *)nerf_mlp_kernel_bf16v3<true, MLP_MODE_POINTS>, (const void *)nerf_mlp_kernel_bf16v3<true, MLP_MODE_RAYS>,
                         (const void *)nerf_mlp_kernel_bf16v3<false, MLP_MODE_RAYS>};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesV3);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_mlp_bf16v3_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : hipErrorInvalidValue; // no sigma-only forward_batch
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
