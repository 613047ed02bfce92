// mlp_kernel_f16x2.hip -- the fused NeRF MLP in "f32 by two-way f16 split" arithmetic (mlp_dtype = NERF_MLP_F16X2).
//
// The cheaper sibling of mlp_kernel_bf16x3.hip.  An f32 value is the sum of two f16 numbers up to 2^-22 relative: x = x1 + x2 with
// x1 = f16(x), x2 = f16(x - x1) (the subtraction is exact in f32; f16 has an 11-bit significand).  A product w x is then
//      (w2 x1 + w1 x2) + w1 x1 + O(2^-22 |w x|)
// -- THREE f16 x f16 products, each formed exactly by v_mfma_f32_32x32x16_f16 and accumulated in f32, small terms first; the dropped
// w2 x2 is 2^-22 of the product.  Three 32-cycle MFMAs replace the six of bf16x3 and the eight 64-cycle MFMAs of the f32 kernel
// (5.3 x fewer matrix cycles per f32 FLOP).  What it gives up against bf16x3: two bits of operand precision (22 instead of 24 --
// against fp64 the layer outputs stay at the f32 kernel's error level, because accumulation rounding dominates both) and RANGE:
// f16 overflows at 65 504, so activations must stay below that (the lego networks: |activation| < 300 in the scene; see the domain
// note in include/nerf_mi355x.h); subnormal f16 operands are honoured by the MFMA (MODE.fp16 denormals on, hipcc's default), so
// small values lose nothing beyond the 2^-24 absolute floor of f16.
//
// Structure, stream order, register layouts, small parameters: exactly the bf16x3 kernel's, with two pieces per unit instead of
// three (mlp_layout.h kChunkBytesF16X2): a unit = the (w1, w2) fragments of one 32 x 16 weight block = 2 KiB, a k-step of an 8-tile
// layer = one 16-KiB chunk, each wave DMAs four 1-KiB pieces per chunk behind units 4..7.  Kernel bodies: mlp_split_kernels.hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"
#include "mlp_seq_common.hip.h"

using namespace nerfmlp;
using namespace mlpdev;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCB = kChunkBytesF16X2, kRS = kRingSlotsF16X2;
#ifndef NERF_F16X2_AHEAD
#define NERF_F16X2_AHEAD 2
#endif
constexpr int kAhead = NERF_F16X2_AHEAD; // operand prefetch distance in units (1..3); a unit is three MFMAs = 96 cycles
// Timing-only diagnostics (results are WRONG with any of these set; never shipped): what the non-MFMA cycles are spent on
#ifndef NERF_F16X2_RANGE_WATCH
#define NERF_F16X2_RANGE_WATCH 1 // 0 (variant builds): without the per-pair range watch, to price it
#endif
#ifndef NERF_F16X2_DIAG_NO_BARRIER
#define NERF_F16X2_DIAG_NO_BARRIER 0
#endif
#ifndef NERF_F16X2_DIAG_NO_DMA
#define NERF_F16X2_DIAG_NO_DMA 0
#endif
#ifndef NERF_F16X2_DIAG_NO_PREP
#define NERF_F16X2_DIAG_NO_PREP 0
#endif
#ifndef NERF_F16X2_DIAG_NO_LDS
#define NERF_F16X2_DIAG_NO_LDS 0
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- weight-stream pipeline: 16-KiB chunks of eight 2-KiB units in an LDS ring filled by LDS-DMA, one s_waitcnt + s_barrier per
// chunk at unit 4, operand fragments prefetched kAhead units ahead (the scheme of mlp_x3_pipe.hip.h with two pieces per unit)
struct PipeH {
    const LDS_AS char *rd_base;   // LDS address (incl. lane * 16) of the chunk the next prefetched unit lives in
    const LDS_AS char *ring_lane;
    uint32_t rd_slot_off;
    u32x4 a[8];                   // A fragments (w1, w2) of four consecutive units (slot = unit & 3)
    uint32_t ring_addr, wr_slot_off, next_off, stream_bytes;
    const char *gbase, *cur_src;
    uint32_t cur_dst, lane16;
    float amax = 0.0f;            // largest |value| this lane has split since the last range_check (f16 overflows at 65 504)
};

template <int OFF>
__device__ __forceinline__ void glds_piece_off(uint32_t lane16, const char *gsrc, uint32_t dst) {
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

__device__ __forceinline__ void pipe_next_chunk(PipeH &P) {
    uint32_t off = P.next_off, slot = P.wr_slot_off;
    asm volatile("" : "+s"(off), "+s"(slot));
    P.cur_src = P.gbase + off;
    P.cur_dst = P.ring_addr + slot;
    off += kCB;
    P.next_off = (off == P.stream_bytes) ? 0u : off;
    slot += kCB;
    P.wr_slot_off = (slot == kRS * kCB) ? 0u : slot;
}

// (Re)start at chunk 0 in the state a steady-state run is in there: chunks 0 .. kRS - 2 issued (every wave its four pieces of
// each) and landed, unit 0 .. kAhead - 1 prefetched.  The caller guarantees that no wave still reads the ring.
__device__ __forceinline__ void pipe_start(PipeH &P) {
    P.next_off = 0;
    P.wr_slot_off = 0;
#pragma unroll
    for (int c = 0; c < kRS - 1; ++c) {
        pipe_next_chunk(P);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int s = 0; s < 2 * kAhead; ++s) P.a[s] = *(const LDS_AS u32x4 *)(P.rd_base + s * 1024);
}

// Unit U (0..7 within its chunk) begins: hand out its two A fragments.  Unit 4: the chunk after this one must have landed (every
// wave waits for its own pieces, then the barrier) and the slot of the previous chunk is refilled with chunk c + kRS - 1.
template <int U>
__device__ __forceinline__ void pipe_take(PipeH &P, f16x8 &a1, f16x8 &a2) {
    if constexpr (U == 4) {
#if NERF_F16X2_DIAG_NO_BARRIER
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRS - 3)) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kRS - 3)) : "memory");
#endif
        pipe_next_chunk(P);
    }
    constexpr int cur = (U & 3) * 2;
    a1 = __builtin_bit_cast(f16x8, P.a[cur]); a2 = __builtin_bit_cast(f16x8, P.a[cur + 1]);
}

template <int U>
__device__ __forceinline__ void pipe_prefetch(PipeH &P) {
    constexpr int nxt = ((U + kAhead) & 3) * 2;
    if constexpr (U + kAhead == 8) { // the unit to fetch opens the next chunk
        uint32_t off = P.rd_slot_off + kCB;
        off = (off == kRS * kCB) ? 0u : off;
        P.rd_slot_off = off;
        P.rd_base = P.ring_lane + off;
    }
    constexpr int nu = (U + kAhead) & 7;
#if NERF_F16X2_DIAG_NO_LDS
    (void)nu; // keep whatever the fragments hold
#else
#pragma unroll
    for (int s = 0; s < 2; ++s) P.a[nxt + s] = *(const LDS_AS u32x4 *)(P.rd_base + (2 * nu + s) * 1024);
#endif
}

// the LDS-DMA piece issued behind unit U: the four pieces of the chunk selected at the last sync go out behind units 4..7
template <int U>
__device__ __forceinline__ void pipe_dma(PipeH &P) {
#if !NERF_F16X2_DIAG_NO_DMA
    if constexpr (U >= 4) glds_piece_off<(U - 4) * 1024>(P.lane16, P.cur_src, P.cur_dst);
#endif
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct B2 { u32x4 h, l; }; // the two f16x8 fragments of one k-step's B operand

// Splitting one pair of f32 values into packed (h, l) f16 pairs, in three stages of a few VALU instructions, one per MFMA gap.
// x - f16(x) is exact in f32 (at most 13 significant bits are left).
struct PrepState { f32x2 x; uint32_t h; };

template <bool RELU, int KS, int Q, int STAGE>
__device__ __forceinline__ void prep_stage(const f32x16 &in, B2 &b, PrepState &st, float &amax) {
#if NERF_F16X2_DIAG_NO_PREP
    if constexpr (STAGE == 2) { b.h[Q] = 0x3c003c00u; b.l[Q] = 0u; }
    return;
#endif
    if constexpr (STAGE == 0) {
        float x0 = in[8 * KS + 2 * Q], x1 = in[8 * KS + 2 * Q + 1];
        asm volatile("" : "+v"(x0), "+v"(x1)); // keep the accumulator reads here (hipcc otherwise hoists a whole layer's)
        if (RELU) { x0 = relu(x0); x1 = relu(x1); }
        // Range watch (one v_max3_f32 per pair): beyond 65 504 the f16 parts become (inf, -inf), the products NaN, and a NaN does not
        // survive the integer-max ReLU of the next layer -- the outputs would be finite and WRONG (measured: sigma 296 instead of 1.5e7)
#if NERF_F16X2_RANGE_WATCH
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x0), "v"(x1)); // as asm: the plain expression made hipcc spill 800 registers
#endif
        st.x = f32x2{x0, x1};
    } else if constexpr (STAGE == 1) {
        const f16x2 hh = __builtin_convertvector(st.x, f16x2);
        st.h = __builtin_bit_cast(uint32_t, hh);
        st.x = st.x - __builtin_convertvector(hh, f32x2);
    } else {
        b.h[Q] = st.h;
        b.l[Q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, f16x2));
    }
}

template <bool RELU, int KS, int Q>
__device__ __forceinline__ void prep_pair(const f32x16 &in, B2 &b, float &amax) {
    PrepState st;
    prep_stage<RELU, KS, Q, 0>(in, b, st, amax); prep_stage<RELU, KS, Q, 1>(in, b, st, amax); prep_stage<RELU, KS, Q, 2>(in, b, st, amax);
}

template <bool RELU, int KS>
__device__ __forceinline__ void prep_all(const f32x16 &in, B2 &b, PipeH &P) {
    prep_pair<RELU, KS, 0>(in, b, P.amax); prep_pair<RELU, KS, 1>(in, b, P.amax); prep_pair<RELU, KS, 2>(in, b, P.amax); prep_pair<RELU, KS, 3>(in, b, P.amax);
}

// End of a tile / chunk: did any operand of this column leave the f16 range?  Counts columns (both lane-halves of a column hold
// features of the same point) into the optional device counter and re-arms the watch.
__device__ __forceinline__ void range_check(PipeH &P, unsigned int *counter, bool valid) {
    unsigned long long bad = __ballot(valid && !(P.amax <= 65504.0f)); // NaN counts as out of range
    P.amax = 0.0f;
    bad = (bad | (bad >> 32)) & 0xffffffffull;
    if (counter && bad && (threadIdx.x & 63) == 0) atomicAdd(counter, (unsigned)__popcll(bad));
}

#define X2_PIN() __builtin_amdgcn_sched_barrier(0)

// One k-step: NT units with the prepared B `bc`.  A unit = the three products of one 32 x 16 weight block, small terms first, on
// one accumulator chain; the other work rides in its three 32-cycle MFMA gaps, at most ~5 single-issue instructions each (the
// guide's rule): the next unit's two ds_reads behind MFMA 1; the three stages of splitting one pair of the NEXT k-step's B operand
// spread over a PAIR of units (8-tile layers: pair q in units 2q, 2q+1 -- stages 0, 1 behind MFMAs 2, 3 of the even unit, stage 2
// behind MFMA 2 of the odd one; viewdirs, 4 units per k-step: all three stages in unit q); the LDS-DMA piece of units 4..7 behind
// MFMA 1 (even units) or MFMA 3 (odd units, whose third gap is otherwise empty).
#ifndef NERF_F16X2_SPREAD
#define NERF_F16X2_SPREAD 1 // 0: everything behind MFMA 1 / per-unit stages (first version, 56.9 % of the bf16-class peak)
#endif
template <int NT, int U0, bool HAS_NEXT, bool NRELU, int NKS>
__device__ __forceinline__ void k_step(f32x16 (&out)[8], const B2 &bc, const f32x16 &nin, B2 &bn, PipeH &P) {
    const f16x8 b1 = __builtin_bit_cast(f16x8, bc.h), b2 = __builtin_bit_cast(f16x8, bc.l);
    PrepState st; // carried from the even to the odd unit of a pair
    static_for<0, NT>([&](auto nt_c) {
        constexpr int nt = decltype(nt_c)::value;
        constexpr int U = U0 + nt;
        constexpr bool pairwise = NERF_F16X2_SPREAD && NT == 8;
        constexpr bool even = (nt & 1) == 0;
        constexpr int Q = NT == 4 ? nt : nt / 2;
        f16x8 a1, a2;
        pipe_take<U>(P, a1, a2);
        X2_PIN();
        out[nt] = MFMA16(a2, b1, out[nt]);
        X2_PIN();
        pipe_prefetch<U>(P);
        if constexpr (!pairwise || even) pipe_dma<U>(P);
        if constexpr (HAS_NEXT && !pairwise && (NT == 4 || even)) prep_stage<NRELU, NKS, Q, 0>(nin, bn, st, P.amax);
        X2_PIN();
        out[nt] = MFMA16(a1, b2, out[nt]);
        X2_PIN();
        if constexpr (HAS_NEXT && pairwise) prep_stage<NRELU, NKS, Q, (even ? 0 : 2)>(nin, bn, st, P.amax);
        if constexpr (HAS_NEXT && !pairwise && (NT == 4 || even)) prep_stage<NRELU, NKS, Q, 1>(nin, bn, st, P.amax);
        X2_PIN();
        out[nt] = MFMA16(a1, b1, out[nt]);
        X2_PIN();
        if constexpr (HAS_NEXT && pairwise && even) prep_stage<NRELU, NKS, Q, 1>(nin, bn, st, P.amax);
        if constexpr (pairwise && !even) pipe_dma<U>(P);
        if constexpr (HAS_NEXT && !pairwise && (NT == 4 || even)) prep_stage<NRELU, NKS, Q, 2>(nin, bn, st, P.amax);
        X2_PIN();
    });
}

using PipeS = PipeH;
using BS = B2;
constexpr int kSplitChunksSigma = kChunksSigmaF16X2, kSplitChunksFull = kChunksFullF16X2, kSplitLdsBytes = kLdsBytesF16X2;
constexpr int kSplitWaveBytes = 4096; // a wave DMAs four 1-KiB pieces of every 16-KiB chunk

} // namespace

#define SPLIT_KERNEL_FUSED nerf_mlp_kernel_f16x2
#define SPLIT_KERNEL_TRUNK nerf_trunk_seq_kernel_f16x2
#define SPLIT_KERNEL_COLOUR nerf_colour_kernel_f16x2
#define SPLIT_FN_INIT nerf_mlp_f16x2_init
#define SPLIT_FN_LAUNCH nerf_mlp_f16x2_launch
#define SPLIT_FN_SEQ_INIT nerf_seq_f16x2_init
#define SPLIT_FN_TRUNK_LAUNCH nerf_trunk_seq_f16x2_launch
#define SPLIT_FN_COLOUR_LAUNCH nerf_colour_f16x2_launch
#include "mlp_split_kernels.hip.h"
