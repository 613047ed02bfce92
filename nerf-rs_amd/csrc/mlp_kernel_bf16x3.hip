// mlp_kernel_bf16x3.hip -- the fused NeRF MLP in "f32 by three-way bf16 split" arithmetic (mlp_dtype = NERF_MLP_BF16X3).
//
// An f32 value is the exact sum of three bf16 numbers up to 2^-27 relative: x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (every subtraction is exact in f32).  A product w x is then
//      w1 x1 + (w1 x2 + w2 x1) + (w1 x3 + w2 x2 + w3 x1) + O(2^-26 |w x|)
// -- six bf16 x bf16 products, each formed EXACTLY by v_mfma_f32_32x32x16_bf16 and accumulated in f32.  The dropped terms are
// below the f32 rounding of the product itself, so the layer arithmetic is f32-accurate (measured against the f32 oracle in
// tests/test_gpu_parity.py with the f32 path's own tolerances), while the matrix work runs on the bf16 cores: six 32-cycle
// MFMAs replace eight 64-cycle f32 MFMAs (2.67 x fewer matrix cycles per f32 FLOP).  Opt-in; the default f32 path
// (mlp_kernel.hip) stays the reference-order kernel.
//
// Structure = the f32 / first bf16 kernel: one persistent 256-thread workgroup per CU, 32 points per wave, activations stay
// in registers as f32 accumulator tiles (X/Y ping-pong, 128 + 128 registers); k-major loop.  Per k-step (K = 16) the B operand
// is split on the fly: 8 f32 values per lane -> ReLU -> three packed bf16x8 fragments (about 60 VALU, prepared one k-step
// ahead, spread behind the MFMAs of the current k-step).  Weights are split on the host.
// Stream (mlp_layout.h): per layer, input tile, k-step and output tile one UNIT = three 1-KiB pieces (w1, w2, w3 fragments of
// the same 32 x 16 block); a k-step of an 8-tile layer = 24 pieces = one 24-KiB chunk; ring of kRingSlotsX3 slots; sync at unit 4 of a chunk;
// each wave DMA's its six pieces of the chunk after next behind units 4, 5, 6, 7, 0, 1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"
#include "mlp_seq_common.hip.h"

using namespace nerfmlp;
using namespace mlpdev;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCB = kChunkBytesX3, kRS = kRingSlotsX3;
#ifndef NERF_X3_AHEAD
#define NERF_X3_AHEAD 2
#endif
constexpr int kX3Ahead = NERF_X3_AHEAD; // operand prefetch distance in units (1..3); a unit is six MFMAs = 192 cycles

#include "mlp_x3_pipe.hip.h"

#ifndef NERF_X3_DIAG_DROP_W3
#define NERF_X3_DIAG_DROP_W3 0
#endif
#ifndef NERF_X3_DIAG_MFMA_16X16
#define NERF_X3_DIAG_MFMA_16X16 0 // timing/power experiment only (results are garbage): the same FLOPs as two 16x16x32 MFMAs
#endif
#if NERF_X3_DIAG_MFMA_16X16
__device__ __forceinline__ f32x16 mfma_as_two_16x16(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, hi, 0, 0, 0);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3]; c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
    return c;
}
#define MFMA16(a, b, c) mfma_as_two_16x16((a), (b), (c))
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

struct B3 { u32x4 h, m, l; }; // the three bf16x8 fragments of one k-step's B operand

// two f32 values -> their three bf16 parts, packed pairwise.  The subtractions are exact: x - bf16(x) has at most 16
// significant bits left, the next one at most 8.
template <bool RELU>
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
    if (RELU) { x0 = relu(x0); x1 = relu(x1); }
    const f32x2 x = {x0, x1};
    const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
    const f32x2 r1 = x - hf;
    const uint32_t mu = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, mu << 16), __builtin_bit_cast(float, mu & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    h = hu; m = mu;
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

// The same split in four stages of about four VALU instructions, one per MFMA gap (an MFMA gap hides ~5 instructions).
struct PrepState { f32x2 x; uint32_t h, m; };

template <bool RELU, int KS, int Q, int STAGE>
__device__ __forceinline__ void prep_stage(const f32x16 &in, B3 &b, PrepState &st) {
    if constexpr (STAGE == 0) {
        float x0 = in[8 * KS + 2 * Q], x1 = in[8 * KS + 2 * Q + 1];
        asm volatile("" : "+v"(x0), "+v"(x1)); // keep the accumulator reads here (hipcc otherwise hoists a whole layer's)
        if (RELU) { x0 = relu(x0); x1 = relu(x1); }
        st.x = f32x2{x0, x1};
    } else if constexpr (STAGE == 1) {
        st.h = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
        const f32x2 hf = {__builtin_bit_cast(float, st.h << 16), __builtin_bit_cast(float, st.h & 0xffff0000u)};
        st.x = st.x - hf;
    } else if constexpr (STAGE == 2) {
        st.m = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
        const f32x2 mf = {__builtin_bit_cast(float, st.m << 16), __builtin_bit_cast(float, st.m & 0xffff0000u)};
        st.x = st.x - mf;
    } else {
        b.h[Q] = st.h; b.m[Q] = st.m;
        b.l[Q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(st.x, bf16x2));
    }
}

// pair Q (0..3) of k-step KS (0/1) of an input tile: registers 8 KS + 2 Q, + 1 (all stages at once: layer starts)
template <bool RELU, int KS, int Q>
__device__ __forceinline__ void prep_pair(const f32x16 &in, B3 &b) {
    PrepState st;
    prep_stage<RELU, KS, Q, 0>(in, b, st); prep_stage<RELU, KS, Q, 1>(in, b, st);
    prep_stage<RELU, KS, Q, 2>(in, b, st); prep_stage<RELU, KS, Q, 3>(in, b, st);
}

template <bool RELU, int KS>
__device__ __forceinline__ void prep_all(const f32x16 &in, B3 &b, PipeX &) {
    prep_pair<RELU, KS, 0>(in, b); prep_pair<RELU, KS, 1>(in, b); prep_pair<RELU, KS, 2>(in, b); prep_pair<RELU, KS, 3>(in, b);
}

#define X3_PIN() __builtin_amdgcn_sched_barrier(0)

// One k-step: NT units (units U0 .. U0 + NT - 1 of the chunk) with the prepared B `bc`.  A unit = the six products of one
// 32 x 16 weight block, small terms first, on one accumulator chain; the other work rides in its MFMA gaps: the next unit's
// three ds_reads behind MFMA 1, the LDS-DMA piece behind MFMA 2, and behind MFMAs 3..6 the four stages of splitting one pair
// of the NEXT k-step's B operand (tile `nin`, k-step NKS, into `bn`; pair q in unit 2 q of an 8-tile layer, unit q of viewdirs).
template <int NT, int U0, bool HAS_NEXT, bool NRELU, int NKS>
__device__ __forceinline__ void k_step(f32x16 (&out)[8], const B3 &bc, const f32x16 &nin, B3 &bn, PipeX &P) {
    const bf16x8 b1 = __builtin_bit_cast(bf16x8, bc.h), b2 = __builtin_bit_cast(bf16x8, bc.m), b3 = __builtin_bit_cast(bf16x8, bc.l);
    static_for<0, NT>([&](auto nt_c) {
        constexpr int nt = decltype(nt_c)::value;
        constexpr int U = U0 + nt;
        constexpr bool prep = HAS_NEXT && (NT == 4 || (nt & 1) == 0);
        constexpr int Q = NT == 4 ? nt : nt / 2;
        bf16x8 a1, a2, a3;
        PrepState st;
        pipe_take<U>(P, a1, a2, a3);
        X3_PIN();
#if NERF_X3_DIAG_DROP_W3 // accuracy / speed experiment only: five products, the weights' third part ignored
        asm volatile("" ::"v"(a3));
#else
        out[nt] = MFMA16(a3, b1, out[nt]);
#endif
        X3_PIN();
        pipe_prefetch<U>(P);
        X3_PIN();
        out[nt] = MFMA16(a2, b2, out[nt]);
        X3_PIN();
        pipe_dma<U>(P);
        X3_PIN();
        out[nt] = MFMA16(a1, b3, out[nt]);
        X3_PIN();
        if constexpr (prep) prep_stage<NRELU, NKS, Q, 0>(nin, bn, st);
        X3_PIN();
        out[nt] = MFMA16(a2, b1, out[nt]);
        X3_PIN();
        if constexpr (prep) prep_stage<NRELU, NKS, Q, 1>(nin, bn, st);
        X3_PIN();
        out[nt] = MFMA16(a1, b2, out[nt]);
        X3_PIN();
        if constexpr (prep) prep_stage<NRELU, NKS, Q, 2>(nin, bn, st);
        X3_PIN();
        out[nt] = MFMA16(a1, b1, out[nt]);
        X3_PIN();
        if constexpr (prep) prep_stage<NRELU, NKS, Q, 3>(nin, bn, st);
        X3_PIN();
    });
}

// bf16 parts have f32's exponent range: nothing can leave it (the f16x2 arithmetic checks its operands here)
__device__ __forceinline__ void range_check(PipeX &, unsigned int *, bool) {}

using PipeS = PipeX;
using BS = B3;
constexpr int kSplitChunksSigma = kChunksSigmaX3, kSplitChunksFull = kChunksFullX3, kSplitLdsBytes = kLdsBytesX3;
constexpr int kSplitWaveBytes = 6144; // a wave DMAs six 1-KiB pieces of every 24-KiB chunk

} // namespace

#define SPLIT_KERNEL_FUSED nerf_mlp_kernel_bf16x3
#define SPLIT_KERNEL_TRUNK nerf_trunk_seq_kernel_x3
#define SPLIT_KERNEL_COLOUR nerf_colour_kernel_x3
#define SPLIT_FN_INIT nerf_mlp_bf16x3_init
#define SPLIT_FN_LAUNCH nerf_mlp_bf16x3_launch
#define SPLIT_FN_SEQ_INIT nerf_seq_x3_init
#define SPLIT_FN_TRUNK_LAUNCH nerf_trunk_seq_x3_launch
#define SPLIT_FN_COLOUR_LAUNCH nerf_colour_x3_launch
#include "mlp_split_kernels.hip.h"
