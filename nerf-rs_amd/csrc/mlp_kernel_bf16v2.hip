// mlp_kernel_bf16v2.hip -- bf16 MLP, output-tile-major, 64 points per wave (BASELINE config C5 study, second design).
//
// mlp_kernel_bf16.hip (v1) mirrors the f32 kernel: k-outer loop, all 8 accumulator tiles of a layer live, 32 points per
// wave.  At bf16 MFMA rates that design is bound by operand delivery: every A fragment (one ds_read_b128, 1 KiB of
// LDS-DMA per four waves) feeds ONE MFMA.  This kernel turns the loop nest around:
//   for each output tile nt:  acc{0,1} = bias;  for each k-step ks:  A = piece(nt, ks);  acc0 += A x B0[ks];  acc1 += A x B1[ks]
// * a wave owns 64 points = two 32-point sub-tiles; each A fragment feeds TWO MFMAs (half the LDS reads and half the
//   LDS-DMA per point; a workgroup covers 256 points per pass over the weight stream);
// * only 2 x 2 accumulator tiles are live (current + the one being converted); the layer's inputs and outputs are
//   bf16-PACKED registers (4 VGPRs per k-step and sub-tile): a finished f32 accumulator tile, ReLU'd and converted pairwise
//   (v_cvt_pk_bf16_f32 + v_pk_max_i16), IS k-steps 2 nt and 2 nt + 1 of the next layer (same k-permutation as v1: element j
//   of lane-half h of k-step s of a tile = the tile's register 8 s + j = feature regFeature(8 s + j, h));
// * the conversion of tile nt - 1 is interleaved, pair by pair, under the MFMAs of tile nt;
// * alpha (dense7) and rgb (viewdirs) heads accumulate in f32 from the f32 accumulators inside those epilogues, so the
//   arithmetic is exactly v1's (and the oracle emulation's): bf16 operands, f32 accumulate, f32 heads.
// Weight stream (host_util.cpp pack_network_bf16_v2): per layer, per output tile, per k-step one 1-KiB piece; 16-KiB chunks
// of 16 pieces; 60 chunks for the sigma layers, 73 for all (viewdirs' 72 pieces zero-padded to 80); ring of kRingSlotsBf16V2 slots; sync at
// piece 8 of every chunk; each wave DMA's one piece at pieces 9, 11, 13, 15.  All layers start at chunk boundaries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"
#include "mlp_seq_common.hip.h"

using namespace nerfmlp;
using namespace mlpdev;

// NERF_V2_F16 = 1 (mlp_kernel_f16v2.hip includes this file with it): the same kernels on f16 operands -- v_mfma_f32_32x32x16_f16, the same fragment
// layouts, the same stream order with f16-rounded weights.  Only the sigma-only forms are built: certify_zero's pre-filter, where f16's
// 11 significand bits make the pre-activation 8 x closer to the exact one than bf16's 8 (so far fewer samples need the exact kernel).  f16 overflows
// at 65 504: an overflow ends in a non-finite density pre-activation (range_left below), which counts in *nonfinite -- the host then goes back to bf16.
#ifndef NERF_V2_F16
#define NERF_V2_F16 0
#endif
#if NERF_V2_F16
typedef _Float16 h16x8 __attribute__((ext_vector_type(8))); // h16 = the 16-bit operand type of this build
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define V2SYM(bf16_name, f16_name) f16_name
#else
typedef __bf16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h16x2 __attribute__((ext_vector_type(2)));
#define V2SYM(bf16_name, f16_name) bf16_name
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef NERF_BV2_SCHED_BARRIER
#define NERF_BV2_SCHED_BARRIER 1 // keep the written interleave (A-operand prefetch distance, epilogue pairs between MFMA pairs)
#endif
#ifndef NERF_BV2_PREFETCH_INPUTS
#define NERF_BV2_PREFETCH_INPUTS 1 // load the next tile's raw inputs one tile ahead
#endif
#ifndef NERF_BV2_DMA_INST_OFFSET
#define NERF_BV2_DMA_INST_OFFSET 1 // one scalar base + M0 per chunk quarter, pieces addressed by the instruction offset
#endif
#ifndef NERF_BV2_DOUBLING
#define NERF_BV2_DOUBLING 1 // encodings by angle doubling from each lane-half's base octave (mlp_common.hip.h)
#endif
#if NERF_BV2_DOUBLING
#define ENCODE_POINT encode_point_doubling
#define ENCODE_DIR encode_dir_doubling
#else
#define ENCODE_POINT encode_point<true>
#define ENCODE_DIR encode_dir<true>
#endif
#ifndef NERF_BV2_M0_PER_CHUNK
#define NERF_BV2_M0_PER_CHUNK 1 // M0 (the LDS destination of an LDS-DMA piece) is written ONCE per chunk, where the chunk is selected, and the four pieces
#endif                          // of the wave's chunk quarter go out by instruction offset alone.  Round 4, tools/probes/lds_dma_stagger_probe.hip: beside MFMAs a
                                // piece costs 25 cycles with M0 saved / written / restored around it, 17 with M0 written only, 0 with M0 left alone -- the
                                // "price of an LDS-DMA instruction" of rounds 1-3 was the price of writing M0.  hipcc emits no M0 use of its own in these
                                // kernels (no LDS-direct, no s_movrel, no sendmsg; tests/test_host_logic.py checks the generated ISA).
#ifndef NERF_BV2_M0_NOSAVE
#define NERF_BV2_M0_NOSAVE 0 // (NERF_BV2_M0_PER_CHUNK=0 only) 1: drop the M0 save/restore around each LDS-DMA piece (+0.3 %)
#endif
#ifndef NERF_BV2_PAIR_READS
#define NERF_BV2_PAIR_READS 1 // A operands fetched two pieces at a time, one s_waitcnt per pair
#endif
#ifndef NERF_BV2_AHEAD
#define NERF_BV2_AHEAD 4
#endif
#ifndef NERF_BV2_HALF_SYNC
#define NERF_BV2_HALF_SYNC 0 // 1 (needs NERF_BV2_RING_SLOTS=5): one vmcnt + s_barrier per TWO chunks.  Three chunks are pre-loaded, chunk k + 3 is DMA'd
#endif                       // during chunk k into the slot of chunk k - 2; the barrier in the middle of every even chunk c proves chunks <= c + 2 landed
                             // (all issued before it) and every wave past chunk c - 1.  Round-4 experiment (profiles/r04_bf16_half_sync_ab.log):
                             // same results, sigma-only kernel 26.1 ms against 26.0 -- the barrier is not what costs -- and the full kernel slower.
#if NERF_BV2_HALF_SYNC
static_assert(NERF_BV2_RING_SLOTS == 5, "NERF_BV2_HALF_SYNC needs a five-slot ring");
#endif
// timing-only diagnostics (results are garbage): which resource bounds the kernel
#ifndef NERF_BV2_DIAG_NO_DMA
#define NERF_BV2_DIAG_NO_DMA 0
#endif
#ifndef NERF_BV2_DIAG_NO_BARRIER
#define NERF_BV2_DIAG_NO_BARRIER 0
#endif
#ifndef NERF_BV2_DIAG_NO_EPILOGUE
#define NERF_BV2_DIAG_NO_EPILOGUE 0
#endif
#ifndef NERF_BV2_DIAG_NO_LDS
#define NERF_BV2_DIAG_NO_LDS 0
#endif

namespace {

constexpr int kCB = kChunkBytesBf16V2, kRS = kRingSlotsBf16V2;
constexpr int kAhead = NERF_BV2_AHEAD; // A-operand prefetch distance in pieces

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct PipeV {
    const LDS_AS char *rd_base;   // LDS address (incl. lane * 16) of the chunk the next prefetched piece lives in
    const LDS_AS char *ring_lane;
    uint32_t rd_slot_off;
    u32x4 a[8];                   // ring of prefetched A operands (piece phase mod 8); kAhead (+1) of them are live
    uint32_t ring_addr, wr_slot_off, next_off, stream_bytes;
    const char *gbase, *cur_src;
    uint32_t cur_dst, lane16;
    uint32_t sync_phase;          // NERF_BV2_HALF_SYNC: 0 = this chunk's middle carries the barrier
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
#if NERF_BV2_M0_PER_CHUNK
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(P.cur_dst) : "memory"); // the chunk quarter's LDS destination; its pieces add their instruction offset
#endif
}

// piece at byte offset `off` of the chunk quarter pipe_next_chunk selected (pipe_start's bursts)
__device__ __forceinline__ void glds_piece_at(uint32_t lane16, const char *gsrc, uint32_t dst, int off) {
#if NERF_BV2_M0_PER_CHUNK
    (void)dst;
    if (off == 0)         asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(gsrc) : "memory");
    else if (off == 1024) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(gsrc) : "memory");
    else if (off == 2048) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane16), "s"(gsrc) : "memory");
    else                  asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(lane16), "s"(gsrc) : "memory");
#else
    glds_piece(lane16, gsrc + off, dst + off);
#endif
}

// (Re)start at chunk 0: chunks 0 .. kRS - 2 loaded and visible, the first kAhead A operands prefetched.  The caller guarantees
// that no wave still reads the ring (kernel start, or after a drain + barrier).
__device__ __forceinline__ void pipe_start(PipeV &P) {
    P.next_off = 0;
    P.wr_slot_off = 0;
    P.sync_phase = 0;
#pragma unroll
    for (int c = 0; c < (NERF_BV2_HALF_SYNC ? 3 : kRS - 1); ++c) {
        pipe_next_chunk(P);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds_piece_at(P.lane16, P.cur_src, P.cur_dst, i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int j = 0; j < kAhead; ++j) P.a[j] = *(const LDS_AS u32x4 *)(P.rd_base + j * 1024);
}

// Consume the piece at in-chunk phase PH (0..15): returns its A operand and refills the prefetch slot with the piece kAhead
// further on.  Phase 8: chunk c + 1 must have landed (every wave waits for its own pieces, then the barrier) and chunk
// c + kRS - 1's DMA starts into the slot chunk c - 1 occupied.
template <int PH>
__device__ __forceinline__ h16x8 pipe_take(PipeV &P) {
    if constexpr (PH == 8) {
        // chunk c + 1 was issued kRS - 2 chunks ago; the 4 (kRS - 3) pieces of the chunks issued since may still be in
        // flight (VMEM returns in order; any compiler-issued access in between only makes this wait longer, never shorter)
#if NERF_BV2_HALF_SYNC
        if (P.sync_phase == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // wave-uniform; every wave has consumed the same number of chunks
        P.sync_phase ^= 1u;
#elif NERF_BV2_DIAG_NO_BARRIER
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRS - 3)) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kRS - 3)) : "memory");
#endif
        pipe_next_chunk(P);
    }
#if NERF_BV2_DIAG_NO_LDS
    if constexpr (PH != 0) { u32x4 a = P.a[0]; asm volatile("" : "+v"(a)); return __builtin_bit_cast(h16x8, a); }
#endif
#if NERF_BV2_PAIR_READS
    static_assert(kAhead % 2 == 0, "paired reads need an even prefetch distance");
    if constexpr ((PH & 1) == 0) {
        // both operands of this piece pair are "used" here, so hipcc waits for them with ONE s_waitcnt
        asm volatile("" : "+v"(P.a[PH % 8]), "+v"(P.a[(PH + 1) % 8]));
        const u32x4 a = P.a[PH % 8];
        if constexpr (PH + kAhead == 16) { // the pair to prefetch opens the next chunk
            uint32_t off = P.rd_slot_off + kCB;
            off = (off == kRS * kCB) ? 0u : off;
            P.rd_slot_off = off;
            P.rd_base = P.ring_lane + off;
        }
        P.a[(PH + kAhead) % 8] = *(const LDS_AS u32x4 *)(P.rd_base + ((PH + kAhead) % 16) * 1024);
        P.a[(PH + kAhead + 1) % 8] = *(const LDS_AS u32x4 *)(P.rd_base + ((PH + kAhead + 1) % 16) * 1024);
        return __builtin_bit_cast(h16x8, a);
    } else {
        return __builtin_bit_cast(h16x8, P.a[PH % 8]);
    }
#else
    const u32x4 a = P.a[PH % 8];
    if constexpr (PH + kAhead == 16) { // the piece to prefetch is the first of the next chunk
        uint32_t off = P.rd_slot_off + kCB;
        off = (off == kRS * kCB) ? 0u : off;
        P.rd_slot_off = off;
        P.rd_base = P.ring_lane + off;
    }
    P.a[(PH + kAhead) % 8] = *(const LDS_AS u32x4 *)(P.rd_base + ((PH + kAhead) % 16) * 1024);
    return __builtin_bit_cast(h16x8, a);
#endif
}

// LDS-DMA piece whose instruction offset OFF advances the global AND the LDS address (both = base + OFF + lane * 16): the four
// pieces of a wave's chunk quarter share one scalar base pair and one M0 value.
template <int OFF>
__device__ __forceinline__ void glds_piece_off(uint32_t lane16, const char *gsrc, uint32_t dst) {
#if NERF_BV2_M0_PER_CHUNK
    (void)dst; // M0 holds it since pipe_next_chunk
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane16), "s"(gsrc), "n"(OFF) : "memory");
#elif NERF_BV2_DMA_INST_OFFSET && NERF_BV2_M0_NOSAVE
    // M0 is written and read inside this one statement and not restored: hipcc emits no M0 use of its own in this kernel
    // (no LDS-direct, no s_movrel, no sendmsg; checked on the generated ISA: every m0 reference is one of these statements)
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(lane16), "s"(gsrc), "s"(dst), "n"(OFF)
                 : "memory");
#elif NERF_BV2_DMA_INST_OFFSET
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane16), "s"(gsrc), "s"(dst), "n"(OFF)
                 : "memory");
#else
    glds_piece(lane16, gsrc + OFF, dst + OFF);
#endif
}

template <int PH>
__device__ __forceinline__ void pipe_dma(PipeV &P) {
#if NERF_BV2_DIAG_NO_DMA
    return;
#endif
    if constexpr (PH >= 9 && (PH & 1) == 1) glds_piece_off<((PH - 9) / 2) * 1024>(P.lane16, P.cur_src, P.cur_dst);
}

#if NERF_V2_F16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

// two f32 -> one packed bf16 pair (RNE); ReLU as a packed signed-16-bit max (a negative bf16 is a negative int16)
template <bool RELU>
__device__ __forceinline__ uint32_t pack2(float x, float y) {
    const f32x2 p = {x, y};
    h16x2 c = __builtin_convertvector(p, h16x2);
    if (RELU) {
        s16x2 i = __builtin_bit_cast(s16x2, c);
        i = __builtin_elementwise_max(i, (s16x2){0, 0});
        return __builtin_bit_cast(uint32_t, i);
    }
    return __builtin_bit_cast(uint32_t, c);
}

// a 16-register f32 tile -> its two packed k-steps (registers 0..7 -> k0, 8..15 -> k1); used for the encodings
__device__ __forceinline__ void pack_tile(const f32x16 &t, u32x4 &k0, u32x4 &k1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        k0[q] = pack2<false>(t[2 * q], t[2 * q + 1]);
        k1[q] = pack2<false>(t[8 + 2 * q], t[8 + 2 * q + 1]);
    }
}

struct Heads {
    float alpha[2];  // per sub-tile partial sums over this lane's features
    float rgb[2][3];
};

// Epilogue of one register pair PR (0..7) of a finished output tile for ONE sub-tile (4 VALU when only converting: two
// accumulator reads, v_cvt_pk_bf16_f32, v_pk_max_i16 -- sized to hide in one MFMA gap).
// HEAD 0: convert only; 1: convert + alpha partial sums; 2: alpha only (sigma kernels); 3: rgb partial sums only.
template <int PR, int SUB, bool RELU, int HEAD, int NTI>
__device__ __forceinline__ void convert_half(const f32x16 &acc, u32x4 &na, u32x4 &nb, Heads &H, const LDS_AS float *small, int h) {
#if NERF_BV2_DIAG_NO_EPILOGUE
    if constexpr (PR != 0) return;
#endif
    constexpr int r0 = 2 * PR, r1 = 2 * PR + 1;
    const float x0 = acc[r0], x1 = acc[r1];
    if constexpr (HEAD == 1 || HEAD == 2) { // alpha = sum_F w[F] relu(h8[F]) in f32 (src/network.rs:216)
        const f32x2 w = *(const LDS_AS f32x2 *)(small + kAlphaWOff + h * 128 + NTI * 16 + r0);
        H.alpha[SUB] = fmaf(w[1], relu(x1), fmaf(w[0], relu(x0), H.alpha[SUB]));
    }
    if constexpr (HEAD == 3) { // rgb pre-activations from relu(viewdirs) in f32 (src/network.rs:222-223)
        const float a0 = relu(x0), a1 = relu(x1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x2 w = *(const LDS_AS f32x2 *)(small + kRgbWOff + (h * 3 + c) * 64 + NTI * 16 + r0);
            H.rgb[SUB][c] = fmaf(w[1], a1, fmaf(w[0], a0, H.rgb[SUB][c]));
        }
    }
    if constexpr (HEAD == 0 || HEAD == 1) {
        const uint32_t c = pack2<RELU>(x0, x1);
        if constexpr (PR < 4) na[PR] = c;
        else                  nb[PR - 4] = c;
    }
}

// NERF_V2_F16: did the f16 range (65 504) not hold for one of this wave's points?  An activation beyond it is packed as +inf; every pre-activation of the
// next layer is then +-inf or NaN (inf x w, inf - inf), +inf and NaN survive the ReLU (a signed-integer max), and so on down to the density
// pre-activation, which comes out non-finite -- unless all 256 features of some layer turned -inf / negative NaN at once.  So nothing is watched in
// the hot loop: a non-finite density pre-activation marks the tile (and is never certified: k_cert_plan treats it as uncertain).  (wave-uniform)
__device__ __forceinline__ bool range_left(float pre0, float pre1) {
#if NERF_V2_F16
    return __any(!(fabsf(pre0) <= 3.0e38f) || !(fabsf(pre1) <= 3.0e38f));
#else
    (void)pre0; (void)pre1;
    return false;
#endif
}

// this lane-half's 16 bias values of output tile nt ([nt][h][16] in LDS)
__device__ __forceinline__ void load_bias(f32x16 &b, const LDS_AS float *bias, int nt, int h) {
    const LDS_AS f32x4 *src = (const LDS_AS f32x4 *)(bias + (nt * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = src[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[4 * q + e] = v[e];
    }
}

// k-step of the NEXT tile behind whose MFMAs pair pr of a finished tile is converted: spread over steps 1 .. LAST (not step
// 0: reading an accumulator right behind the MFMA that finished it costs the MFMA-write -> VALU-read wait states)
template <int LAST>
constexpr int pair_step(int pr) { return LAST == 15 ? (pr == 0 ? 1 : 2 * pr) // 16-piece tiles: even steps, the DMA pieces go out on odd ones
                                                    : 1 + pr * (LAST - 1) / 7; }

#if NERF_BV2_SCHED_BARRIER
#define BV2_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define BV2_PIN() ((void)0)
#endif

struct Acc { f32x16 a0, a1, b0, b1; }; // two accumulator sets x two sub-tiles; even output tiles use set a, odd ones set b

// One output tile NTI of a layer: KS pieces from stream phase PH (piece index mod 16), 2 MFMAs per piece.  `bv` holds this
// tile's bias (loaded one tile ahead) and is the C operand of the first MFMA pair; the next tile's bias is loaded into it
// right after.  Under the MFMAs runs the epilogue of the previously finished tile (p0, p1), pair pr at step
// pair_step<EP_LAST>(pr), sub-tile 0's half behind the first MFMA of the step, sub-tile 1's behind the second:
//   EP = 1: the previous tile of this layer -> out*[2 (NTI - 1)], [2 (NTI - 1) + 1] (e* arguments);
//   EP = 2: the LAST tile of the previous layer, whose packed result is k-steps PIDX, PIDX + 1 of this layer's own input:
//           written into in*[PIDX], in*[PIDX + 1] before the MFMAs of step PIDX read them (EP_LAST < PIDX).
template <int KS, int NT, int NTI, int PH, int EP, bool EP_RELU, int EP_HEAD, int EP_NTI, int EP_LAST>
__device__ __forceinline__ void out_tile(const u32x4 (&in0)[KS], const u32x4 (&in1)[KS], u32x4 &e0a, u32x4 &e0b, u32x4 &e1a, u32x4 &e1b,
                                         f32x16 &c0, f32x16 &c1, const f32x16 &p0, const f32x16 &p1, f32x16 &bv, const LDS_AS float *bias,
                                         const LDS_AS float *small, Heads &H, PipeV &P, int h) {
    static_for<0, KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        constexpr int ph = (PH + ks) % 16;
        const h16x8 a = pipe_take<ph>(P);
        if constexpr (ks == 0) c0 = MFMA16(a, __builtin_bit_cast(h16x8, in0[ks]), bv);
        else                   c0 = MFMA16(a, __builtin_bit_cast(h16x8, in0[ks]), c0);
        BV2_PIN();
        if constexpr (EP != 0)
            static_for<0, 8>([&](auto pr_c) {
                constexpr int pr = decltype(pr_c)::value;
                if constexpr (pair_step<EP_LAST>(pr) == ks) convert_half<pr, 0, EP_RELU, EP_HEAD, EP_NTI>(p0, e0a, e0b, H, small, h);
            });
        pipe_dma<ph>(P);
        BV2_PIN();
        if constexpr (ks == 0) c1 = MFMA16(a, __builtin_bit_cast(h16x8, in1[ks]), bv);
        else                   c1 = MFMA16(a, __builtin_bit_cast(h16x8, in1[ks]), c1);
        BV2_PIN();
        if constexpr (EP != 0)
            static_for<0, 8>([&](auto pr_c) {
                constexpr int pr = decltype(pr_c)::value;
                if constexpr (pair_step<EP_LAST>(pr) == ks) convert_half<pr, 1, EP_RELU, EP_HEAD, EP_NTI>(p1, e1a, e1b, H, small, h);
            });
        if constexpr (ks == 1 && NTI + 1 < NT) load_bias(bv, bias, NTI + 1, h); // the first MFMA pair has consumed bv
        BV2_PIN();
    });
}

// One layer: NT (even) output tiles, so every layer starts on accumulator set a and ends on set b.
//   PEND:  the previous layer deferred its last tile's epilogue (in set b): it runs under this layer's tile 0 and lands in
//          in*[PIDX], in*[PIDX + 1]; PEND_RELU is that layer's activation.
//   DEFER: leave this layer's last tile to the next layer in the same way instead of a tail burst (not for the head layers,
//          whose sums are needed right away).
// Every layer starts at stream phase 0 (layers are whole chunks; viewdirs is padded).
template <int KS, int NT, bool RELU_OUT, int HEAD, bool PEND, bool PEND_RELU, int PIDX, bool DEFER>
__device__ __forceinline__ void layer(u32x4 (&in0)[KS], u32x4 (&in1)[KS], u32x4 (&out0)[16], u32x4 (&out1)[16], Acc &C,
                                      const LDS_AS float *bias, const LDS_AS float *small, Heads &H, PipeV &P, int h) {
    static_assert(NT % 2 == 0, "layers alternate two accumulator sets and must end on set b");
    static_assert(!DEFER || HEAD == 0, "a deferred epilogue only converts");
    f32x16 bv;
    load_bias(bv, bias, 0, h);
    static_for<0, NT>([&](auto nt_c) {
        constexpr int nt = decltype(nt_c)::value;
        constexpr int ph = (nt * KS) % 16;
        constexpr int o = (2 * (nt > 0 ? nt - 1 : 0)) & 15;
        if constexpr (nt == 0) {
            if constexpr (PEND)
                out_tile<KS, NT, 0, ph, 2, PEND_RELU, 0, 0, PIDX - 2>(in0, in1, in0[PIDX], in0[PIDX + 1], in1[PIDX], in1[PIDX + 1], C.a0, C.a1, C.b0,
                                                                      C.b1, bv, bias, small, H, P, h);
            else
                out_tile<KS, NT, 0, ph, 0, false, 0, 0, 2>(in0, in1, out0[0], out0[1], out1[0], out1[1], C.a0, C.a1, C.b0, C.b1, bv, bias, small, H, P, h);
        } else if constexpr ((nt & 1) == 0) {
            out_tile<KS, NT, nt, ph, 1, RELU_OUT, HEAD, nt - 1, KS - 1>(in0, in1, out0[o], out0[o + 1], out1[o], out1[o + 1], C.a0, C.a1, C.b0, C.b1, bv,
                                                                        bias, small, H, P, h);
        } else {
            out_tile<KS, NT, nt, ph, 1, RELU_OUT, HEAD, nt - 1, KS - 1>(in0, in1, out0[o], out0[o + 1], out1[o], out1[o + 1], C.b0, C.b1, C.a0, C.a1, bv,
                                                                        bias, small, H, P, h);
        }
    });
    if constexpr (!DEFER) {
        constexpr int o = (2 * (NT - 1)) & 15;
        static_for<0, 8>([&](auto pr_c) {
            constexpr int pr = decltype(pr_c)::value;
            convert_half<pr, 0, RELU_OUT, HEAD, NT - 1>(C.b0, out0[o], out0[o + 1], H, small, h);
            convert_half<pr, 1, RELU_OUT, HEAD, NT - 1>(C.b1, out1[o], out1[o + 1], H, small, h);
        });
    }
}

// dense0 .. dense7 on the packed encodings E* (the sigma part of the stream, 60 chunks); the alpha partial sums land in H.  PACK_H8:
// also leave the bf16-packed relu(h8) in Y* (the bottleneck layer's input) -- only needed when a colour head follows.
template <bool PACK_H8>
__device__ __forceinline__ void trunk_layers(u32x4 (&E0)[4], u32x4 (&E1)[4], u32x4 (&X0)[16], u32x4 (&X1)[16], u32x4 (&Y0)[16], u32x4 (&Y1)[16], Acc &C,
                                             const LDS_AS float *small, Heads &H, PipeV &P, int h) {
    layer<4, 8, true, 0, false, false, 0, true>(E0, E1, X0, X1, C, small + kBiasOff + 0 * 256, small, H, P, h);   // dense0 (src/network.rs:204)
    layer<16, 8, true, 0, true, true, 14, true>(X0, X1, Y0, Y1, C, small + kBiasOff + 1 * 256, small, H, P, h);
    layer<16, 8, true, 0, true, true, 14, true>(Y0, Y1, X0, X1, C, small + kBiasOff + 2 * 256, small, H, P, h);
    layer<16, 8, true, 0, true, true, 14, true>(X0, X1, Y0, Y1, C, small + kBiasOff + 3 * 256, small, H, P, h);
    layer<16, 8, true, 0, true, true, 14, true>(Y0, Y1, X0, X1, C, small + kBiasOff + 4 * 256, small, H, P, h);
    {   // dense5 on [encoding (4 k-steps) ; h4 (16 k-steps)] (src/network.rs:209-210)
        u32x4 C0[20], C1[20];
#pragma unroll
        for (int k = 0; k < 4; ++k) { C0[k] = E0[k]; C1[k] = E1[k]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) { C0[4 + k] = X0[k]; C1[4 + k] = X1[k]; }
        layer<20, 8, true, 0, true, true, 18, true>(C0, C1, Y0, Y1, C, small + kBiasOff + 5 * 256, small, H, P, h); // h4's last tile lands in C*[18], [19]
    }
    layer<16, 8, true, 0, true, true, 14, true>(Y0, Y1, X0, X1, C, small + kBiasOff + 6 * 256, small, H, P, h);
    // dense7: alpha head from the f32 accumulators
    layer<16, 8, true, PACK_H8 ? 1 : 2, true, true, 14, false>(X0, X1, Y0, Y1, C, small + kBiasOff + 7 * 256, small, H, P, h);
}

// bottleneck + viewdirs + rgb (the colour part of the stream, 13 chunks) on the packed relu(h8) in Y*: sigmoid colours of both sub-tiles
__device__ __forceinline__ void colour_layers(u32x4 (&X0)[16], u32x4 (&X1)[16], u32x4 (&Y0)[16], u32x4 (&Y1)[16], Acc &C, const float (&d0)[3],
                                              const float (&d1)[3], const LDS_AS float *small, Heads &H, PipeV &P, int h, float (&c0)[3], float (&c1)[3]) {
    layer<16, 8, false, 0, false, false, 0, true>(Y0, Y1, X0, X1, C, small + kBiasOff + 8 * 256, small, H, P, h); // bottleneck: no activation (:218)
    u32x4 V0[18], V1[18];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V0[k] = X0[k]; V1[k] = X1[k]; }
    {
        f32x16 D;
        ENCODE_DIR(d0[0], d0[1], d0[2], h, D); pack_tile(D, V0[16], V0[17]);
        ENCODE_DIR(d1[0], d1[1], d1[2], h, D); pack_tile(D, V1[16], V1[17]);
    }
    layer<18, 4, true, 3, true, false, 14, false>(V0, V1, Y0, Y1, C, small + kBiasViewOff, small, H, P, h); // viewdirs + rgb partial sums (:220-223)
    {   // viewdirs' 72 pieces end at phase 8; the stream carries 8 zero pieces up to the chunk end: step over them
        h16x8 d;
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
    for (int c = 0; c < 3; ++c) { // sigmoid (src/network.rs:165)
        c0[c] = 1.0f / (1.0f + expf(-(xhalf_sum(H.rgb[0][c]) + small[kMiscOff + 1 + c])));
        c1[c] = 1.0f / (1.0f + expf(-(xhalf_sum(H.rgb[1][c]) + small[kMiscOff + 1 + c])));
    }
}

} // namespace

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void V2SYM(nerf_mlp_kernel_bf16v2, nerf_mlp_kernel_f16v2)(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;

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
    auto raw = [&](int tile_idx, int sub) -> RawIn { // clamped: padding lanes and the look-ahead tile read the last point
        RawIn r;
        int i = tile_idx * kPointsPerBlockBf16V2 + wave * 64 + sub * 32 + p;
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
#if NERF_BV2_PREFETCH_INPUTS
    RawIn nx0 = raw(blockIdx.x, 0), nx1 = raw(blockIdx.x, 1);
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int i0 = tile * kPointsPerBlockBf16V2 + wave * 64 + p, i1 = i0 + 32;
        const bool v0 = i0 < A.n_points, v1 = i1 < A.n_points;
#if NERF_BV2_PREFETCH_INPUTS
        const RawIn in0 = nx0, in1 = nx1;
        const int nt_idx = tile + gridDim.x < n_tiles ? tile + gridDim.x : tile;
        nx0 = raw(nt_idx, 0); nx1 = raw(nt_idx, 1);
#else
        const RawIn in0 = raw(tile, 0), in1 = raw(tile, 1);
#endif

        // position encodings -> 4 packed k-steps per sub-tile
        u32x4 E0[4], E1[4];
        {
            float px, py, pz;
            f32x16 E[2];
            point_of<MODE>(A, in0, px, py, pz);
            ENCODE_POINT(px, py, pz, h, E);
            pack_tile(E[0], E0[0], E0[1]); pack_tile(E[1], E0[2], E0[3]);
            point_of<MODE>(A, in1, px, py, pz);
            ENCODE_POINT(px, py, pz, h, E);
            pack_tile(E[0], E1[0], E1[1]); pack_tile(E[1], E1[2], E1[3]);
        }
        u32x4 X0[16], X1[16], Y0[16], Y1[16];
        Acc C;
        Heads H;
        H.alpha[0] = H.alpha[1] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) H.rgb[0][c] = H.rgb[1][c] = 0.f;

        trunk_layers<FULL>(E0, E1, X0, X1, Y0, Y1, C, small, H, P, h);
        const float pre0 = xhalf_sum(H.alpha[0]) + small[kMiscOff + 0], pre1 = xhalf_sum(H.alpha[1]) + small[kMiscOff + 0];
        if (range_left(pre0, pre1) && A.nonfinite && lane == 0) atomicAdd(A.nonfinite, 1u); // NERF_V2_F16 only: this wave's 64 points cannot be trusted
        // raw_pre (zero certification, nerf_api.cpp): the pre-activation itself leaves the kernel -- how far below 0 it is decides
        // whether the f32 kernel needs to look at the sample at all
        const float s0 = A.raw_pre ? pre0 : fmaxf(pre0, 0.f); // ReLU(alpha) (src/network.rs:216)
        const float s1 = A.raw_pre ? pre1 : fmaxf(pre1, 0.f);
        if (h == 0) {
            if (v0) A.sigma_out[i0] = s0;
            if (v1) A.sigma_out[i1] = s1;
        }
        if constexpr (FULL) {
            if (A.skip_empty) { // exact empty-tile skip, see mlp_kernel.hip: all 256 sigmas are 0 -> colours are never used
                LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
                const bool any_wg = tile_has_density(vote, (v0 && s0 > 0.0f) || (v1 && s1 > 0.0f), wave, lane);
                if (!any_wg) {
                    if (h == 0) {
                        if (v0) { A.rgb_out[3 * (size_t)i0] = 0.f; A.rgb_out[3 * (size_t)i0 + 1] = 0.f; A.rgb_out[3 * (size_t)i0 + 2] = 0.f; }
                        if (v1) { A.rgb_out[3 * (size_t)i1] = 0.f; A.rgb_out[3 * (size_t)i1 + 1] = 0.f; A.rgb_out[3 * (size_t)i1 + 2] = 0.f; }
                    }
                    if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 2ull); // counter unit = 128 points
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // in-flight chunks landed; ring idle
                    pipe_start(P);
                    continue;
                }
            }
            const float d0[3] = {in0.dx, in0.dy, in0.dz}, d1[3] = {in1.dx, in1.dy, in1.dz};
            float c0[3], c1[3];
            colour_layers(X0, X1, Y0, Y1, C, d0, d1, small, H, P, h, c0, c1);
            if (h == 0) {
                if (v0) { A.rgb_out[3 * (size_t)i0] = c0[0]; A.rgb_out[3 * (size_t)i0 + 1] = c0[1]; A.rgb_out[3 * (size_t)i0 + 2] = c0[2]; }
                if (v1) { A.rgb_out[3 * (size_t)i1] = c1[0]; A.rgb_out[3 * (size_t)i1 + 1] = c1[1]; A.rgb_out[3 * (size_t)i1 + 2] = c1[2]; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA may be in flight when the workgroup's LDS is released
    if (A.clock_out && tid == 0) { // diagnostic: shader clock = d(memtime) / d(memrealtime) x 100 MHz
        A.clock_out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        A.clock_out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

// ---- exact dead-sample skipping in the bf16 arithmetic (skip_dead with mlp_dtype = NERF_MLP_BF16; scheme: mlp_kernel_seq.hip, shared
// half: mlp_seq_common.hip.h) ---------------------------------------------------------------------------------------------------------
// A wave's two 32-point sub-tiles are two independent ray cursors: each takes rays from the device-side queue and walks its ray front
// to back in 32-sample chunks up to the reference's T < 1e-4 cut (src/lib.rs:276-279).  Two launches, as for the split arithmetics:
// the trunk exports the bf16-PACKED relu(h8) of the live samples (what the bottleneck layer reads: 16 k-steps x 16 B per lane-half =
// 512 B per sample, half of the f32 tiles' 1 KiB), the colour kernel runs bottleneck + viewdirs + rgb on the compacted slots, 256 per
// workgroup.  Per column the arithmetic is the fused kernel's (same layers, same stream), so the frame equals the non-skipping bf16 frame
// bit for bit.
namespace {
constexpr int kH8TileBytesBf16 = 16 * 64 * 16; // one 32-sample export tile: [k-step][lane] u32x4 = 16 KiB

__device__ __forceinline__ void pipe_begin(PipeV &P, const LDS_AS char *lds, int lane, int wave, const char *stream, int n_chunks) {
    P.lane16 = lane * 16;
    P.ring_lane = lds + P.lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 4096;
    P.stream_bytes = n_chunks * kCB;
    P.gbase = stream + wave * 4096;
    __syncthreads();
    pipe_start(P);
}

template <bool EXPORT>
__device__ __forceinline__ void chunk_finish_bf16(mlpseq::RayWork &W, const SeqArgs &A, const mlpseq::ChunkIn &c, float sigma, const u32x4 (&Y)[16],
                                                  int lane, int p, int h) {
    using namespace mlpseq;
    const LiveInfo li = chunk_scan(W, A, c, sigma, p, h);
    if (EXPORT) {
        if (li.n_live) {
            unsigned b = 0;
            if (lane == 0) b = atomicAdd(A.live_count, (unsigned)li.n_live);
            b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
            if (li.live) {
                const unsigned slot = b + (unsigned)__popcll(li.mask & ((1ull << p) - 1ull));
                char *dst = (char *)A.h8 + (size_t)(slot >> 5) * kH8TileBytesBf16 + ((slot & 31) + 32 * h) * 16;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) *(u32x4 *)(dst + ks * 1024) = Y[ks];
                if (h == 0) A.slot_point[slot] = (unsigned)(c.base + c.s);
            }
        }
    }
    chunk_advance(W, A, c, li.cut, p, h);
}
} // namespace

// certify_zero's pre-filter (SeqArgs.prefilter): the raw pre-activation leaves the kernel (how far below 0 it is decides whether the exact
// kernel looks at the sample at all), and the ray is retired where the bf16 transmittance predicts the cut -- nothing is exported.
__device__ __forceinline__ void chunk_finish_prefilter(mlpseq::RayWork &W, const SeqArgs &A, const mlpseq::ChunkIn &c, float pre, int p, int h) {
    using namespace mlpseq;
    const int M = A.samples_per_ray;
    if (c.valid && h == 0) A.sigma_out[c.base + c.s] = pre;
    float delta = (c.s + 1 < M) ? c.t_next - c.t : A.far_ - c.t;
    if (delta < 0.0f) delta = 0.0f;
    const float alpha = c.valid ? 1.0f - expf(-fmaxf(pre, 0.f) * delta) : 0.0f;
    float T = W.T;
    bool cut = false;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float al = lane_value(alpha, k);
        T = cut ? T : T * (1.0f - al);
        cut = cut || T < A.prefilter_cut_T;
    }
    W.T = T;
    if (c.has) { const int left = M - 32 * W.chunk; W.samples_done += (unsigned)(left < 32 ? left : 32); }
    chunk_advance(W, A, c, cut, p, h);
}

template <bool EXPORT>
__global__ __launch_bounds__(256, 1) void V2SYM(nerf_trunk_seq_kernel_bf16, nerf_trunk_seq_kernel_f16v2)(const SeqArgs A) {
    using namespace mlpseq;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);
    LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;
    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    PipeV P;
    pipe_begin(P, lds, lane, wave, (const char *)A.wstream, kChunksSigmaBf16V2);

    RayWork W0, W1;
    work_init(W0, A);
    work_init(W1, A);
    for (;;) {
        work_take(W0, A, lane);
        work_take(W1, A, lane);
        if (!work_vote(W0.ray < A.n_rays || W1.ray < A.n_rays, vote, wave, lane)) break;
        const ChunkIn c0 = chunk_inputs(W0, A, p), c1 = chunk_inputs(W1, A, p);
        u32x4 E0[4], E1[4];
        {
            f32x16 E[2];
            ENCODE_POINT(c0.px, c0.py, c0.pz, h, E);
            pack_tile(E[0], E0[0], E0[1]); pack_tile(E[1], E0[2], E0[3]);
            ENCODE_POINT(c1.px, c1.py, c1.pz, h, E);
            pack_tile(E[0], E1[0], E1[1]); pack_tile(E[1], E1[2], E1[3]);
        }
        u32x4 X0[16], X1[16], Y0[16], Y1[16];
        Acc C;
        Heads H;
        H.alpha[0] = H.alpha[1] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) H.rgb[0][c] = H.rgb[1][c] = 0.f;
        trunk_layers<EXPORT>(E0, E1, X0, X1, Y0, Y1, C, small, H, P, h);
        const float pre0 = xhalf_sum(H.alpha[0]) + small[kMiscOff + 0], pre1 = xhalf_sum(H.alpha[1]) + small[kMiscOff + 0];
        if (range_left(pre0, pre1) && A.nonfinite && lane == 0) atomicAdd(A.nonfinite, 1u); // NERF_V2_F16 only
        if (!EXPORT && A.prefilter) { // wave-uniform
            chunk_finish_prefilter(W0, A, c0, pre0, p, h);
            chunk_finish_prefilter(W1, A, c1, pre1, p, h);
            continue;
        }
        const float s0 = fmaxf(pre0, 0.f);
        const float s1 = fmaxf(pre1, 0.f);
        chunk_finish_bf16<EXPORT>(W0, A, c0, s0, Y0, lane, p, h);
        chunk_finish_bf16<EXPORT>(W1, A, c1, s1, Y1, lane, p, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    work_done(W0, A, lane);
    work_done(W1, A, lane);
}

#if !NERF_V2_F16 // the f16 build holds the sigma-only forms only
__global__ __launch_bounds__(256, 1) void nerf_colour_kernel_bf16(const ColourArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;
    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    PipeV P;
    pipe_begin(P, lds, lane, wave, (const char *)A.wstream + (size_t)kChunksSigmaBf16V2 * kCB, kChunksFullBf16V2 - kChunksSigmaBf16V2);

    const unsigned n_live = *A.live_count;
    const int n_tiles = (int)((n_live + (unsigned)kPointsPerBlockBf16V2 - 1u) / (unsigned)kPointsPerBlockBf16V2);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned slot0 = (unsigned)tile * kPointsPerBlockBf16V2 + wave * 64 + p, slot1 = slot0 + 32;
        const bool v0 = slot0 < n_live, v1 = slot1 < n_live;
        const unsigned i0 = A.slot_point[v0 ? slot0 : n_live - 1], i1 = A.slot_point[v1 ? slot1 : n_live - 1];
        const float *d0 = A.ray_dirs + 3 * (size_t)(i0 / (unsigned)A.samples_per_ray), *d1 = A.ray_dirs + 3 * (size_t)(i1 / (unsigned)A.samples_per_ray);
        const float d0x = d0[0], d0y = d0[1], d0z = d0[2], d1x = d1[0], d1y = d1[1], d1z = d1[2];
        u32x4 X0[16], X1[16], Y0[16], Y1[16];
        {   // the padding slots of the last tile read whatever the buffer holds (allocated; their columns are never stored)
            const char *src = (const char *)A.h8 + (size_t)(tile * 8 + wave * 2) * kH8TileBytesBf16 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                Y0[ks] = *(const u32x4 *)(src + ks * 1024);
                Y1[ks] = *(const u32x4 *)(src + kH8TileBytesBf16 + ks * 1024);
            }
        }
        Acc C;
        Heads H;
        H.alpha[0] = H.alpha[1] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) H.rgb[0][c] = H.rgb[1][c] = 0.f;
        const float d0v[3] = {d0x, d0y, d0z}, d1v[3] = {d1x, d1y, d1z};
        float c0[3], c1[3];
        colour_layers(X0, X1, Y0, Y1, C, d0v, d1v, small, H, P, h, c0, c1);
        if (h == 0) {
            if (v0) { A.rgb_out[3 * (size_t)i0] = c0[0]; A.rgb_out[3 * (size_t)i0 + 1] = c0[1]; A.rgb_out[3 * (size_t)i0 + 2] = c0[2]; }
            if (v1) { A.rgb_out[3 * (size_t)i1] = c1[0]; A.rgb_out[3 * (size_t)i1 + 1] = c1[1]; A.rgb_out[3 * (size_t)i1 + 2] = c1[2]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t nerf_seq_bf16_init() {
    const void *ks[3] = {(const void *)nerf_trunk_seq_kernel_bf16<true>, (const void *)nerf_trunk_seq_kernel_bf16<false>, (const void *)nerf_colour_kernel_bf16};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesBf16V2);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t nerf_seq_h8_bytes_bf16(size_t n_samples) { // capacity for n_samples live samples, rounded up to whole colour workgroups (256 slots)
    const size_t tiles = (n_samples + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2 * 8 + 8;
    return tiles * kH8TileBytesBf16;
}

hipError_t nerf_trunk_seq_bf16_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream) {
    if (a.n_rays <= 0 || a.samples_per_ray <= 0) return hipSuccess;
    const long long wg_rays = ((long long)a.n_rays + 7) / 8; // eight ray cursors per workgroup
    if (n_blocks > wg_rays) n_blocks = (int)wg_rays;
    if (n_blocks < 1) n_blocks = 1;
    if (export_live) hipLaunchKernelGGL(nerf_trunk_seq_kernel_bf16<true>, dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    else hipLaunchKernelGGL(nerf_trunk_seq_kernel_bf16<false>, dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}

hipError_t nerf_colour_bf16_launch(const ColourArgs &a, int n_blocks, hipStream_t stream) {
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL(nerf_colour_kernel_bf16, dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((nerf_mlp_kernel_bf16v2<FULL, MODE>), dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_bf16v2_init() {
    // forward_batch always evaluates the full head, so the sigma-only kernel exists in ray mode only
    const void *ks[3] = {(const void *)nerf_mlp_kernel_bf16v2<true, MLP_MODE_POINTS>, (const void *)nerf_mlp_kernel_bf16v2<true, MLP_MODE_RAYS>,
                         (const void *)nerf_mlp_kernel_bf16v2<false, MLP_MODE_RAYS>};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesBf16V2);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_mlp_bf16v2_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : hipErrorInvalidValue; // no sigma-only forward_batch
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
#else // NERF_V2_F16: the pre-filter's two launches (sigma only; certify_zero, nerf_api.cpp cert_pass)
hipError_t nerf_prefilter_f16v2_init() {
    const void *ks[2] = {(const void *)nerf_trunk_seq_kernel_f16v2<false>, (const void *)nerf_mlp_kernel_f16v2<false, MLP_MODE_RAYS>};
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesBf16V2);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_trunk_seq_f16v2_launch(const SeqArgs &a, int n_blocks, hipStream_t stream) { // the ray-sequential pre-filter
    if (a.n_rays <= 0 || a.samples_per_ray <= 0) return hipSuccess;
    if (!a.prefilter) return hipErrorInvalidValue;
    const long long wg_rays = ((long long)a.n_rays + 7) / 8; // eight ray cursors per workgroup
    if (n_blocks > wg_rays) n_blocks = (int)wg_rays;
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL(nerf_trunk_seq_kernel_f16v2<false>, dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_f16v2_launch(const MlpArgs &a, int n_blocks, hipStream_t stream) { // the fused pre-filter over all samples (sigma only, ray mode)
    if (a.n_points <= 0) return hipSuccess;
    if (a.mode != MLP_MODE_RAYS) return hipErrorInvalidValue;
    const int n_tiles = (a.n_points + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL((nerf_mlp_kernel_f16v2<false, MLP_MODE_RAYS>), dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}
#ifdef NERF_V2_F16_FULL // experiment (variant builds): the full f16 kernel, to price an f16 twin of the bf16 mode
hipError_t nerf_mlp_f16v2_full_launch(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    if (a.mode != MLP_MODE_RAYS) return hipErrorInvalidValue;
    const int n_tiles = (a.n_points + kPointsPerBlockBf16V2 - 1) / kPointsPerBlockBf16V2;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)nerf_mlp_kernel_f16v2<true, MLP_MODE_RAYS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesBf16V2); attr_set = true; }
    hipLaunchKernelGGL((nerf_mlp_kernel_f16v2<true, MLP_MODE_RAYS>), dim3(n_blocks), dim3(256), kLdsBytesBf16V2, stream, a);
    return hipGetLastError();
}
#endif
#endif
