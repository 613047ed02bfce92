// mlp_kernel_seq.hip -- exact dead-sample skipping (opt-in `skip_dead`), fp32 MFMA, gfx950.
//
// Two facts of the reference make work provably dead (compute_weights, src/lib.rs:261-280; integrate_ray, :176-195):
//   (A) once the transmittance T falls below 1e-4 every later weight of the ray is exactly 0 (:276-279): neither the
//       density nor the colour of those samples can reach the pixel;
//   (B) a sample with weight w = T * alpha == 0 (sigma == 0, or delta == 0) contributes 0 * rgb = 0: its colour head
//       (bottleneck + viewdirs + rgb, 17 % of an evaluation) is dead, its density is not (it decides the weights).
// The fused kernel (mlp_kernel.hip) evaluates everything.  Here ONE kernel does only the live work:
//
//   RAY-SEQUENTIAL trunk (dense0..7 + alpha -> sigma).  A wave owns one ray at a time and walks its samples in chunks of 32
//       (one MFMA column each), front to back; after every chunk it continues the reference's transmittance recurrence (same
//       operations, same order as k_composite) and RETIRES the ray at the T < 1e-4 cut -- the remaining chunks are never
//       evaluated (A).  Rays come from a device-side queue (one atomic per ray), so waves that retire rays early take more rays.
//   COLOUR PASSES on the live samples only, in the same kernel (round 3; rounds 1-2 exported 1 KiB per live sample to HBM for a
//       second launch: 16.5 GB written + 16.7 GB read per 800x800 frame, several passes per frame to bound that buffer).  The
//       trunk outputs h8 of the samples with w > 0 are COMPACTED IN LDS: a staging ring of 3 x 32 columns x 1 KiB beside the
//       weight ring; whenever 64 columns are staged the four waves of the workgroup run the colour head on them together,
//       N-SPLIT: every wave takes all 64 columns (two interleaved groups of 32; B operands re-read from the staging tiles) and a
//       quarter of the output features (bottleneck: 2 of 8 output tiles, viewdirs: 1 of 4), intermediate activations return to
//       the same staging tiles.  The weight ring switches to the colour part of the stream for the pass and prefetches the
//       trunk's first chunks at its end.
//
// Every value that reaches a pixel is produced by the same instruction sequence on the same operands as in the fused kernel
// (an MFMA output tile is a k-ordered fmaf chain per column, independent of the other columns and tiles; LDS round trips are
// exact), so the image is BIT-IDENTICAL to the non-skipping frame; only the amount of work changes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_f32_core.hip.h"
#include "mlp_kernel.h"
#include "mlp_seq_common.hip.h"

using namespace nerfmlp;
using namespace mlpdev;
using namespace mlpf32;
using namespace mlpseq;

namespace {

// Start the weight-stream pipeline on `n_chunks` chunks beginning at `stream` (prologue of the fused kernel).
__device__ __forceinline__ void pipe_start(Pipe &P, const LDS_AS char *lds, int lane16, int wave, const char *stream, int n_chunks) {
    P.lane16 = lane16;
    P.ring_lane = lds + lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 4096;
    P.wr_slot_off = 0;
    P.next_off = 0;
    P.stream_bytes = n_chunks * kChunkBytes;
    P.gbase = stream + wave * 4096;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kRingSlots - 1; ++c) pipe_issue(P);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int j = 0; j < NERF_LDS_GROUP; ++j) {
        P.nx[2 * j] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048);
        P.nx[2 * j + 1] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048 + 1024);
    }
}

} // namespace

// Timing-only diagnostics (results are WRONG with any of these set; never part of the product build -- `make variant` only):
#ifndef NERF_SEQ_DIAG_NO_PASS
#define NERF_SEQ_DIAG_NO_PASS 0 // 1: live samples are staged but no colour pass ever runs
#endif
#ifndef NERF_SEQ_DIAG_STAMP
#define NERF_SEQ_DIAG_STAMP 0 // k > 0: wave 0 of every workgroup accumulates s_memtime cycles of phase k of the colour passes into the NEXT launch
                              // slot's counters (read back as nerf_stats.n_hybrid_rays = passes, n_exec_coarse_trunk += cycles): 1 whole pass,
                              // 2 bottleneck loop, 3 viewdirs loop, 4 tail (rgb head), 5 from the vote barrier to the first pass, 6 between the two loops
#endif
#ifndef NERF_SEQ_DIAG_NO_STAGE
#define NERF_SEQ_DIAG_NO_STAGE 0 // 1: no staging either (the trunk alone, with the scan)
#endif

// ---- LDS layout of the EXPORT kernel: [weight ring][small parameters][staging: 3 tiles x 32 columns x 1 KiB][per staged column:
// sample index + ray direction][live counts of the four waves] -----------------------------------------------------------
constexpr int kStageTiles = 3;
constexpr int kStageCols = kStageTiles * 32;
constexpr int kStageTileBytes = 32 * 1024;            // [t * 4 + q][column + 32 h][4 floats]: h8 in the register layout of 8 activation tiles
constexpr int kStageOff = kLdsBytes;
constexpr int kStageMetaOff = kStageOff + kStageTiles * kStageTileBytes;
constexpr int kStageVoteOff = kStageMetaOff + kStageCols * 16;
constexpr int kSeqLdsBytes = kStageVoteOff + 16;
static_assert(kSeqLdsBytes <= 160 * 1024, "LDS");
constexpr int kColourChunks = kChunksFull - kChunksSigma; // 16 (bottleneck) + 9 (viewdirs)
static_assert(kRingSlots == 3, "the colour passes keep all three ring slots in flight");

namespace {

// ---- the weight ring during colour passes ---------------------------------------------------------------------------------
// A wave reads a whole chunk's A operands into registers half a chunk ahead, so a chunk's slot is free as soon as every wave
// has done that: THREE chunks are kept in flight (the trunk keeps two), because a colour chunk lasts only 2 048 cycles per wave
// (32 MFMAs) against 4 096 in the trunk and the L2 -> LDS latency must still hide behind it.
//   start:       issue chunks 0, 1, 2; wait for chunk 0 (vmcnt(8)) + barrier
//   mid chunk c: wait for chunk c + 1 (vmcnt(4): chunk c + 2 may be in flight) + barrier; read chunk c + 1's operands; issue chunk c + 3
//                into the slot of chunk c
// The stream wraps (a following pass finds its chunks 0, 1, 2 issued).  In the LAST pass of a burst the three chunks behind the
// end are the TRUNK's chunks 0 and 1 (and nothing): the trunk then continues in its usual two-chunks-in-flight state, no restart.
__device__ __forceinline__ void ring_issue_start(Pipe &P, const char *gbase, uint32_t stream_bytes, int n_chunks) {
    P.gbase = gbase;
    P.stream_bytes = stream_bytes;
    P.next_off = 0;
    P.wr_slot_off = 0;
    for (int c = 0; c < n_chunks; ++c) pipe_issue(P);
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
}

__device__ __forceinline__ void pipe_prime(Pipe &P) { // operands of the first macro-step of the trunk (what pipe_start / pipe_restart do)
#pragma unroll
    for (int j = 0; j < NERF_LDS_GROUP; ++j) {
        P.nx[2 * j] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048);
        P.nx[2 * j + 1] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048 + 1024);
    }
}

__device__ __forceinline__ void ring_advance(Pipe &P) {
    uint32_t off = P.rd_slot_off + kChunkBytes;
    off = (off == kRingSlots * kChunkBytes) ? 0u : off;
    P.rd_slot_off = off;
    P.rd_base = P.ring_lane + off;
}

// mid-chunk sync of a colour pass; returns whether a chunk was selected for refill (its four pieces follow behind the next MFMAs)
__device__ __forceinline__ bool colour_sync(Pipe &P, int chunk, bool last_pass, const char *trunk_gbase) {
    if (last_pass && chunk == kColourChunks - 3) { P.gbase = trunk_gbase; P.stream_bytes = kChunksSigma * kChunkBytes; P.next_off = 0; }
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    const bool refill = !(last_pass && chunk == kColourChunks - 1);
    if (refill) pipe_next_chunk(P);
    return refill;
}

struct Stage {
    LDS_AS char *tiles;     // staging tiles (a ring of kStageTiles)
    LDS_AS f32x4 *meta;     // per staged column: {sample index (bits), ray direction x, y, z}
    int head;               // first tile the next colour pass consumes (wave-uniform)
    int count;              // staged columns, counted from the start of tile `head` (wave-uniform)
};

__device__ __forceinline__ void load_b_tile(f32x4 (&raw)[4], const LDS_AS char *tile_lane, int tt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = *(const LDS_AS f32x4 *)(tile_lane + (tt * 4 + q) * 1024);
}

__device__ __forceinline__ void bias_tile(f32x16 &acc, const LDS_AS float *bias_nt_h) {
    const LDS_AS f32x4 *b = (const LDS_AS f32x4 *)bias_nt_h;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 u = b[q]; acc[4 * q + 0] = u[0]; acc[4 * q + 1] = u[1]; acc[4 * q + 2] = u[2]; acc[4 * q + 3] = u[3]; }
}

__device__ __forceinline__ void store_tile(LDS_AS char *tile, int lane, int slot_t, const f32x16 &acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 u;
        u[0] = acc[4 * q + 0]; u[1] = acc[4 * q + 1]; u[2] = acc[4 * q + 2]; u[3] = acc[4 * q + 3];
        *(LDS_AS f32x4 *)(tile + lane * 16 + (slot_t * 4 + q) * 1024) = u;
    }
}

// One colour pass: bottleneck + viewdirs + rgb (src/network.rs:218-223) on up to 64 staged columns = two GROUPS of 32 (tiles
// S.head and S.head + 1 of the staging ring; `n_cols` of them are real, the rest is stale LDS whose results are never stored).
// All four waves; wave w computes output tiles 2w, 2w+1 of the bottleneck and output tile w of viewdirs for ALL columns, the
// two groups interleaved: every A operand feeds both, which doubles the MFMAs per chunk -- a chunk carries ~300-500 cycles of fixed
// cost (mid-chunk wait + barrier, four DMA pieces per wave, operand reads): single-group passes ran at 92 cycles per MFMA, two groups at
// 73-80 (profiles/r03_seq_colour_pass.md; a dependent MFMA itself issues at 64.00: tools/probes/mfma_chain_probe.hip).
// Waves 0..2 each compute one colour channel.  On entry chunk 0 of the colour stream has landed (every wave past the barrier that
// proves it) and chunks 1, 2 are in flight.
__device__ __forceinline__ void colour_pass(Pipe &P, const Stage &S, int n_cols, bool last_pass, const char *trunk_gbase, const SeqArgs &A,
                                            const LDS_AS float *small, int wave, int lane, unsigned long long &diag) {
    (void)diag;
    const int p = lane & 31, h = lane >> 5;
    const int t0 = S.head, t1 = S.head + 1 == kStageTiles ? 0 : S.head + 1;
    LDS_AS char *tile0 = S.tiles + t0 * kStageTileBytes, *tile1 = S.tiles + t1 * kStageTileBytes;
    const LDS_AS char *tl0 = tile0 + lane * 16, *tl1 = tile1 + lane * 16;
    const f32x4 m0 = S.meta[t0 * 32 + p], m1 = S.meta[t1 * 32 + p]; // {sample index, direction}

    // ---- bottleneck: 128 k-steps, 2 KiB of the stream each (8 output tiles), chunk = 8 k-steps; this wave: tiles 2w, 2w+1 =
    // elements {2 (w & 1), 2 (w & 1) + 1} of piece g = w >> 1
    const int a_off = (wave >> 1) * 1024 + (wave & 1) * 8;
    f32x2 a_cur[8], a_nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_cur[j] = *(const LDS_AS f32x2 *)(P.rd_base + j * 2048 + a_off);
    f32x4 bn0[4], bn1[4];
    load_b_tile(bn0, tl0, 0);
    load_b_tile(bn1, tl1, 0);
    f32x16 acc00, acc01, acc10, acc11; // [group][tile 2w + k]
    bias_tile(acc00, small + kBiasOff + 8 * 256 + ((2 * wave) * 2 + h) * 16);
    bias_tile(acc01, small + kBiasOff + 8 * 256 + ((2 * wave + 1) * 2 + h) * 16);
    acc10 = acc00; acc11 = acc01;
    int chunk = 0;
#if NERF_SEQ_DIAG_STAMP == 2
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    for (int tt = 0; tt < 8; ++tt) {
        float bq0[16], bq1[16]; // B operands of this input tile: relu(h8) (the bottleneck reads the ReLU'd dense7 output, :218), one VALU burst
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { bq0[4 * q + e] = relu(bn0[q][e]); bq1[4 * q + e] = relu(bn1[q][e]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half, ++chunk) {
            bool refill = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j == 4) {
                    refill = colour_sync(P, chunk, last_pass, trunk_gbase);
                    ring_advance(P);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a_nxt[k] = *(const LDS_AS f32x2 *)(P.rd_base + k * 2048 + a_off);
                    if (half == 1 && tt < 7) { load_b_tile(bn0, tl0, tt + 1); load_b_tile(bn1, tl1, tt + 1); } // next input tile, half a chunk ahead
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float b0 = bq0[8 * half + j], b1 = bq1[8 * half + j];
                acc00 = MFMA(a_cur[j][0], b0, acc00);
                if (j >= 4) { __builtin_amdgcn_sched_barrier(0); if (refill) pipe_issue_piece(P, j - 4); __builtin_amdgcn_sched_barrier(0); }
                acc10 = MFMA(a_cur[j][0], b1, acc10);
                acc01 = MFMA(a_cur[j][1], b0, acc01);
                acc11 = MFMA(a_cur[j][1], b1, acc11);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) a_cur[k] = a_nxt[k];
        }
    }
#if NERF_SEQ_DIAG_STAMP == 2
    diag += __builtin_amdgcn_s_memtime() - ts0;
#endif
#if NERF_SEQ_DIAG_STAMP == 6
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    // bottleneck outputs (no activation) -> the staging tiles, as input tiles 2w, 2w+1 of viewdirs
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave has read its last h8 operands
    store_tile(tile0, lane, 2 * wave, acc00); store_tile(tile0, lane, 2 * wave + 1, acc01);
    store_tile(tile1, lane, 2 * wave, acc10); store_tile(tile1, lane, 2 * wave + 1, acc11);
    // ---- viewdirs: 144 k-steps, 1 KiB of the stream each (4 output tiles), chunk = 16 k-steps = one input tile; this wave: tile w
    const int v_off = wave * 4;
    float c_cur[16], c_nxt[16];
    // a_cur holds this chunk's operands in the bottleneck's addressing (fetched at the last mid-chunk sync): fetch viewdirs' own
#pragma unroll
    for (int j = 0; j < 16; ++j) c_cur[j] = *(const LDS_AS float *)(P.rd_base + j * 1024 + v_off);
    f32x16 V0, V1;
    bias_tile(V0, small + kBiasViewOff + (wave * 2 + h) * 16);
    V1 = V0;
    f32x16 D0, D1; // the ninth input tile: the columns' direction encodings (every wave needs them for its output tile)
    encode_dir<NERF_FAST_SINCOS != 0>(m0[1], m0[2], m0[3], h, D0);
    encode_dir<NERF_FAST_SINCOS != 0>(m1[1], m1[2], m1[3], h, D1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // bottleneck outputs visible
    load_b_tile(bn0, tl0, 0);
    load_b_tile(bn1, tl1, 0);
#if NERF_SEQ_DIAG_STAMP == 6
    diag += __builtin_amdgcn_s_memtime() - ts0;
#endif
#if NERF_SEQ_DIAG_STAMP == 3
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    for (int tt = 0; tt < 9; ++tt, ++chunk) { // input tiles 0..7: the bottleneck outputs (no activation); 8: the direction encoding (:219-220)
        float bq0[16], bq1[16];
        if (tt < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { bq0[4 * q + e] = bn0[q][e]; bq1[4 * q + e] = bn1[q][e]; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { bq0[r] = D0[r]; bq1[r] = D1[r]; }
        }
        bool refill = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == 8) {
                refill = colour_sync(P, chunk, last_pass, trunk_gbase);
                ring_advance(P);
                if (tt < 8) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) c_nxt[k] = *(const LDS_AS float *)(P.rd_base + k * 1024 + v_off);
                    if (tt < 7) { load_b_tile(bn0, tl0, tt + 1); load_b_tile(bn1, tl1, tt + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            V0 = MFMA(c_cur[j], bq0[j], V0);
            if (j >= 8 && (j & 1) == 0) { __builtin_amdgcn_sched_barrier(0); if (refill) pipe_issue_piece(P, (j - 8) >> 1); __builtin_amdgcn_sched_barrier(0); }
            V1 = MFMA(c_cur[j], bq1[j], V1);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) c_cur[k] = c_nxt[k];
    }
#if NERF_SEQ_DIAG_STAMP == 3
    diag += __builtin_amdgcn_s_memtime() - ts0;
#endif
#if NERF_SEQ_DIAG_STAMP == 4
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    // viewdirs outputs -> the staging tiles (input-tile slot w); waves 0..2 each run one channel of the rgb head (VALU, the fused
    // kernel's own code and order per channel) for both groups
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave has read its last operands from the tiles
    store_tile(tile0, lane, wave, V0);
    store_tile(tile1, lane, wave, V1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x16 VV0[4], VV1[4];
    if (wave < 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 u = *(const LDS_AS f32x4 *)(tl0 + (t * 4 + q) * 1024), v = *(const LDS_AS f32x4 *)(tl1 + (t * 4 + q) * 1024);
                VV0[t][4 * q + 0] = u[0]; VV0[t][4 * q + 1] = u[1]; VV0[t][4 * q + 2] = u[2]; VV0[t][4 * q + 3] = u[3];
                VV1[t][4 * q + 0] = v[0]; VV1[t][4 * q + 1] = v[1]; VV1[t][4 * q + 2] = v[2]; VV1[t][4 * q + 3] = v[3];
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the tiles are free for new deposits from here on
    if (wave < 3) {
        const float c0 = rgb_channel(VV0, small, h, wave), c1 = rgb_channel(VV1, small, h, wave);
        if (h == 0 && p < n_cols) A.rgb_out[3 * (size_t)__builtin_bit_cast(unsigned, m0[0]) + wave] = c0;
        if (h == 0 && 32 + p < n_cols) A.rgb_out[3 * (size_t)__builtin_bit_cast(unsigned, m1[0]) + wave] = c1;
    }
#if NERF_SEQ_DIAG_STAMP == 4
    diag += __builtin_amdgcn_s_memtime() - ts0;
#endif
}

} // namespace

template <bool EXPORT>
__global__ __launch_bounds__(256, 1) void nerf_trunk_seq_kernel(const SeqArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRingSlots * kChunkBytes);
    LDS_AS int *vote = (LDS_AS int *)(lds + kRingSlots * kChunkBytes) + kMiscOff + 8; // 4 ints of the padding behind the small parameters

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;
    {
        float *dst = (float *)(smem + kRingSlots * kChunkBytes);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    const char *trunk_gbase = (const char *)A.wstream + wave * 4096;
    const char *colour_gbase = trunk_gbase + (size_t)kChunksSigma * kChunkBytes;
    Pipe P;
    pipe_start(P, lds, lane * 16, wave, (const char *)A.wstream, kChunksSigma);

    Stage S;
    S.tiles = (LDS_AS char *)smem + kStageOff;
    S.meta = (LDS_AS f32x4 *)((LDS_AS char *)smem + kStageMetaOff);
    S.head = 0; S.count = 0;
    LDS_AS int *nlive = (LDS_AS int *)((LDS_AS char *)smem + kStageVoteOff);

    RayWork W;
    work_init(W, A);
    unsigned long long diag_cycles = 0;
    unsigned diag_passes = 0;
    for (;;) {
        const bool running = work_acquire(W, A, vote, wave, lane);
        if (!running && !(EXPORT && S.count > 0)) break;
        ChunkIn c{};
        LiveInfo li{};
        f32x16 Y[8];
        if (running) {
            c = chunk_inputs(W, A, p);
            f32x16 E[2];
            encode_point<NERF_FAST_SINCOS != 0>(c.px, c.py, c.pz, h, E);
            f32x16 X[8];
            load_bias<8>(X, small + kBiasOff + 0 * 256, h);
            tile_steps<8, false>(E[0], X, P);
            tile_steps<8, false>(E[1], X, P);
            hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
            hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
            hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
            hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
            load_bias<8>(Y, small + kBiasOff + 5 * 256, h); // dense5 on [encoding ; h4] (src/network.rs:209-210)
            tile_steps<8, false>(E[0], Y, P);
            tile_steps<8, false>(E[1], Y, P);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) tile_steps<8, true>(X[tt], Y, P);
            hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
            hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);
            const float sigma = alpha_head(Y, small, h);
            li = chunk_scan(W, A, c, sigma, p, h);
        }
        if (EXPORT) {
            // The four waves' live samples join the staging ring in wave order; whenever 64 columns are staged (or a wave's samples
            // would not fit into the 96) the workgroup runs colour passes.  All of this is workgroup-uniform control flow.
            if (running) {
                if (lane == 0) nlive[wave] = li.n_live;
                W.live_done += (unsigned)li.n_live;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // counts visible; the ring is idle, its prefetches landed
#if NERF_SEQ_DIAG_STAMP == 5
            const unsigned long long diag_vote = __builtin_amdgcn_s_memtime();
            bool diag_first = false;
#endif
            int n_w4[4] = {0, 0, 0, 0};
            if (running) {
#pragma unroll
                for (int w = 0; w < 4; ++w) n_w4[w] = __builtin_amdgcn_readfirstlane(nlive[w]);
            }
            const int total = n_w4[0] + n_w4[1] + n_w4[2] + n_w4[3];
            // 0 = trunk stream (its chunks 0, 1 prefetched), 1 = colour stream issued, 2 = colour stream running (chunk 0 proven landed)
            int ring = 0;
#if !(NERF_SEQ_DIAG_NO_PASS || NERF_SEQ_DIAG_NO_STAGE)
            if (S.count + total >= 64 || (!running && S.count > 0)) { // a pass will run in this step: start its stream under the deposits
                ring_issue_start(P, colour_gbase, kColourChunks * kChunkBytes, 3);
                ring = 1;
            }
#endif
            for (int w = 0; w <= 4; ++w) {
                const int n_w = w == 0 ? n_w4[0] : w == 1 ? n_w4[1] : w == 2 ? n_w4[2] : w == 3 ? n_w4[3] : 0;
                // flush condition: wave w's samples would overflow the 96 columns / end of the step with two full tiles / end of the kernel
                const bool flush = w < 4 ? (S.count + n_w > kStageCols) : (S.count >= 64 || (!running && S.count > 0));
#if NERF_SEQ_DIAG_NO_PASS || NERF_SEQ_DIAG_NO_STAGE
                if (flush) S.count &= 63;
                if (false) {
#else
                if (flush) {
#endif
                    if (ring == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // chunk 0 landed, deposits visible
                    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                    // deposits since the last pass visible
                    ring = 2;
                    while (S.count >= 64 || (!running && S.count > 0)) {
                        const int n_cols = S.count < 64 ? S.count : 64;
                        const int rest = S.count - n_cols;
                        const bool last = !(rest >= 64 || (!running && rest > 0)) && w == 4 && running; // the trunk continues right after this pass
#if NERF_SEQ_DIAG_STAMP == 1
                        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#if NERF_SEQ_DIAG_STAMP == 5
                        if (!diag_first) { diag_cycles += __builtin_amdgcn_s_memtime() - diag_vote; diag_first = true; }
#endif
                        colour_pass(P, S, n_cols, last, trunk_gbase, A, small, wave, lane, diag_cycles);
                        ++diag_passes;
#if NERF_SEQ_DIAG_STAMP == 1
                        diag_cycles += __builtin_amdgcn_s_memtime() - ts0;
#endif
                        S.head = (S.head + (n_cols > 32 ? 2 : 1)) % kStageTiles;
                        S.count = rest;
                        if (last) { ring = 0; pipe_prime(P); }
                    }
                }
                if (!NERF_SEQ_DIAG_NO_STAGE && w < 4 && wave == w && li.live) {
                    int slot = S.head * 32 + S.count + __popcll(li.mask & ((1ull << p) - 1ull));
                    slot = slot >= kStageCols ? slot - kStageCols : slot;
                    LDS_AS char *dst = S.tiles + (slot >> 5) * kStageTileBytes + ((slot & 31) + 32 * h) * 16;
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
                            v[0] = Y[tt][4 * q + 0]; v[1] = Y[tt][4 * q + 1]; v[2] = Y[tt][4 * q + 2]; v[3] = Y[tt][4 * q + 3];
                            *(LDS_AS f32x4 *)(dst + (tt * 4 + q) * 1024) = v;
                        }
                    if (h == 0) {
                        f32x4 m;
                        m[0] = __builtin_bit_cast(float, (unsigned)(c.base + c.s)); m[1] = c.dx; m[2] = c.dy; m[3] = c.dz;
                        S.meta[slot] = m;
                    }
                }
                S.count += n_w;
            }
            if (ring != 0) { // passes ran in the middle of the step only (a wave's samples did not fit): back to the trunk's stream
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the colour stream's in-flight chunks landed
                ring_issue_start(P, trunk_gbase, kChunksSigma * kChunkBytes, kRingSlots - 1);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                pipe_prime(P);
            }
        }
        if (!running) break;
        chunk_advance(W, A, c, li.cut, p, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // drain the unused DMA prefetches before the LDS allocation is released
    work_done(W, A, lane);
#if NERF_SEQ_DIAG_STAMP
    if (EXPORT && A.stats && wave == 0 && lane == 0) { // the next launch slot's counters: {u32 head, u32 live = passes, u64 = cycles}
        atomicAdd(A.stats + 2, diag_cycles);
        atomicAdd((unsigned *)(A.stats + 1) + 1, diag_passes);
    }
#else
    (void)diag_cycles; (void)diag_passes;
#endif
}

hipError_t nerf_seq_init() {
    hipError_t e = hipFuncSetAttribute((const void *)nerf_trunk_seq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSeqLdsBytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)nerf_trunk_seq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
}

size_t nerf_seq_h8_bytes(size_t n_samples) {
    const size_t tiles = (n_samples + kPointsPerBlock - 1) / kPointsPerBlock * kWavesPerBlock; // whole workgroup tiles of the colour kernel
    return tiles * kH8TileFloats * sizeof(float);
}

hipError_t nerf_trunk_seq_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream) {
    if (a.n_rays <= 0 || a.samples_per_ray <= 0) return hipSuccess;
    n_blocks = trunk_blocks(a, n_blocks);
    if (export_live && !a.rgb_out) return hipErrorInvalidValue;
    if (export_live) hipLaunchKernelGGL(nerf_trunk_seq_kernel<true>, dim3(n_blocks), dim3(256), kSeqLdsBytes, stream, a);
    else hipLaunchKernelGGL(nerf_trunk_seq_kernel<false>, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}
