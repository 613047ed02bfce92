// mlp_f32_core.hip.h -- the fp32-MFMA machinery shared by the fused MLP kernel (mlp_kernel.hip) and the ray-sequential
// trunk / compacted colour-head kernels (mlp_kernel_seq.hip): the LDS weight-stream pipeline, one input tile of a layer
// on v_mfma_f32_32x32x2_f32, bias initialisation and the VALU alpha head.  See mlp_layout.h for the data layout and
// DESIGN.md section 4.1 for why it is scheduled the way it is.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_common.hip.h"
#include "mlp_layout.h"

// Tuning switches (A/B-tested on MI355X).
#ifndef NERF_DMA_SPREAD
#define NERF_DMA_SPREAD 1 // 1: one LDS-DMA piece per macro-step behind an MFMA; 0: four pieces in a burst at the sync
#endif
#ifndef NERF_FAST_SINCOS
#define NERF_FAST_SINCOS 1 // 1: branch-free Cody-Waite + minimax sincos (<= 1.6 ulp for |x| <= 2^11); 0: ocml sincosf
#endif
// Timing-only diagnostics (results are WRONG with any of these set; never shipped):
#ifndef NERF_DIAG_NO_BARRIER
#define NERF_DIAG_NO_BARRIER 0
#endif
#ifndef NERF_DIAG_NO_DMA
#define NERF_DIAG_NO_DMA 0
#endif
#ifndef NERF_DIAG_NO_LDS
#define NERF_DIAG_NO_LDS 0
#endif
#define NERF_STR2(x) #x
#define NERF_STR(x) NERF_STR2(x)
// chunks allowed to stay in flight across the mid-chunk sync: kRingSlots - 3 (4 pieces each)
#define NERF_SYNC_VMCNT ((NERF_RING_SLOTS - 3) * 4)
#ifndef NERF_RELU_GROUP
#define NERF_RELU_GROUP 16 // (1: 92.1 %, 2: 93.4 %, 4: 94.1 %, 8: 94.3 %, 16: 94.4 % of the fp32 MFMA roofline) B operands (AGPR read + ReLU) are prepared for this many k-steps in ONE contiguous VALU burst
#endif
#ifndef NERF_LDS_GROUP
#define NERF_LDS_GROUP 1 // A operands are fetched from LDS for this many macro-steps per burst (1, 2 or 4)
#endif
#ifndef NERF_PIN_CHAINS
#define NERF_PIN_CHAINS 0 // 1: zero-instruction asm touching all accumulators after every input tile (keeps MFMA chains in program order)
#endif
#ifndef NERF_PREFETCH_INPUTS
#define NERF_PREFETCH_INPUTS 1 // 1: the next tile's t / direction are loaded one tile ahead
#endif



namespace mlpf32 {

using namespace nerfmlp;
using namespace mlpdev;

// ---- weight-stream pipeline -------------------------------------------------------------------
// The stream is consumed in "macro-steps" of 2 KiB = the A operands of 8 MFMAs (one k-step of an
// 8-tile layer, or two k-steps of the 4-tile viewdirs layer); 8 macro-steps per 16-KiB chunk.
struct Pipe {
    const LDS_AS char *rd_base; // LDS address (incl. lane*16) of the chunk the NEXT macro-step to fetch lives in
    const LDS_AS char *ring_lane; // ring base + lane*16
    uint32_t rd_slot_off;       // wave-uniform byte offset of that chunk's slot
    f32x4 nx[2 * NERF_LDS_GROUP]; // prefetched A operands of the next group of macro-steps
    f32x4 cu[2 * NERF_LDS_GROUP]; // A operands of the current group
    uint32_t ring_addr;      // LDS byte address of the ring + wave*4 KiB (DMA destination base)
    uint32_t wr_slot_off;    // byte offset of the slot the next DMA chunk goes to
    uint32_t next_off;       // byte offset in the stream of the next chunk to DMA
    uint32_t stream_bytes;   // bytes per tile
    const char *gbase;       // wave-uniform: stream + wave*4 KiB
    const char *cur_src;     // wave-uniform: this wave's quarter of the chunk being DMA'd
    uint32_t cur_dst;        // its LDS destination
    uint32_t lane16;
};

// Select the next chunk: its stream offset and ring slot (kept opaque so the 145 values are not constant-folded
// into 145 live address registers).
__device__ __forceinline__ void pipe_next_chunk(Pipe &P) {
    uint32_t off = P.next_off, slot = P.wr_slot_off;
    asm volatile("" : "+s"(off), "+s"(slot));
    P.cur_src = P.gbase + off;
    P.cur_dst = P.ring_addr + slot;
    off += kChunkBytes;
    P.next_off = (off == P.stream_bytes) ? 0u : off;
    slot += kChunkBytes;
    P.wr_slot_off = (slot == kRingSlots * kChunkBytes) ? 0u : slot;
#if NERF_M0_PER_CHUNK
    dma_set_dst(P.cur_dst);
#endif
}

__device__ __forceinline__ void pipe_issue_piece(Pipe &P, int i) {
#if !NERF_DIAG_NO_DMA && NERF_M0_PER_CHUNK
    glds_piece_m0(P.lane16, P.cur_src, i);
#elif !NERF_DIAG_NO_DMA
    glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
#else
    (void)P; (void)i;
#endif
}

__device__ __forceinline__ void pipe_issue(Pipe &P) {
    pipe_next_chunk(P);
#pragma unroll
    for (int i = 0; i < 4; ++i) pipe_issue_piece(P, i);
}

// Middle of chunk c: chunk c+1 (issued one chunk ago) must have landed; every wave is past chunk c-1, so
// its slot can be refilled with chunk c+2 (macro-steps 4..7 of chunk c issue one piece each, or all four here).
__device__ __forceinline__ void pipe_sync(Pipe &P) {
#if NERF_DIAG_NO_BARRIER
    asm volatile("s_waitcnt vmcnt(" NERF_STR(NERF_SYNC_VMCNT) ")" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(" NERF_STR(NERF_SYNC_VMCNT) ")\n\ts_barrier" ::: "memory");
#endif
#if NERF_DMA_SPREAD
    pipe_next_chunk(P);
#else
    pipe_issue(P);
#endif
}

// Called between the MFMAs of macro-step `ms` (0..7 within its chunk).
__device__ __forceinline__ void pipe_mid_step(Pipe &P, int ms) {
#if NERF_DMA_SPREAD
    if (ms >= 4) {
        __builtin_amdgcn_sched_barrier(0);
        pipe_issue_piece(P, ms - 4);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    (void)P; (void)ms;
#endif
}

// Abandon the rest of the current tile's weight stream and start over at chunk 0 (empty-tile skipping).  Every wave
// of the workgroup calls it at the same program point, right after a workgroup barrier, at a chunk boundary (all four
// pieces of the last selected chunk have been issued).
__device__ __forceinline__ void pipe_restart(Pipe &P) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // in-flight chunks landed; nobody reads the ring now
    P.next_off = 0;
    P.wr_slot_off = 0;
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

// Take the prefetched operands of macro-step `ms` (0..7 within its chunk) and start fetching those of the next one.
// All in-chunk addressing is a per-chunk base + immediate offset: VALU instructions are NOT free next to fp32 MFMAs
// (they share the vector datapath), so the ring arithmetic is one v_add per chunk plus scalar ops.
__device__ __forceinline__ void pipe_advance(Pipe &P, int ms, f32x4 &a0, f32x4 &a1) {
    constexpr int LG = NERF_LDS_GROUP;
    if (ms % LG == 0) {
#pragma unroll
        for (int j = 0; j < 2 * LG; ++j) P.cu[j] = P.nx[j];
        int nxt = ms + LG;
        if (nxt == 8) {
            uint32_t off = P.rd_slot_off + kChunkBytes;
            off = (off == kRingSlots * kChunkBytes) ? 0u : off;
            P.rd_slot_off = off;
            P.rd_base = P.ring_lane + off;
            nxt = 0;
        }
#pragma unroll
        for (int j = 0; j < LG; ++j) {
#if NERF_DIAG_NO_LDS
            asm volatile("" : "+v"(P.nx[2 * j]), "+v"(P.nx[2 * j + 1]));
#else
            P.nx[2 * j] = *(const LDS_AS f32x4 *)(P.rd_base + (nxt + j) * 2048);
            P.nx[2 * j + 1] = *(const LDS_AS f32x4 *)(P.rd_base + (nxt + j) * 2048 + 1024);
#endif
        }
        // keep the ds_reads of the NEXT group ahead of this group's MFMAs (otherwise hipcc sinks them below the
        // MFMAs into the same registers and exposes the LDS latency)
        __builtin_amdgcn_sched_barrier(0);
    }
    a0 = P.cu[2 * (ms % LG)];
    a1 = P.cu[2 * (ms % LG) + 1];
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---- one input tile (16 k-steps) of a layer with NT output tiles --------------------------------
// Every input tile starts on a chunk boundary (16 macro-steps for NT=8, 8 for NT=4), so the mid-chunk
// sync lands on macro-step 4 of every chunk.
template <int NT, bool RELU>
__device__ __forceinline__ void tile_steps(const f32x16 &in, f32x16 (&out)[NT], Pipe &P) {
    static_assert(NT == 8 || NT == 4, "NT");
    // Every interruption of the fp32 MFMA stream by VALU work costs more than the VALU instructions themselves, so
    // the B operands of G consecutive k-steps are prepared in one burst (G more live VGPRs).
    constexpr int G = RELU ? NERF_RELU_GROUP : 1;
    float bq[G];
    if constexpr (NT == 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 a0, a1;
            if ((r & 7) == 4) pipe_sync(P);
            pipe_advance(P, r & 7, a0, a1);
            if (r % G == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) bq[g] = RELU ? relu(in[r + g]) : in[r + g];
                if (G > 1) __builtin_amdgcn_sched_barrier(0);
            }
            const float b = bq[r % G];
            out[0] = MFMA(a0[0], b, out[0]); out[1] = MFMA(a0[1], b, out[1]);
            pipe_mid_step(P, r & 7);
            out[2] = MFMA(a0[2], b, out[2]); out[3] = MFMA(a0[3], b, out[3]);
            out[4] = MFMA(a1[0], b, out[4]); out[5] = MFMA(a1[1], b, out[5]);
            out[6] = MFMA(a1[2], b, out[6]); out[7] = MFMA(a1[3], b, out[7]);
        }
    } else {
        constexpr int G2 = G < 2 ? 2 : G;
        float bq2[G2];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x4 a0, a1;
            if (r / 2 == 4) pipe_sync(P);
            pipe_advance(P, r / 2, a0, a1);
            if (r % G2 == 0) {
#pragma unroll
                for (int g = 0; g < G2; ++g) bq2[g] = RELU ? relu(in[r + g]) : in[r + g];
                if (G2 > 2) __builtin_amdgcn_sched_barrier(0);
            }
            const float b0 = bq2[r % G2], b1 = bq2[r % G2 + 1];
            out[0] = MFMA(a0[0], b0, out[0]); out[1] = MFMA(a0[1], b0, out[1]);
            pipe_mid_step(P, r / 2);
            out[2] = MFMA(a0[2], b0, out[2]); out[3] = MFMA(a0[3], b0, out[3]);
            out[0] = MFMA(a1[0], b1, out[0]); out[1] = MFMA(a1[1], b1, out[1]);
            out[2] = MFMA(a1[2], b1, out[2]); out[3] = MFMA(a1[3], b1, out[3]);
        }
    }
#if NERF_PIN_CHAINS
    if constexpr (NT == 8)
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]), "+a"(out[4]), "+a"(out[5]), "+a"(out[6]), "+a"(out[7]));
    else
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]));
#endif
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&out)[NT], const LDS_AS float *bias, int h) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const LDS_AS f32x4 *b = (const LDS_AS f32x4 *)(bias + (nt * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[q];
            out[nt][4 * q + 0] = v[0]; out[nt][4 * q + 1] = v[1]; out[nt][4 * q + 2] = v[2]; out[nt][4 * q + 3] = v[3];
        }
    }
}

// hidden 256 -> 256 layer: bias init + 8 input tiles
template <bool RELU>
__device__ __forceinline__ void hidden_layer(const f32x16 (&in)[8], f32x16 (&out)[8], const LDS_AS float *bias,
                                             Pipe &P, int h) {
    load_bias<8>(out, bias, h);
#pragma unroll
    for (int t = 0; t < 8; ++t) tile_steps<8, RELU>(in[t], out, P);
}

// alpha head on the VALU: sigma = relu(b + sum_F w[F] relu(h8[F]))  (src/network.rs:216); alpha_pre = the value inside the relu
__device__ __forceinline__ float alpha_pre(const f32x16 (&Y)[8], const LDS_AS float *small, int h) {
    const LDS_AS f32x4 *w = (const LDS_AS f32x4 *)(small + kAlphaWOff + h * 128);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = w[t * 4 + q];
            a0 = fmaf(wv[0], relu(Y[t][4 * q + 0]), a0);
            a1 = fmaf(wv[1], relu(Y[t][4 * q + 1]), a1);
            a2 = fmaf(wv[2], relu(Y[t][4 * q + 2]), a2);
            a3 = fmaf(wv[3], relu(Y[t][4 * q + 3]), a3);
        }
    }
    return xhalf_sum((a0 + a1) + (a2 + a3)) + small[kMiscOff + 0];
}

__device__ __forceinline__ float alpha_head(const f32x16 (&Y)[8], const LDS_AS float *small, int h) { return fmaxf(alpha_pre(Y, small, h), 0.f); }

} // namespace mlpf32
