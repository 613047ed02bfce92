// mlp_kernel_bf16.hip -- the fused NeRF MLP with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//
// BASELINE.json config C5 ("bf16 MLP on CDNA4 MFMA ... bandwidth/compute roofline study").  Same structure as the fp32
// kernel (mlp_kernel.hip): one persistent 256-thread workgroup per CU, a wave owns 32 points, activations stay in
// registers as fp32 accumulators in the 32x32 C/D layout.  A finished accumulator tile becomes the next layer's B
// operand without leaving registers: registers 8s..8s+7 of a tile, ReLU'd and converted pairwise with
// v_cvt_pk_bf16_f32, are the 8-element B fragment of k-step s (K = 16) -- element j of lane-half h is feature
// 16 s + 8 (j >> 2) + 4 h + (j & 3) of the tile, and the host packs the weight rows in exactly that order
// (host_util.cpp::pack_network_bf16).  Encodings, biases, accumulation, the alpha / rgb heads and the sigmoid stay fp32.
//
// What changes against fp32: the MFMA rate is 16x, so per 32-point tile the matrix work is 1160 MFMAs x 32 cycles =
// 37 k cycles while the whole bf16 network (1.16 MiB) still has to stream L2 -> LDS once per tile and per CU:
// 32 B/clk/CU.  The kernel is therefore bound by the weight stream (LDS-DMA issue + L2 bandwidth), not by the matrix
// pipe -- which is what the study is meant to show.
//
// Stream geometry: macro-step = 8 KiB = the A operands of 8 MFMAs (one K=16 step of an 8-tile layer, or two steps of
// the 4-tile viewdirs layer); 32-KiB chunks (4 macro-steps) in a 3-slot ring; one vmcnt(0)+s_barrier per chunk at
// macro-step 2; every macro-step issues 2 of the 8 LDS-DMA pieces this wave contributes to the chunk after next.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_common.hip.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"

using namespace nerfmlp;
using namespace mlpdev;

// Timing-only diagnostics (results are WRONG with any of these set; never shipped)
#ifndef NERF_BDIAG_NO_BARRIER
#define NERF_BDIAG_NO_BARRIER 0
#endif
#ifndef NERF_BDIAG_NO_DMA
#define NERF_BDIAG_NO_DMA 0
#endif
#ifndef NERF_BDIAG_NO_LDS
#define NERF_BDIAG_NO_LDS 0
#endif
#ifndef NERF_BDIAG_NO_BPREP
#define NERF_BDIAG_NO_BPREP 0
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));


namespace {

constexpr int kCB = kChunkBytesBf16;   // 32 KiB
constexpr int kRS = kRingSlotsBf16;    // 3
constexpr int kMS = 8192;              // bytes per macro-step

struct PipeB {
    const LDS_AS char *rd_base;   // LDS address (incl. lane*16) of the chunk the next macro-step to fetch lives in
    const LDS_AS char *ring_lane; // ring base + lane*16
    uint32_t rd_slot_off;
    bf16x8 nx[4];                 // prefetched A operands of the next HALF macro-step (4 MFMAs)
    uint32_t ring_addr;           // LDS byte address of the ring + wave*8 KiB (DMA destination base)
    uint32_t wr_slot_off, next_off, stream_bytes;
    const char *gbase;            // stream + wave*8 KiB
    const char *cur_src;
    uint32_t cur_dst;
    uint32_t lane16;
};

__device__ __forceinline__ void pipe_next_chunk(PipeB &P) {
    uint32_t off = P.next_off, slot = P.wr_slot_off;
    asm volatile("" : "+s"(off), "+s"(slot));
    P.cur_src = P.gbase + off;
    P.cur_dst = P.ring_addr + slot;
    off += kCB;
    P.next_off = (off == P.stream_bytes) ? 0u : off;
    slot += kCB;
    P.wr_slot_off = (slot == kRS * kCB) ? 0u : slot;
}

// this wave's 8 KiB of the chunk = 8 pieces; piece i
__device__ __forceinline__ void pipe_issue_piece(PipeB &P, int i) {
#if !NERF_BDIAG_NO_DMA
    glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
#else
    (void)P; (void)i;
#endif
}

// Half `hf` (0/1) of macro-step `ms` (0..3 within its chunk) begins.  At (ms == 2, hf == 0) the chunk after this one
// must have landed and the slot of the previous chunk may be refilled.  Takes the 4 prefetched A operands and starts
// fetching the next 4 (the other half of this macro-step, or the first half of the next one).  Only 4 + 4 operand
// fragments are live at a time: 8 + 8 cost 32 more VGPRs and made the full-head kernel spill.
__device__ __forceinline__ void pipe_half(PipeB &P, int ms, int hf, bf16x8 (&a)[4]) {
    if (ms == 2 && hf == 0) {
#if NERF_BDIAG_NO_BARRIER
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        pipe_next_chunk(P);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = P.nx[j];
    int nms = ms, nhf = hf + 1;
    if (nhf == 2) {
        nhf = 0;
        nms = ms + 1;
        if (nms == 4) {
            uint32_t off = P.rd_slot_off + kCB;
            off = (off == kRS * kCB) ? 0u : off;
            P.rd_slot_off = off;
            P.rd_base = P.ring_lane + off;
            nms = 0;
        }
    }
#if NERF_BDIAG_NO_LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(P.nx[j]));
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) P.nx[j] = *(const LDS_AS bf16x8 *)(P.rd_base + nms * kMS + (nhf * 4 + j) * 1024);
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// Bring the pipeline into the state a steady-state run is in when chunk 0 starts: chunk 0 landed; chunk 1 selected with
// the four pieces macro-steps 2,3 of "chunk -1" would have issued (macro-steps 0,1 of chunk 0 issue its pieces 4..7);
// operands of the first half macro-step fetched.  Called by every wave at the same point, after a workgroup barrier.
__device__ __forceinline__ void pipe_start(PipeB &P) {
    P.next_off = 0;
    P.wr_slot_off = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        pipe_next_chunk(P);
#pragma unroll
        for (int i = 0; i < (c < 1 ? 8 : 4); ++i) pipe_issue_piece(P, i);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) P.nx[j] = *(const LDS_AS bf16x8 *)(P.rd_base + j * 1024);
}

// The two DMA pieces of macro-step ms: pieces of the chunk selected at the last sync, in issue order
// ms 2 -> 0,1   ms 3 -> 2,3   ms 0 -> 4,5   ms 1 -> 6,7.
__device__ __forceinline__ void pipe_dma(PipeB &P, int ms, int k) {
    const int base = ((ms + 2) & 3) * 2;
    __builtin_amdgcn_sched_barrier(0);
    pipe_issue_piece(P, base + k);
    __builtin_amdgcn_sched_barrier(0);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// B fragment of k-step s: registers 8s .. 8s+7 of the tile -> 8 bf16 (element j = register 8s + j).
// The sources are pinned by an empty asm volatile so that hipcc cannot hoist the conversions of a whole layer's input
// tiles to the top of the layer (that cost 130 spilled VGPRs in the viewdirs layer).  ReLU is applied AFTER the
// conversion as one packed signed-16-bit max per pair (a negative bf16 is a negative int16; rounding keeps the sign).
template <bool RELU>
__device__ __forceinline__ bf16x8 make_b(const f32x16 &in, int s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = in[8 * s + j];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    bf16x8 b;
#if NERF_BDIAG_NO_BPREP
    b = __builtin_bit_cast(bf16x8, f32x4{x[0], x[1], x[2], x[3]});
    return b;
#endif
    typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 p = {x[2 * q], x[2 * q + 1]};
        bf16x2 c = __builtin_convertvector(p, bf16x2);
        if (RELU) {
            s16x2 i = __builtin_bit_cast(s16x2, c);
            i = __builtin_elementwise_max(i, (s16x2){0, 0});
            c = __builtin_bit_cast(bf16x2, i);
        }
        b[2 * q] = c[0];
        b[2 * q + 1] = c[1];
    }
    return b;
}

// One input tile (32 features) of a layer with NT output tiles.  MS0 = macro-step (within its chunk) the tile starts at.
template <int NT, bool RELU, int MS0, bool ACC_IN = true>
__device__ __forceinline__ void tile_steps(f32x16 &in, f32x16 (&out)[8], PipeB &P) { // NT = tiles of `out` in use
    // Pin the source tile where it is consumed: this volatile asm "redefines" the accumulator tuple, so the 16
    // v_accvgpr_reads below cannot be hoisted above it (hipcc otherwise hoists the reads of ALL input tiles of the
    // viewdirs layer to its start: 128 extra live VGPRs, 128 spills).  Encoding tiles live in VGPRs: no pin.
    if constexpr (ACC_IN) asm volatile("" : "+a"(in));
    const bf16x8 b0 = make_b<RELU>(in, 0), b1 = make_b<RELU>(in, 1);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[4];
    if constexpr (NT == 8) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 b = s ? b1 : b0;
            pipe_half(P, MS0 + s, 0, a);
            out[0] = MFMA16(a[0], b, out[0]); out[1] = MFMA16(a[1], b, out[1]);
            pipe_dma(P, MS0 + s, 0);
            out[2] = MFMA16(a[2], b, out[2]); out[3] = MFMA16(a[3], b, out[3]);
            pipe_half(P, MS0 + s, 1, a);
            out[4] = MFMA16(a[0], b, out[4]); out[5] = MFMA16(a[1], b, out[5]);
            pipe_dma(P, MS0 + s, 1);
            out[6] = MFMA16(a[2], b, out[6]); out[7] = MFMA16(a[3], b, out[7]);
        }
    } else {
        pipe_half(P, MS0, 0, a);
        out[0] = MFMA16(a[0], b0, out[0]); out[1] = MFMA16(a[1], b0, out[1]);
        pipe_dma(P, MS0, 0);
        out[2] = MFMA16(a[2], b0, out[2]); out[3] = MFMA16(a[3], b0, out[3]);
        pipe_half(P, MS0, 1, a);
        out[0] = MFMA16(a[0], b1, out[0]); out[1] = MFMA16(a[1], b1, out[1]);
        pipe_dma(P, MS0, 1);
        out[2] = MFMA16(a[2], b1, out[2]); out[3] = MFMA16(a[3], b1, out[3]);
    }
    // Keep every accumulation chain in program order: without this data dependence hipcc defers the whole chain of
    // some output tiles by hundreds of MFMAs (sched_barrier does not stop it), keeping their A operands alive --
    // which it then spills and reloads in front of each deferred MFMA.  Zero instructions.
    if constexpr (NT == 8)
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]), "+a"(out[4]), "+a"(out[5]), "+a"(out[6]), "+a"(out[7]));
    else
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]));
}

// A macro-step of stream padding: keep the pipeline bookkeeping (sync, prefetch, DMA), no matrix work.
template <int MS>
__device__ __forceinline__ void skip_macro_step(PipeB &P) {
    bf16x8 a[4];
    pipe_half(P, MS, 0, a);
    pipe_dma(P, MS, 0);
    pipe_half(P, MS, 1, a);
    pipe_dma(P, MS, 1);
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&out)[8], const LDS_AS float *bias, int h) {
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

template <bool RELU>
__device__ __forceinline__ void hidden_layer(f32x16 (&in)[8], f32x16 (&out)[8], const LDS_AS float *bias, PipeB &P, int h) {
    load_bias<8>(out, bias, h);
    tile_steps<8, RELU, 0>(in[0], out, P); tile_steps<8, RELU, 2>(in[1], out, P);
    tile_steps<8, RELU, 0>(in[2], out, P); tile_steps<8, RELU, 2>(in[3], out, P);
    tile_steps<8, RELU, 0>(in[4], out, P); tile_steps<8, RELU, 2>(in[5], out, P);
    tile_steps<8, RELU, 0>(in[6], out, P); tile_steps<8, RELU, 2>(in[7], out, P);
}

// alpha head on the VALU (f32): sigma = relu(b + sum_F w[F] relu(h8[F])).  The reads of each accumulator tile are pinned
// (empty asm volatile) so that hipcc does not hoist all 128 AGPR reads at once (spills next to the 64 A-operand VGPRs).
__device__ __forceinline__ float alpha_head(const f32x16 (&Y)[8], const LDS_AS float *small, int h) {
    const LDS_AS f32x4 *w = (const LDS_AS f32x4 *)(small + kAlphaWOff + h * 128);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x0 = Y[t][4 * q + 0], x1 = Y[t][4 * q + 1], x2 = Y[t][4 * q + 2], x3 = Y[t][4 * q + 3];
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            const f32x4 wv = w[t * 4 + q];
            a0 = fmaf(wv[0], relu(x0), a0);
            a1 = fmaf(wv[1], relu(x1), a1);
            a2 = fmaf(wv[2], relu(x2), a2);
            a3 = fmaf(wv[3], relu(x3), a3);
        }
    }
    return fmaxf(xhalf_sum((a0 + a1) + (a2 + a3)) + small[kMiscOff + 0], 0.f);
}

} // namespace

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void nerf_mlp_kernel_bf16(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;
    const int lane16 = lane * 16;

    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }

    PipeB P;
    P.lane16 = lane16;
    P.ring_lane = lds + lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 8192;
    P.stream_bytes = (FULL ? kChunksFullBf16 : kChunksSigmaBf16) * kCB;
    P.gbase = (const char *)A.wstream + wave * 8192;
    __syncthreads();
    pipe_start(P);

    const int n_tiles = (A.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    RawIn nxt = load_raw<MODE>(A, blockIdx.x, wave, p);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int i = tile * kPointsPerBlock + wave * kPointsPerWave + p;
        const bool valid = i < A.n_points;
        const RawIn in = nxt;
        nxt = load_raw<MODE>(A, tile + gridDim.x, wave, p);
        float px, py, pz;
        point_of<MODE>(A, in, px, py, pz);
        const float dx = in.dx, dy = in.dy, dz = in.dz;

        f32x16 E[2];
        encode_point<true>(px, py, pz, h, E);

        f32x16 X[8], Y[8];
        load_bias<8>(X, small + kBiasOff + 0 * 256, h);
        tile_steps<8, false, 0, false>(E[0], X, P);
        tile_steps<8, false, 2, false>(E[1], X, P);
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
        load_bias<8>(Y, small + kBiasOff + 5 * 256, h);
        tile_steps<8, false, 0, false>(E[0], Y, P);
        tile_steps<8, false, 2, false>(E[1], Y, P);
        tile_steps<8, true, 0>(X[0], Y, P); tile_steps<8, true, 2>(X[1], Y, P);
        tile_steps<8, true, 0>(X[2], Y, P); tile_steps<8, true, 2>(X[3], Y, P);
        tile_steps<8, true, 0>(X[4], Y, P); tile_steps<8, true, 2>(X[5], Y, P);
        tile_steps<8, true, 0>(X[6], Y, P); tile_steps<8, true, 2>(X[7], Y, P);
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);

        const float sigma = alpha_head(Y, small, h);
        if (valid && h == 0) A.sigma_out[i] = sigma;

        if (FULL && A.skip_empty) { // exact empty-tile skip, see mlp_kernel.hip
            LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
            const bool any_wg = tile_has_density(vote, valid && sigma > 0.0f, wave, lane);
            if (!any_wg) {
                if (valid && h == 0) {
                    A.rgb_out[3 * (size_t)i + 0] = 0.f; A.rgb_out[3 * (size_t)i + 1] = 0.f; A.rgb_out[3 * (size_t)i + 2] = 0.f;
                }
                if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 1ull);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // in-flight chunks landed; ring idle
                pipe_start(P);
                continue;
            }
        }

        if (FULL) {
            hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck, no activation on its output
            f32x16 D;
            encode_dir<true>(dx, dy, dz, h, D);
            // viewdirs accumulates into Y[0..3]: Y is dead after the bottleneck, and saying so explicitly keeps hipcc
            // from allocating a third accumulator set (which spilled 130 VGPRs)
            f32x16 (&V)[8] = Y;
            load_bias<4>(V, small + kBiasViewOff, h);
            tile_steps<4, false, 0>(X[0], V, P); tile_steps<4, false, 1>(X[1], V, P);
            tile_steps<4, false, 2>(X[2], V, P); tile_steps<4, false, 3>(X[3], V, P);
            tile_steps<4, false, 0>(X[4], V, P); tile_steps<4, false, 1>(X[5], V, P);
            tile_steps<4, false, 2>(X[6], V, P); tile_steps<4, false, 3>(X[7], V, P);
            tile_steps<4, false, 0, false>(D, V, P);
            skip_macro_step<1>(P); skip_macro_step<2>(P); skip_macro_step<3>(P); // stream padding to the chunk end
            float c[3];
            rgb_head(V, small, h, c);
            if (valid && h == 0) {
                A.rgb_out[3 * (size_t)i + 0] = c[0];
                A.rgb_out[3 * (size_t)i + 1] = c[1];
                A.rgb_out[3 * (size_t)i + 2] = c[2];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((nerf_mlp_kernel_bf16<FULL, MODE>), dim3(n_blocks), dim3(256), kLdsBytesBf16, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_bf16_init() {
    const void *ks[4] = {(const void *)nerf_mlp_kernel_bf16<true, MLP_MODE_POINTS>, (const void *)nerf_mlp_kernel_bf16<false, MLP_MODE_POINTS>,
                         (const void *)nerf_mlp_kernel_bf16<true, MLP_MODE_RAYS>, (const void *)nerf_mlp_kernel_bf16<false, MLP_MODE_RAYS>};
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesBf16);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_mlp_bf16_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_POINTS>(a, n_blocks, stream);
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
