// mlp_kernel_seq.hip -- exact dead-sample skipping (opt-in `skip_dead`), fp32 MFMA, gfx950.
//
// Two facts of the reference make work provably dead (compute_weights, src/lib.rs:261-280; integrate_ray, :176-195):
//   (A) once the transmittance T falls below 1e-4 every later weight of the ray is exactly 0 (:276-279): neither the
//       density nor the colour of those samples can reach the pixel;
//   (B) a sample with weight w = T * alpha == 0 (sigma == 0, or delta == 0) contributes 0 * rgb = 0: its colour head
//       (bottleneck + viewdirs + rgb, 17 % of an evaluation) is dead, its density is not (it decides the weights).
// The fused kernel (mlp_kernel.hip) evaluates everything.  Here the evaluation is split in two launches:
//
//   nerf_trunk_seq_kernel   RAY-SEQUENTIAL trunk (dense0..7 + alpha -> sigma).  A wave owns one ray at a time and walks its
//       samples in chunks of 32 (one MFMA column each), front to back; after every chunk it continues the reference's
//       transmittance recurrence (same operations, same order as k_composite) and RETIRES the ray at the T < 1e-4 cut -- the
//       remaining chunks are never evaluated (A).  Rays come from a device-side queue (one atomic per ray), so waves that
//       retire rays early simply take more rays.  For every sample with w > 0 the wave exports the trunk's output h8
//       (256 floats, register layout) to a compacted HBM buffer (B).
//   nerf_colour_kernel      the colour head on the compacted live samples only, same MFMA sequence as the fused kernel's
//       tail, scattering rgb back to the sample's slot.
//
// Every value that reaches a pixel is produced by the same instruction sequence as in the fused kernel (MFMA columns are
// independent), so the image is BIT-IDENTICAL to the non-skipping frame; only the amount of work changes.
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
    Pipe P;
    pipe_start(P, lds, lane * 16, wave, (const char *)A.wstream, kChunksSigma);

    RayWork W;
    work_init(W, A);
    while (work_acquire(W, A, vote, wave, lane)) {
        const ChunkIn c = chunk_inputs(W, A, p);
        f32x16 E[2];
        encode_point<NERF_FAST_SINCOS != 0>(c.px, c.py, c.pz, h, E);
        f32x16 X[8], Y[8];
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
        chunk_finish<EXPORT>(W, A, c, sigma, Y, lane, p, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // drain the unused DMA prefetches before the LDS allocation is released
    work_done(W, A, lane);
}

__global__ __launch_bounds__(256, 1) void nerf_colour_kernel(const ColourArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRingSlots * kChunkBytes);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    {
        float *dst = (float *)(smem + kRingSlots * kChunkBytes);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    Pipe P;
    pipe_start(P, lds, lane * 16, wave, (const char *)A.wstream + (size_t)kChunksSigma * kChunkBytes, kChunksFull - kChunksSigma);

    unsigned n_live;
    const int n_tiles = colour_tiles(A, &n_live);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x16 X[8], Y[8];
        const ColourIn c = colour_inputs(A, n_live, tile, wave, lane, Y);
        hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck (no activation, src/network.rs:218)
        f32x16 D;
        encode_dir<NERF_FAST_SINCOS != 0>(c.dx, c.dy, c.dz, h, D);
        f32x16 V[4];
        load_bias<4>(V, small + kBiasViewOff, h);
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) tile_steps<4, false>(X[tt], V, P); // viewdirs on [bottleneck ; dir encoding] (:219-222)
        tile_steps<4, false>(D, V, P);
        float rgb[3];
        rgb_head(V, small, h, rgb);
        colour_store(A, c, rgb, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t nerf_seq_init() {
    const void *ks[3] = {(const void *)nerf_trunk_seq_kernel<true>, (const void *)nerf_trunk_seq_kernel<false>, (const void *)nerf_colour_kernel};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t nerf_seq_h8_bytes(size_t n_samples) {
    const size_t tiles = (n_samples + kPointsPerBlock - 1) / kPointsPerBlock * kWavesPerBlock; // whole workgroup tiles of the colour kernel
    return tiles * kH8TileFloats * sizeof(float);
}

hipError_t nerf_trunk_seq_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream) {
    if (a.n_rays <= 0 || a.samples_per_ray <= 0) return hipSuccess;
    n_blocks = trunk_blocks(a, n_blocks);
    if (export_live) hipLaunchKernelGGL(nerf_trunk_seq_kernel<true>, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    else hipLaunchKernelGGL(nerf_trunk_seq_kernel<false>, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t nerf_colour_launch(const ColourArgs &a, int n_blocks, hipStream_t stream) {
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL(nerf_colour_kernel, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}
