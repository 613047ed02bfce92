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

using namespace nerfmlp;
using namespace mlpdev;
using namespace mlpf32;

namespace {

constexpr int kH8TileFloats = 32 * 64 * 4; // one 32-sample export tile: [t*4+q][lane][4] = 32 KiB

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

__device__ __forceinline__ float lane_value(float v, int k) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
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

    const int M = A.samples_per_ray;
    const int n_chunks = (M + 31) >> 5;
    int ray = A.n_rays, chunk = n_chunks; // no ray yet
    float T = 1.0f;                       // transmittance in front of the current chunk (wave-uniform)
    unsigned long long chunks_done = 0;
    for (;;) {
        if (chunk >= n_chunks) { // next ray from the queue
            unsigned r = 0;
            if (lane == 0) r = atomicAdd(A.ray_counter, 1u);
            r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
            ray = r < (unsigned)A.n_rays ? (int)r : A.n_rays;
            chunk = 0;
            T = 1.0f;
        }
        const bool has = ray < A.n_rays;
        // the four waves walk the weight stream in lockstep: leave together once nobody has a ray
        if (lane == 0) vote[wave] = has ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if ((vote[0] | vote[1] | vote[2] | vote[3]) == 0) break;

        const int s = chunk * 32 + p;
        const bool valid = has && s < M;
        const size_t base = (size_t)(has ? ray : 0) * M;
        const float t = A.t[base + (s < M ? s : M - 1)];
        const float t_next = A.t[base + (s + 1 < M ? s + 1 : M - 1)];
        const float *dv = A.ray_dirs + 3 * (size_t)(has ? ray : 0);
        const float dx = dv[0], dy = dv[1], dz = dv[2];
        // p = origin + d_hat * t, multiply and add rounded separately (src/lib.rs:436)
        const float px = __fadd_rn(A.origin[0], __fmul_rn(dx, t));
        const float py = __fadd_rn(A.origin[1], __fmul_rn(dy, t));
        const float pz = __fadd_rn(A.origin[2], __fmul_rn(dz, t));

        f32x16 E[2];
        encode_point<NERF_FAST_SINCOS != 0>(px, py, pz, h, E);
        f32x16 X[8], Y[8];
        load_bias<8>(X, small + kBiasOff + 0 * 256, h);
        tile_steps<8, false>(E[0], X, P);
        tile_steps<8, false>(E[1], X, P);
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
        load_bias<8>(Y, small + kBiasOff + 5 * 256, h);
        tile_steps<8, false>(E[0], Y, P);
        tile_steps<8, false>(E[1], Y, P);
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) tile_steps<8, true>(X[tt], Y, P);
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);
        const float sigma = alpha_head(Y, small, h);
        if (valid && h == 0) A.sigma_out[base + s] = sigma;

        // compute_weights through this chunk (src/lib.rs:261-280): the operations and their order are k_composite's
        float delta = (s + 1 < M) ? t_next - t : A.far_ - t;
        if (delta < 0.0f) delta = 0.0f;
        const float alpha = valid ? 1.0f - expf(-sigma * delta) : 0.0f;
        float my_w = 0.0f;
        bool cut = false;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float al = lane_value(alpha, k);
            const float wk = cut ? 0.0f : T * al;
            if (p == k) my_w = wk;
            T = cut ? T : T * (1.0f - al);
            cut = cut || T < 1e-4f;
        }
        if (has) ++chunks_done;

        if (EXPORT) {
            const bool live = my_w > 0.0f; // exactly the samples whose colour reaches the pixel
            const unsigned long long m = __ballot(live) & 0xffffffffull;
            const int n_live = __popcll(m);
            if (n_live) {
                unsigned b = 0;
                if (lane == 0) b = atomicAdd(A.live_count, (unsigned)n_live);
                b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
                if (live) {
                    const unsigned slot = b + (unsigned)__popcll(m & ((1ull << p) - 1ull));
                    float *dst = A.h8 + (size_t)(slot >> 5) * kH8TileFloats + ((slot & 31) + 32 * h) * 4;
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
                            v[0] = Y[tt][4 * q + 0]; v[1] = Y[tt][4 * q + 1]; v[2] = Y[tt][4 * q + 2]; v[3] = Y[tt][4 * q + 3];
                            *(f32x4 *)(dst + (tt * 4 + q) * 256) = v;
                        }
                    if (h == 0) A.slot_point[slot] = (unsigned)(base + s);
                }
            }
        }
        chunk = (cut || !has) ? n_chunks : chunk + 1; // (A): behind the cut nothing of this ray is evaluated
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // drain the unused DMA prefetches before the LDS allocation is released
    if (A.stats && lane == 0 && chunks_done) atomicAdd(A.stats, chunks_done);
}

__global__ __launch_bounds__(256, 1) void nerf_colour_kernel(const ColourArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRingSlots * kChunkBytes);
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
    pipe_start(P, lds, lane * 16, wave, (const char *)A.wstream + (size_t)kChunksSigma * kChunkBytes, kChunksFull - kChunksSigma);

    const unsigned n_live = *A.live_count;
    const int n_tiles = (int)((n_live + (unsigned)kPointsPerBlock - 1u) / (unsigned)kPointsPerBlock);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned slot = (unsigned)tile * kPointsPerBlock + wave * kPointsPerWave + p;
        const bool valid = slot < n_live;
        const unsigned i = A.slot_point[valid ? slot : n_live - 1];
        const float *dv = A.ray_dirs + 3 * (size_t)(i / (unsigned)A.samples_per_ray);
        const float dx = dv[0], dy = dv[1], dz = dv[2];
        const float *src = A.h8 + (size_t)(tile * kWavesPerBlock + wave) * kH8TileFloats + lane * 4;
        f32x16 X[8], Y[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4 *)(src + (tt * 4 + q) * 256);
                Y[tt][4 * q + 0] = v[0]; Y[tt][4 * q + 1] = v[1]; Y[tt][4 * q + 2] = v[2]; Y[tt][4 * q + 3] = v[3];
            }
        hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck (no activation, src/network.rs:218)
        f32x16 D;
        encode_dir<NERF_FAST_SINCOS != 0>(dx, dy, dz, h, D);
        f32x16 V[4];
        load_bias<4>(V, small + kBiasViewOff, h);
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) tile_steps<4, false>(X[tt], V, P);
        tile_steps<4, false>(D, V, P);
        float c[3];
        rgb_head(V, small, h, c);
        if (valid && h == 0) {
            A.rgb_out[3 * (size_t)i + 0] = c[0];
            A.rgb_out[3 * (size_t)i + 1] = c[1];
            A.rgb_out[3 * (size_t)i + 2] = c[2];
        }
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
    const long long wave_rays = ((long long)a.n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
    if (n_blocks > wave_rays) n_blocks = (int)wave_rays;
    if (n_blocks < 1) n_blocks = 1;
    if (export_live) hipLaunchKernelGGL(nerf_trunk_seq_kernel<true>, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    else hipLaunchKernelGGL(nerf_trunk_seq_kernel<false>, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t nerf_colour_launch(const ColourArgs &a, int n_blocks, hipStream_t stream) {
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL(nerf_colour_kernel, dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}
