// mlp_kernel.hip -- fused NeRF MLP forward for gfx950 (MI355X), fp32 MFMA.
//
// Replaces Network::forward_batch (reference src/network.rs:197-237) and, in ray mode, the point fill of
// render_block (src/lib.rs:392-402, 432-443): p = origin + d_hat * t.
//
// Structure (see mlp_layout.h for the data layout):
//   * one persistent 256-thread workgroup per CU, 4 waves, each wave owns 32 points per tile;
//   * a wave's activations (256 features x 32 points) live in 128 registers in the C/D layout of
//     v_mfma_f32_32x32x2_f32; that layout IS the B-operand layout of the next layer's MFMA once the
//     weight rows are permuted (done once on the host), so activations never touch LDS or HBM;
//   * ReLU is applied on the consumer side (one v_max per k-step, hidden under 8 MFMAs);
//   * biases initialise the accumulators (as the reference does: fill_with_bias, src/network.rs:149-159);
//   * the packed weight stream (2.3 MB, L2-resident) is DMA'd global->LDS (global_load_lds_dwordx4)
//     in 16-KiB chunks into a 3-slot ring shared by the 4 waves; one s_barrier per chunk, placed in the
//     middle of the chunk so the next chunk's first operands can be fetched before they are needed;
//   * alpha (N=1) and rgb (N=3) heads run on the VALU from the register-resident activations.
// Per 32-point wave tile: 9280 MFMAs (full) / 7680 (sigma only) x 64 cycles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_f32_core.hip.h"
#include "mlp_kernel.h"

using namespace nerfmlp;
using namespace mlpdev;

using namespace mlpf32;

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void nerf_mlp_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRingSlots * kChunkBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31;
    const int h = lane >> 5;
    const int lane16 = lane * 16;

    // resident small parameters -> LDS
    {
        float *dst = (float *)(smem + kRingSlots * kChunkBytes);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }

    Pipe P;
    P.lane16 = lane16;
    P.ring_lane = lds + lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * 4096;
    P.wr_slot_off = 0;
    P.next_off = 0;
    P.stream_bytes = (FULL ? kChunksFull : kChunksSigma) * kChunkBytes;
    P.gbase = (const char *)A.wstream + wave * 4096;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kRingSlots - 1; ++c) pipe_issue(P);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned long long clk0 = 0, rt0 = 0;
    if (A.clock_out) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int j = 0; j < NERF_LDS_GROUP; ++j) { // prime: operands of the first group of macro-steps
        P.nx[2 * j] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048);
        P.nx[2 * j + 1] = *(const LDS_AS f32x4 *)(P.rd_base + j * 2048 + 1024);
    }

    const int n_points = MODE == MLP_MODE_LIST ? list_length(A) : A.n_points; // list mode: the length lives on the device
    const int n_tiles = (n_points + kPointsPerBlock - 1) / kPointsPerBlock;
#if NERF_PREFETCH_INPUTS
    RawIn nxt = load_raw<MODE>(A, blockIdx.x, wave, p);
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int slot = tile * kPointsPerBlock + wave * kPointsPerWave + p;
        const bool valid = slot < n_points;
#if NERF_PREFETCH_INPUTS
        const RawIn in = nxt;
        nxt = load_raw<MODE>(A, tile + gridDim.x, wave, p); // consumed one tile (~250 us) later
#else
        const RawIn in = load_raw<MODE>(A, tile, wave, p);
#endif
        float px, py, pz;
        point_of<MODE>(A, in, px, py, pz);
        const float dx = in.dx, dy = in.dy, dz = in.dz;
        const unsigned entry = __builtin_bit_cast(unsigned, in.b);
        const size_t i = MODE == MLP_MODE_LIST ? (size_t)(entry & 0x7fffffffu) : (size_t)slot; // where the outputs go
        // an audited certificate (certify_zero): the raw pre-activation leaves the kernel -- k_cert_audit compares it with 0
        const bool audit = MODE == MLP_MODE_LIST && (entry >> 31) != 0;

        f32x16 E[2];
        encode_point<NERF_FAST_SINCOS != 0>(px, py, pz, h, E);

        f32x16 X[8], Y[8];

        // dense0: 64 slots -> 256, output X
        load_bias<8>(X, small + kBiasOff + 0 * 256, h);
        tile_steps<8, false>(E[0], X, P);
        tile_steps<8, false>(E[1], X, P);

        float sigma = 0.f;
        // dense1..4
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
        // dense5: [encoding ; h4] -> 256 (skip connection, src/network.rs:209-210)
        load_bias<8>(Y, small + kBiasOff + 5 * 256, h);
        tile_steps<8, false>(E[0], Y, P);
        tile_steps<8, false>(E[1], Y, P);
#pragma unroll
        for (int t = 0; t < 8; ++t) tile_steps<8, true>(X[t], Y, P);
        // dense6, dense7
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);
        const float pre = alpha_pre(Y, small, h);
        sigma = fmaxf(pre, 0.f);
        if (valid && h == 0) A.sigma_out[i] = audit ? pre : sigma;
        if (FULL && A.skip_empty) {
            // Empty-tile skip (exact): if sigma == 0 for all 128 points of this workgroup's tile, then
            // alpha = 1 - exp(-0 * delta) = 0 and w = T * 0 = 0 exactly for each of them (src/lib.rs:271-272), so their
            // colours never reach a pixel: skip bottleneck + viewdirs + rgb (17 % of a full evaluation), write rgb = 0.
            LDS_AS int *vote = (LDS_AS int *)(lds + kRingSlots * kChunkBytes) + kMiscOff + 8;
            const bool any_wg = tile_has_density(vote, valid && sigma > 0.0f, wave, lane);
            if (!any_wg) {
                if (valid && h == 0) {
                    A.rgb_out[3 * (size_t)i + 0] = 0.f; A.rgb_out[3 * (size_t)i + 1] = 0.f; A.rgb_out[3 * (size_t)i + 2] = 0.f;
                }
                if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 1ull);
                pipe_restart(P);
                continue;
            }
        }
        if (FULL) hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck (no activation, :218)

        if (FULL) {
            f32x16 D;
            encode_dir<NERF_FAST_SINCOS != 0>(dx, dy, dz, h, D);
            // viewdirs: [bottleneck ; dir encoding] -> 128 (src/network.rs:219-222)
            f32x16 V[4];
            load_bias<4>(V, small + kBiasViewOff, h);
#pragma unroll
            for (int t = 0; t < 8; ++t) tile_steps<4, false>(X[t], V, P);
            tile_steps<4, false>(D, V, P);
            float c[3];
            rgb_head(V, small, h, c);
            if (valid && h == 0) {
                A.rgb_out[3 * (size_t)i + 0] = c[0];
                A.rgb_out[3 * (size_t)i + 1] = c[1];
                A.rgb_out[3 * (size_t)i + 2] = c[2];
            }
        }
    }
    // drain the (unused) prefetches before the LDS allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (A.clock_out && tid == 0) { // diagnostic: shader clock = d(memtime) / d(memrealtime) x 100 MHz
        A.clock_out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        A.clock_out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((nerf_mlp_kernel<FULL, MODE>), dim3(n_blocks), dim3(256), kLdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t nerf_mlp_init() {
    // forward_batch always evaluates the full head, so the sigma-only kernel exists in ray mode only
    const void *ks[5] = {(const void *)nerf_mlp_kernel<true, MLP_MODE_POINTS>, (const void *)nerf_mlp_kernel<true, MLP_MODE_RAYS>,
                         (const void *)nerf_mlp_kernel<false, MLP_MODE_RAYS>, (const void *)nerf_mlp_kernel<true, MLP_MODE_LIST>,
                         (const void *)nerf_mlp_kernel<false, MLP_MODE_LIST>};
    for (int i = 0; i < 5; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t nerf_mlp_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_LIST) {
        if (!a.point_list || !a.point_list_count) return hipErrorInvalidValue;
        return full ? launch_t<true, MLP_MODE_LIST>(a, n_blocks, stream) : launch_t<false, MLP_MODE_LIST>(a, n_blocks, stream);
    }
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : hipErrorInvalidValue; // no sigma-only forward_batch
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
