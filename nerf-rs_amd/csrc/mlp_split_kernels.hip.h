// mlp_split_kernels.hip.h -- the kernel bodies shared by the "f32 by operand splitting" arithmetics: bf16x3
// (mlp_kernel_bf16x3.hip: three bf16 parts, six products) and f16x2 (mlp_kernel_f16x2.hip: two f16 parts, three products).
// Textually included at the end of each of those files, after the file has defined, in an anonymous namespace, its
// primitives -- PipeS (weight-stream pipeline) with pipe_start(), BS (split B operand), prep_all, k_step, range_check, kCB / kRS,
// kSplitChunksSigma / kSplitChunksFull, kSplitLdsBytes, kSplitWaveBytes (the layers built from them -- tile_steps, eight_tiles,
// hidden_layer, load_bias, alpha_head -- are defined here, once) --
// and the SPLIT_* names of the kernels and host functions it instantiates here:
//   SPLIT_KERNEL_FUSED<FULL, MODE>   the fused MLP (forward_batch + point fill), structure of mlp_kernel.hip
//   SPLIT_KERNEL_TRUNK<EXPORT>       skip_dead: ray-sequential trunk     } scheme: mlp_kernel_seq.hip,
//   SPLIT_KERNEL_COLOUR              skip_dead: compacted colour head    } shared half: mlp_seq_common.hip.h
// The accumulator tiles have the f32 kernel's register layout (C/D of the 32x32 MFMAs), so the small parameters and the
// exported h8 tiles are shared with it.

// ---- layers on top of the arithmetic's k_step / prep_all (identical for every split arithmetic) ---------------------------------
namespace {

// One input tile (two k-steps) of a layer with NT output tiles.  `b` holds the split B of this tile's k-step 0 on entry and
// of the next tile's k-step 0 on exit (if HAS_NEXT).  Every tile starts on a chunk boundary: a k-step is 8 units = one chunk in
// the 8-tile layers and 4 units in the 4-tile viewdirs layer, where the tile's second k-step therefore starts at unit 4.
template <int NT, bool RELU, bool ACC_IN, bool HAS_NEXT, bool NRELU, bool NACC_IN>
__device__ __forceinline__ void tile_steps(f32x16 &in, f32x16 &nin, f32x16 (&out)[8], BS &b, PipeS &P) {
    if constexpr (ACC_IN) asm volatile("" : "+a"(in));
    BS b1;
    k_step<NT, 0, true, RELU, 1>(out, b, in, b1, P);
    if constexpr (HAS_NEXT && NACC_IN) asm volatile("" : "+a"(nin));
    k_step<NT, (NT == 8 ? 0 : 4), HAS_NEXT, NRELU, 0>(out, b1, nin, b, P);
    // keep every accumulation chain in program order (hipcc otherwise defers whole chains across the sched_barriers)
    if constexpr (NT == 8)
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]), "+a"(out[4]), "+a"(out[5]), "+a"(out[6]), "+a"(out[7]));
    else
        asm volatile("" : "+a"(out[0]), "+a"(out[1]), "+a"(out[2]), "+a"(out[3]));
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

// the 8 input tiles of a 256-wide activation; `b` = split B of in[0]'s k-step 0 on entry
template <int NT, bool RELU>
__device__ __forceinline__ void eight_tiles(f32x16 (&in)[8], f32x16 (&out)[8], BS &b, PipeS &P) {
    tile_steps<NT, RELU, true, true, RELU, true>(in[0], in[1], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[1], in[2], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[2], in[3], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[3], in[4], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[4], in[5], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[5], in[6], out, b, P);
    tile_steps<NT, RELU, true, true, RELU, true>(in[6], in[7], out, b, P);
}

template <bool RELU>
__device__ __forceinline__ void hidden_layer(f32x16 (&in)[8], f32x16 (&out)[8], const LDS_AS float *bias, PipeS &P, int h) {
    load_bias<8>(out, bias, h);
    BS b;
    asm volatile("" : "+a"(in[0]));
    prep_all<RELU, 0>(in[0], b, P);
    eight_tiles<8, RELU>(in, out, b, P);
    tile_steps<8, RELU, true, false, false, false>(in[7], in[7], out, b, P);
}

// `nonfinite` (optional device counter): points whose density pre-activation is NaN / inf.  With NERF_MLP_F16X2 an activation beyond
// the f16 range splits into (inf, -inf) and turns into NaN in the next layer; fmaxf below would silently return 0 for it.  One v_cmp
// per tile makes that observable (nerf_stats.n_nonfinite_points; nerf_forward_batch_ex fails with NERF_ERR_STATE).
constexpr float kUncertainZeroMargin = 4e-5f; // 2 x the absolute part of k_resample's density-error bound (sampling_kernels.hip)
__device__ __forceinline__ float alpha_head(const f32x16 (&Y)[8], const LDS_AS float *small, int h, unsigned int *nonfinite, bool valid, float *pre_out = nullptr) {
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
    const float pre = xhalf_sum((a0 + a1) + (a2 + a3)) + small[kMiscOff + 0];
    if (pre_out) *pre_out = pre;
    if (nonfinite) {
        const unsigned long long bad = __ballot(valid && !(fabsf(pre) <= 3.0e38f)) & 0xffffffffull; // one lane-half per point
        if (bad && (threadIdx.x & 63) == 0) atomicAdd(nonfinite, (unsigned)__popcll(bad));
    }
    // An "uncertain zero": the density is 0 here, but the pre-activation sits within the split arithmetics' own error of 0, so the f32
    // kernel may see a tiny positive density where this one sees none.  It is returned as -0.0f: every consumer treats it as 0
    // (-(-0) delta = 0, alpha = 0, sigma > 0 false), and hybrid sampling's flag (k_resample) gives the sample an error bound instead
    // of trusting the zero -- a 255 M-ray fuzz found one all-empty ray whose f32 twin had one density of 4e-7 (tools/fuzz_hybrid_flags.py).
    return (pre <= 0.f && pre > -kUncertainZeroMargin) ? -0.0f : fmaxf(pre, 0.f);
}

} // namespace

template <bool FULL, int MODE>
__global__ __launch_bounds__(256, 1) void SPLIT_KERNEL_FUSED(const MlpArgs A) {
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
    PipeS P;
    P.lane16 = lane * 16;
    P.ring_lane = lds + P.lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * kSplitWaveBytes; // this wave's six pieces of a chunk
    P.stream_bytes = (FULL ? kSplitChunksFull : kSplitChunksSigma) * kCB;
    P.gbase = (const char *)A.wstream + wave * kSplitWaveBytes;
    __syncthreads();
    pipe_start(P);
    uint64_t clk0 = 0, rt0 = 0;
    if (A.clock_out) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    const int n_points = MODE == MLP_MODE_LIST ? list_length(A) : A.n_points; // list mode (certify_zero): the length lives on the device
    const int n_tiles = (n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    RawIn nxt = load_raw<MODE>(A, blockIdx.x, wave, p);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int slot = tile * kPointsPerBlock + wave * kPointsPerWave + p;
        const bool valid = slot < n_points;
        const RawIn in = nxt;
        nxt = load_raw<MODE>(A, tile + gridDim.x, wave, p);
        float px, py, pz;
        point_of<MODE>(A, in, px, py, pz);
        const float dx = in.dx, dy = in.dy, dz = in.dz;
        const unsigned entry = __builtin_bit_cast(unsigned, in.b);
        const size_t i = MODE == MLP_MODE_LIST ? (size_t)(entry & 0x7fffffffu) : (size_t)slot; // where the outputs go
        const bool audit = MODE == MLP_MODE_LIST && (entry >> 31) != 0; // an audited certificate: the raw pre-activation leaves the kernel (k_cert_audit)

        f32x16 E[2];
        encode_point<true>(px, py, pz, h, E);

        f32x16 X[8], Y[8];
        BS b;
        load_bias<8>(X, small + kBiasOff + 0 * 256, h);          // dense0 (src/network.rs:204)
        prep_all<false, 0>(E[0], b, P);
        tile_steps<8, false, false, true, false, false>(E[0], E[1], X, b, P);
        tile_steps<8, false, false, false, false, false>(E[1], E[1], X, b, P);
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
        load_bias<8>(Y, small + kBiasOff + 5 * 256, h);          // dense5 on [encoding ; h4] (:209-210)
        prep_all<false, 0>(E[0], b, P);
        tile_steps<8, false, false, true, false, false>(E[0], E[1], Y, b, P);
        tile_steps<8, false, false, true, true, true>(E[1], X[0], Y, b, P);
        eight_tiles<8, true>(X, Y, b, P);
        tile_steps<8, true, true, false, false, false>(X[7], X[7], Y, b, P);
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);

        float pre;
        const float sigma = alpha_head(Y, small, h, A.nonfinite, valid, &pre);
        if (valid && h == 0) A.sigma_out[i] = audit ? pre : sigma;
        if (!FULL) range_check(P, A.nonfinite, valid);

        if (FULL && A.skip_empty) { // exact empty-tile skip, see mlp_kernel.hip
            LDS_AS int *vote = (LDS_AS int *)(lds + kRS * kCB) + kMiscOff + 8;
            const bool any_wg = tile_has_density(vote, valid && sigma > 0.0f, wave, lane);
            if (!any_wg) {
                if (valid && h == 0) {
                    A.rgb_out[3 * (size_t)i + 0] = 0.f; A.rgb_out[3 * (size_t)i + 1] = 0.f; A.rgb_out[3 * (size_t)i + 2] = 0.f;
                }
                if (A.skip_counter && tid == 0) atomicAdd(A.skip_counter, 1ull);
                range_check(P, A.nonfinite, valid);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // in-flight chunks landed; ring idle
                pipe_start(P);
                continue;
            }
        }

        if (FULL) {
            hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck: no activation on its output (:218)
            f32x16 D;
            encode_dir<true>(dx, dy, dz, h, D);
            f32x16 (&V)[8] = Y;                                          // Y is dead after the bottleneck
            load_bias<4>(V, small + kBiasViewOff, h);
            asm volatile("" : "+a"(X[0]));
            prep_all<false, 0>(X[0], b, P);
            eight_tiles<4, false>(X, V, b, P);
            tile_steps<4, false, true, true, false, false>(X[7], D, V, b, P);
            tile_steps<4, false, false, false, false, false>(D, D, V, b, P);
            float c[3];
            rgb_head(V, small, h, c);
            if (valid && h == 0) {
                A.rgb_out[3 * (size_t)i + 0] = c[0];
                A.rgb_out[3 * (size_t)i + 1] = c[1];
                A.rgb_out[3 * (size_t)i + 2] = c[2];
            }
            range_check(P, A.nonfinite, valid);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (A.clock_out && tid == 0) { // diagnostic: shader clock = d(memtime) / d(memrealtime) x 100 MHz
        A.clock_out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        A.clock_out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

// ---- exact dead-sample skipping in this arithmetic (skip_dead with mlp_dtype = NERF_MLP_BF16X3; scheme: mlp_kernel_seq.hip,
// shared half: mlp_seq_common.hip.h): the ray-sequential trunk and the compacted colour head with this file's layers.  The
// accumulator tiles have the f32 kernel's register layout, so the exported h8 tiles and the small parameters are shared with it.
namespace {
__device__ __forceinline__ void pipe_begin(PipeS &P, const LDS_AS char *lds, int lane, int wave, const char *stream, int n_chunks) {
    P.lane16 = lane * 16;
    P.ring_lane = lds + P.lane16;
    P.ring_addr = (uint32_t)(uintptr_t)lds + wave * kSplitWaveBytes;
    P.stream_bytes = n_chunks * kCB;
    P.gbase = stream + wave * kSplitWaveBytes;
    __syncthreads();
    pipe_start(P);
}
} // namespace

template <bool EXPORT>
__global__ __launch_bounds__(256, 1) void SPLIT_KERNEL_TRUNK(const SeqArgs A) {
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
    PipeS P;
    pipe_begin(P, lds, lane, wave, (const char *)A.wstream, kSplitChunksSigma);

    RayWork W;
    work_init(W, A);
    while (work_acquire(W, A, vote, wave, lane)) {
        const ChunkIn c = chunk_inputs(W, A, p);
        f32x16 E[2];
        encode_point<true>(c.px, c.py, c.pz, h, E);
        f32x16 X[8], Y[8];
        BS b;
        load_bias<8>(X, small + kBiasOff + 0 * 256, h);
        prep_all<false, 0>(E[0], b, P);
        tile_steps<8, false, false, true, false, false>(E[0], E[1], X, b, P);
        tile_steps<8, false, false, false, false, false>(E[1], E[1], X, b, P);
        hidden_layer<true>(X, Y, small + kBiasOff + 1 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 2 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 3 * 256, P, h);
        hidden_layer<true>(Y, X, small + kBiasOff + 4 * 256, P, h);
        load_bias<8>(Y, small + kBiasOff + 5 * 256, h);
        prep_all<false, 0>(E[0], b, P);
        tile_steps<8, false, false, true, false, false>(E[0], E[1], Y, b, P);
        tile_steps<8, false, false, true, true, true>(E[1], X[0], Y, b, P);
        eight_tiles<8, true>(X, Y, b, P);
        tile_steps<8, true, true, false, false, false>(X[7], X[7], Y, b, P);
        hidden_layer<true>(Y, X, small + kBiasOff + 6 * 256, P, h);
        hidden_layer<true>(X, Y, small + kBiasOff + 7 * 256, P, h);
        const float sigma = alpha_head(Y, small, h, A.nonfinite, c.valid);
        range_check(P, A.nonfinite, c.valid);
        chunk_finish<EXPORT>(W, A, c, sigma, Y, lane, p, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    work_done(W, A, lane);
}

__global__ __launch_bounds__(256, 1) void SPLIT_KERNEL_COLOUR(const ColourArgs A) {
    using namespace mlpseq;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LDS_AS char *lds = (const LDS_AS char *)smem;
    const LDS_AS float *small = (const LDS_AS float *)(lds + kRS * kCB);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    {
        float *dst = (float *)(smem + kRS * kCB);
        for (int i = tid; i < kSmallFloats; i += 256) dst[i] = A.small_params[i];
    }
    PipeS P;
    pipe_begin(P, lds, lane, wave, (const char *)A.wstream + (size_t)kSplitChunksSigma * kCB, kSplitChunksFull - kSplitChunksSigma);

    unsigned n_live;
    const int n_tiles = colour_tiles(A, &n_live);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x16 X[8], Y[8];
        const ColourIn c = colour_inputs(A, n_live, tile, wave, lane, Y);
        hidden_layer<true>(Y, X, small + kBiasOff + 8 * 256, P, h); // bottleneck
        f32x16 D;
        encode_dir<true>(c.dx, c.dy, c.dz, h, D);
        f32x16 (&V)[8] = Y;
        load_bias<4>(V, small + kBiasViewOff, h);
        BS b;
        asm volatile("" : "+a"(X[0]));
        prep_all<false, 0>(X[0], b, P);
        eight_tiles<4, false>(X, V, b, P);
        tile_steps<4, false, true, true, false, false>(X[7], D, V, b, P);
        tile_steps<4, false, false, false, false, false>(D, D, V, b, P);
        float rgb[3];
        rgb_head(V, small, h, rgb);
        colour_store(A, c, rgb, h);
        range_check(P, A.nonfinite, c.valid);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t SPLIT_FN_SEQ_INIT() {
    const void *ks[3] = {(const void *)SPLIT_KERNEL_TRUNK<true>, (const void *)SPLIT_KERNEL_TRUNK<false>, (const void *)SPLIT_KERNEL_COLOUR};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t SPLIT_FN_TRUNK_LAUNCH(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream) {
    if (a.n_rays <= 0 || a.samples_per_ray <= 0) return hipSuccess;
    n_blocks = mlpseq::trunk_blocks(a, n_blocks);
    if (export_live) hipLaunchKernelGGL(SPLIT_KERNEL_TRUNK<true>, dim3(n_blocks), dim3(256), kSplitLdsBytes, stream, a);
    else hipLaunchKernelGGL(SPLIT_KERNEL_TRUNK<false>, dim3(n_blocks), dim3(256), kSplitLdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t SPLIT_FN_COLOUR_LAUNCH(const ColourArgs &a, int n_blocks, hipStream_t stream) {
    if (n_blocks < 1) n_blocks = 1;
    hipLaunchKernelGGL(SPLIT_KERNEL_COLOUR, dim3(n_blocks), dim3(256), kSplitLdsBytes, stream, a);
    return hipGetLastError();
}

template <bool FULL, int MODE>
static hipError_t launch_t(const MlpArgs &a, int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((SPLIT_KERNEL_FUSED<FULL, MODE>), dim3(n_blocks), dim3(256), kSplitLdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t SPLIT_FN_INIT() {
    // forward_batch always evaluates the full head, so the sigma-only kernel exists in ray mode only
    const void *ks[4] = {(const void *)SPLIT_KERNEL_FUSED<true, MLP_MODE_POINTS>, (const void *)SPLIT_KERNEL_FUSED<true, MLP_MODE_RAYS>,
                         (const void *)SPLIT_KERNEL_FUSED<false, MLP_MODE_RAYS>, (const void *)SPLIT_KERNEL_FUSED<true, MLP_MODE_LIST>};
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t SPLIT_FN_LAUNCH(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream) {
    if (a.n_points <= 0) return hipSuccess;
    const int n_tiles = (a.n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    if (n_blocks > n_tiles) n_blocks = n_tiles;
    if (n_blocks < 1) n_blocks = 1;
    if (a.mode == MLP_MODE_LIST) // certify_zero: the fine pass on the listed samples (the coarse pass of a split render is the f32 kernel's)
        return full && a.point_list && a.point_list_count ? launch_t<true, MLP_MODE_LIST>(a, n_blocks, stream) : hipErrorInvalidValue;
    if (a.mode == MLP_MODE_POINTS)
        return full ? launch_t<true, MLP_MODE_POINTS>(a, n_blocks, stream) : hipErrorInvalidValue; // no sigma-only forward_batch
    return full ? launch_t<true, MLP_MODE_RAYS>(a, n_blocks, stream) : launch_t<false, MLP_MODE_RAYS>(a, n_blocks, stream);
}
