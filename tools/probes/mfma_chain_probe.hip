// mfma_chain_probe.hip -- issue cost of DEPENDENT 16-bit MFMAs: how many independent accumulation chains does a wave need before
// v_mfma_f32_32x32x16_bf16 (8 passes, 32 cycles) and v_mfma_f32_16x16x32_bf16 (4 passes, 16 cycles) issue back to back?
// The bf16 kernel (mlp_kernel_bf16v2.hip) runs exactly TWO chains per wave (acc0 += A B0, acc1 += A B1, output-tile-major).
// One wave per SIMD, operands in registers, nothing else in the loop; shader cycles from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip ; run: ./mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void mm32(bf16x8 a, bf16x8 b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mmf32(float a, float b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mm16(bf16x8 a, bf16x8 b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

// the f32 shape of the headline kernel (16 passes, 64 cycles); ACC: accumulators in AGPRs ("+a") as in the f32 kernels
template <int NCH, bool ACC>
__global__ __launch_bounds__(256, 1) void probe_f32(const u32x4 *src, float *sink, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = __builtin_bit_cast(float, src[lane][0]), b = __builtin_bit_cast(float, src[64 + lane][0]);
    f32x16 c[NCH];
    for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if constexpr (ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[i]) : "v"(a), "v"(b));
                else mmf32(a, b, c[i]);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH, bool ACC>
static void run_f32(const u32x4 *d_src, float *d_sink, unsigned long long *d_cyc, int blocks) {
    const int iters = 10000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe_f32<NCH, ACC>), dim3(blocks), dim3(256), 0, 0, d_src, d_sink, d_cyc, iters);
        if (hipDeviceSynchronize() != hipSuccess) exit(1);
    }
    unsigned long long c = 0;
    if (hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) exit(1);
    printf("{\"shape\": \"32x32x2_f32 (%s accumulators)\", \"chains\": %d, \"cycles_per_mfma\": %.2f, \"ideal\": 64}\n", ACC ? "AGPR" : "VGPR", NCH,
           (double)c / ((double)iters * 8 * NCH));
}

template <int NCH, bool BIG>
__global__ __launch_bounds__(256, 1) void probe(const u32x4 *src, float *sink, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    const bf16x8 a = __builtin_bit_cast(bf16x8, src[lane]), b = __builtin_bit_cast(bf16x8, src[64 + lane]);
    f32x16 c32[NCH];
    f32x4 c16[NCH];
    for (int i = 0; i < NCH; ++i) { for (int r = 0; r < 16; ++r) c32[i][r] = 0.f; for (int r = 0; r < 4; ++r) c16[i][r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if constexpr (BIG) mm32(a, b, c32[i]); else mm16(a, b, c16[i]);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) { for (int r = 0; r < 16; ++r) s += c32[i][r]; for (int r = 0; r < 4; ++r) s += c16[i][r]; }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH, bool BIG>
static void run(const u32x4 *d_src, float *d_sink, unsigned long long *d_cyc, int blocks) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe<NCH, BIG>), dim3(blocks), dim3(256), 0, 0, d_src, d_sink, d_cyc, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<NCH, BIG>), dim3(blocks), dim3(256), 0, 0, d_src, d_sink, d_cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long c = 0;
    CK(hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
    const double per = (double)c / ((double)iters * 8 * NCH);
    printf("{\"shape\": \"%s\", \"chains\": %d, \"cycles_per_mfma\": %.2f, \"ideal\": %d}\n", BIG ? "32x32x16_bf16" : "16x16x32_bf16", NCH, per, BIG ? 32 : 16);
}

int main() {
    const int blocks = 256;
    u32x4 *d_src; float *d_sink; unsigned long long *d_cyc;
    CK(hipMalloc(&d_src, 128 * sizeof(u32x4)));
    CK(hipMemset(d_src, 0, 128 * sizeof(u32x4))); // all-zero operands: the clock stays up, cycles are what is measured anyway
    CK(hipMalloc(&d_sink, blocks * 256 * sizeof(float)));
    CK(hipMalloc(&d_cyc, blocks * sizeof(unsigned long long)));
    run<1, true>(d_src, d_sink, d_cyc, blocks);
    run<2, true>(d_src, d_sink, d_cyc, blocks);
    run<3, true>(d_src, d_sink, d_cyc, blocks);
    run<4, true>(d_src, d_sink, d_cyc, blocks);
    run<1, false>(d_src, d_sink, d_cyc, blocks);
    run<2, false>(d_src, d_sink, d_cyc, blocks);
    run<4, false>(d_src, d_sink, d_cyc, blocks);
    run<8, false>(d_src, d_sink, d_cyc, blocks);
    run_f32<1, false>(d_src, d_sink, d_cyc, blocks);
    run_f32<2, false>(d_src, d_sink, d_cyc, blocks);
    run_f32<1, true>(d_src, d_sink, d_cyc, blocks);
    run_f32<2, true>(d_src, d_sink, d_cyc, blocks);
    run_f32<4, true>(d_src, d_sink, d_cyc, blocks);
    return 0;
}
