// Power/clock probe: the same bf16 FLOPs per wave as v_mfma_f32_32x32x16_bf16 (variant 0) or v_mfma_f32_16x16x32_bf16
// (variant 1), operands re-read from LDS (3 ds_read_b128 per 6 / 12 MFMAs, as in mlp_kernel_bf16x3.hip), random data.
// Prints achieved PFLOP/s and the in-kernel clock.  hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void probe(const u32x4 *src, float *out, unsigned long long *clk, int iters) {
    __shared__ u32x4 lds[4096]; // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 b1 = src[lane], b2 = src[64 + lane], b3 = src[128 + lane];
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int base = ((it * 8 + u) * 192) & 4095;
            const u32x4 a1 = lds[(base + lane) & 4095], a2 = lds[(base + 64 + lane) & 4095], a3 = lds[(base + 128 + lane) & 4095];
            const bf16x8 A1 = __builtin_bit_cast(bf16x8, a1), A2 = __builtin_bit_cast(bf16x8, a2), A3 = __builtin_bit_cast(bf16x8, a3);
            const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1), B2 = __builtin_bit_cast(bf16x8, b2), B3 = __builtin_bit_cast(bf16x8, b3);
            if (VARIANT == 0) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[u], 0, 0, 0);
            } else {
                f32x4 p = {acc[u][0], acc[u][1], acc[u][2], acc[u][3]}, q = {acc[u][4], acc[u][5], acc[u][6], acc[u][7]};
#pragma unroll
                for (int k = 0; k < 2; ++k) { // the same FLOPs: 2 x 16x16x32 per 32x32x16
                    f32x4 &c = k ? q : p;
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B3, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, c, 0, 0, 0);
                }
                acc[u][0] = p[0]; acc[u][1] = p[1]; acc[u][2] = p[2]; acc[u][3] = p[3];
                acc[u][4] = q[0]; acc[u][5] = q[1]; acc[u][6] = q[2]; acc[u][7] = q[3];
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int iters = 40000, nb = 256;
    std::vector<unsigned int> h(4096 * 4);
    srand(1);
    for (auto &v : h) { // random bf16 pairs of moderate magnitude
        unsigned short a = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)), b = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
        v = (unsigned int)a | ((unsigned int)b << 16);
    }
    u32x4 *src; float *out; unsigned long long *clk;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, nb * 256 * 4); hipMalloc(&clk, nb * 16);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(256), 0, 0, src, out, clk, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(256), 0, 0, src, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(nb * 2);
            hipMemcpy(c.data(), clk, nb * 16, hipMemcpyDeviceToHost);
            const double mhz = 100.0 * (double)c[0] / (double)c[1];
            const double flops = (double)nb * 4 * iters * 8 * 6 * 32768.0; // 6 x 32x32x16 per unit, 8 units per iteration
            printf("variant %d (%s): %.1f ms  %.3f PFLOP/s  clock %.0f MHz  cycles per 32x32x16-equivalent %.1f\n", variant,
                   variant ? "16x16x32" : "32x32x16", ms, flops / (ms * 1e-3) / 1e15, mhz, (double)c[0] / ((double)iters * 48));
        }
    return 0;
}
