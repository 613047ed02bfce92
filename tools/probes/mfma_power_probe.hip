// mfma_power_probe.hip -- what the 16-bit matrix cores of THIS MI355X deliver on random operands, with nothing else in the loop.
//
// The 16-bit MLP kernels (bf16v2, bf16x3, f16x2) sit at ~0.59-0.67 of the nominal 2.5 PFLOP/s while their in-kernel clock falls to
// ~2.0 GHz.  This probe prices the ceiling: bare MFMA loops (operands in registers, one wave per SIMD, one workgroup per CU,
// random data so that the datapath toggles) for both shapes and both 16-bit types, plus the same loops with the A operand re-read
// from LDS every step (what a weight-streaming kernel has to do).  Reports TFLOP/s, fraction of 2.5 PF and the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz; MI355X_MICROARCH.md "DVFS give-back" item 6).
//
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip ; run: ./mfma_power_probe [seconds per variant]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum { V_BF16_32 = 0, V_BF16_16 = 1, V_F16_32 = 2, V_F16_16 = 3 };

template <int V> struct Op;
template <> struct Op<V_BF16_32> { using T = bf16x8; static constexpr bool k32 = true; };
template <> struct Op<V_BF16_16> { using T = bf16x8; static constexpr bool k32 = false; };
template <> struct Op<V_F16_32> { using T = f16x8; static constexpr bool k32 = true; };
template <> struct Op<V_F16_16> { using T = f16x8; static constexpr bool k32 = false; };

// Inline asm with the accumulators tied in place ("+v": VGPR-form accumulators): the builtin form let hipcc rotate the 16 small
// accumulator tiles through AGPRs with ~100 v_accvgpr moves per iteration, which measured the moves, not the matrix pipe.
__device__ __forceinline__ void mm32(bf16x8 a, bf16x8 b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mm32(f16x8 a, f16x8 b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mm16(bf16x8 a, bf16x8 b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mm16(f16x8 a, f16x8 b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

// Output tile per wave is 64 x 64 in both shapes: 2 x 2 tiles of 32x32 (4 MFMAs of 32 768 FLOP) or 4 x 4 tiles of 16x16 (16 MFMAs of
// 16 384 FLOP) per k-step: 131 072 / 262 144 FLOP per step.  NK k-steps of operands are held in registers and cycled through.
// LDSA: the A fragments come from LDS (ds_read_b128, lane-linear) every step instead of from registers.
template <int V, bool LDSA>
__global__ __launch_bounds__(256, 1) void probe(const u32x4 *src, float *sink, unsigned long long *stamps, int iters) {
    using T = typename Op<V>::T;
    constexpr bool K32 = Op<V>::k32;
    constexpr int NA = K32 ? 2 : 4, NK = 4;
    __shared__ u32x4 lds[NK * 4 * 64 * 4]; // [k][frag][lane] per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T a[NK][NA], b[NK][NA];
    for (int k = 0; k < NK; ++k)
        for (int i = 0; i < NA; ++i) {
            const u32x4 va = src[((blockIdx.x * 4 + wave) * NK * 8 + k * 8 + i) * 64 + lane];
            const u32x4 vb = src[((blockIdx.x * 4 + wave) * NK * 8 + k * 8 + 4 + i) * 64 + lane];
            a[k][i] = __builtin_bit_cast(T, va);
            b[k][i] = __builtin_bit_cast(T, vb);
            lds[((wave * NK + k) * 4 + i) * 64 + lane] = va;
        }
    __syncthreads();
    f32x16 c32[2][2];
    f32x4 c16[4][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c32[i][j][r] = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) c16[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            T av[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (LDSA) av[i] = __builtin_bit_cast(T, lds[((wave * NK + k) * 4 + i) * 64 + lane]);
                else av[i] = a[k][i];
            }
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    if constexpr (K32) mm32(av[i], b[k][j], c32[i][j]);
                    else mm16(av[i], b[k][j], c16[i][j]);
                }
        }
        // keep the accumulators bounded without leaving the matrix pipe idle for long: nothing (f32 does not overflow in this many steps
        // with operands of magnitude <= 2^-4: |sum| <= iters * NK * 16 * 2^-8)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += c32[i][j][r];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += c16[i][j][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint16_t rnd16(uint32_t &st, bool bf) {
    st = st * 1664525u + 1013904223u;
    // random sign, exponent in [2^-8, 2^-4), random mantissa: every operand bit toggles
    const uint32_t sign = (st >> 31) & 1u, e = (st >> 28) & 3u, man = (st >> 8);
    if (bf) return (uint16_t)((sign << 15) | ((127u - 8u + e) << 7) | (man & 0x7fu));
    return (uint16_t)((sign << 15) | ((15u - 8u + e) << 10) | (man & 0x3ffu));
}

template <int V, bool LDSA>
static void run(const char *name, int n_cus, double seconds, int data) { // data: 0 random, 1 zeros, 2 = A random (weights), B like ReLU outputs (half of them 0, the rest positive)
    const bool zeros = data == 1;
    constexpr bool K32 = Op<V>::k32;
    const bool bf = (V == V_BF16_32 || V == V_BF16_16);
    const size_t n_vec = (size_t)n_cus * 4 * 4 * 8 * 64;
    std::vector<uint16_t> h(n_vec * 8);
    uint32_t st = 12345u + V;
    for (size_t i = 0; i < h.size(); ++i) {
        uint16_t x = zeros ? 0 : rnd16(st, bf);
        if (data == 2 && ((i / 8 / 64) % 8) >= 4) { // a B fragment (src layout: [..][k][8 frags: 4 A then 4 B][64 lanes][8 values])
            st = st * 1664525u + 1013904223u;
            x = (st >> 30) & 1u ? (uint16_t)(x & 0x7fffu) : (uint16_t)0;
        }
        h[i] = x;
    }
    u32x4 *d_src; float *d_sink; unsigned long long *d_st;
    CK(hipMalloc((void **)&d_src, n_vec * 16)); CK(hipMalloc((void **)&d_sink, (size_t)n_cus * 256 * 4)); CK(hipMalloc((void **)&d_st, (size_t)n_cus * 16));
    CK(hipMemcpy(d_src, h.data(), n_vec * 16, hipMemcpyHostToDevice));
    const int iters = 20000; // x 4 k-steps
    const double flop = (double)n_cus * 4 * iters * 4 * (K32 ? 4 * 32768.0 : 16 * 16384.0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm up for `seconds`, then time the same number of launches again
    probe<V, LDSA><<<n_cus, 256>>>(d_src, d_sink, d_st, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); probe<V, LDSA><<<n_cus, 256>>>(d_src, d_sink, d_st, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms1 = 0; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int reps = std::max(2, (int)(seconds * 1000.0 / ms1));
    for (int r = 0; r < reps; ++r) probe<V, LDSA><<<n_cus, 256>>>(d_src, d_sink, d_st, iters);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) probe<V, LDSA><<<n_cus, 256>>>(d_src, d_sink, d_st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> sv((size_t)n_cus * 2);
    CK(hipMemcpy(sv.data(), d_st, sv.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (int i = 0; i < n_cus; ++i) if (sv[2 * i + 1]) mhz.push_back((double)sv[2 * i] / (double)sv[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double tf = flop * reps / (ms * 1e-3) / 1e12;
    const double cyc_per_step = mhz.empty() ? 0 : (ms * 1e-3 / reps) * mhz[mhz.size() / 2] * 1e6 / ((double)iters * 4);
    printf("{\"variant\": \"%s\", \"data\": \"%s\", \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"clock_mhz_median\": %.0f, \"cycles_per_kstep\": %.1f, \"ms_per_launch\": %.2f, \"launches\": %d}\n",
           name, zeros ? "zeros" : data == 2 ? "A random, B relu-like (half zeros, positive)" : "random", tf, tf / 2500.0, mhz.empty() ? 0.0 : mhz[mhz.size() / 2], cyc_per_step, ms / reps, reps);
    fflush(stdout);
    CK(hipFree(d_src)); CK(hipFree(d_sink)); CK(hipFree(d_st));
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, n_cus);
    run<V_BF16_32, false>("bf16 32x32x16 regs", n_cus, seconds, 0);
    run<V_BF16_16, false>("bf16 16x16x32 regs", n_cus, seconds, 0);
    run<V_F16_32, false>("f16 32x32x16 regs", n_cus, seconds, 0);
    run<V_F16_16, false>("f16 16x16x32 regs", n_cus, seconds, 0);
    run<V_BF16_32, true>("bf16 32x32x16 A from LDS", n_cus, seconds, 0);
    run<V_BF16_16, true>("bf16 16x16x32 A from LDS", n_cus, seconds, 0);
    run<V_F16_32, true>("f16 32x32x16 A from LDS", n_cus, seconds, 0);
    run<V_F16_16, true>("f16 16x16x32 A from LDS", n_cus, seconds, 0);
    run<V_BF16_32, true>("bf16 32x32x16 A from LDS", n_cus, seconds, 2);
    run<V_BF16_16, true>("bf16 16x16x32 A from LDS", n_cus, seconds, 2);
    run<V_F16_32, true>("f16 32x32x16 A from LDS", n_cus, seconds, 2);
    run<V_F16_16, true>("f16 16x16x32 A from LDS", n_cus, seconds, 2);
    run<V_BF16_32, false>("bf16 32x32x16 regs", n_cus, seconds, 1);
    run<V_BF16_16, false>("bf16 16x16x32 regs", n_cus, seconds, 1);
    return 0;
}
