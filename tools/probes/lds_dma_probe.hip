// lds_dma_probe.hip -- what does one LDS-DMA piece (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB, L2 -> LDS) cost a wave that is
// busy issuing MFMAs?  Three findings of round 3 point at it: the 16x16x32 bf16 kernel loses 12 % to its LDS-DMA, a colour pass of
// the f32 skip_dead kernel pays ~220 cycles of DMA issue per 16-KiB chunk, the bf16 kernel ~10 k of 104 k cycles per tile.
// One workgroup of 4 waves per CU (one wave per SIMD), source 2 MiB (L2-resident), destination a 48-KiB LDS ring; per iteration a
// wave issues 8 MFMAs (v_mfma_f32_32x32x16_bf16, 32 cycles each, zero operands) and P DMA pieces between them, with at most 8 pieces in
// flight (s_waitcnt vmcnt).  Shader cycles per iteration from s_memtime; the difference to P = 0 is what the pieces cost.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip ; run: ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void mm32(bf16x8 a, bf16x8 b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

// W = 16: dwordx4 (one instruction per KiB); W = 4: dword (four instructions per KiB, the same bytes); W = 0: no LDS-DMA at all --
// register-staged (global_load_dwordx4 into VGPRs, ds_write_b128 one iteration later)
template <int W>
__device__ __forceinline__ void glds(uint32_t lane_off, const char *gsrc, uint32_t dst) {
    uint32_t keep;
    if constexpr (W == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gsrc), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gsrc), "s"(dst) : "memory");
}

// P pieces of 1 KiB per 8 MFMAs; MF = false: no MFMAs at all (raw issue rate of the pieces); READS: each wave also reads 8 x 1 KiB
// of the ring per iteration with ds_read_b128 (what the kernels' A-operand fetch does beside the DMA)
template <int P, int W, bool MF, bool READS>
__global__ __launch_bounds__(256, 1) void probe(const char *src, float *sink, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bf16x8 a = {}, b = {};
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    u32x4 acc = {0, 0, 0, 0};
    u32x4 stage[P > 0 ? P : 1];
    for (int i = 0; i < (P > 0 ? P : 1); ++i) stage[i] = acc;
    const uint32_t lane_off = lane * (W == 4 ? 4 : 16);
    const uint32_t ring = (uint32_t)(uintptr_t)lds;
    uint32_t goff = (blockIdx.x * 4 + wave) * 4096u, slot = wave * 1024u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MF) { mm32(a, b, c0); }
            if (W == 0 && j < P) { // register-staged: global_load_dwordx4 -> VGPRs now, ds_write_b128 of it one iteration later
                if (it > 0) *(u32x4 *)(lds + ((slot + lane * 16u) % 49152u)) = stage[j < P ? j : 0];
                stage[j < P ? j : 0] = *(const u32x4 *)(src + ((goff & 0x1fffffu) + lane * 16u));
                goff += 16384u; slot = (slot + 4096u) % 49152u;
            } else if (j < P) {
                constexpr int NI = W == 16 ? 1 : 4;
#pragma unroll
                for (int q = 0; q < NI; ++q) glds<W>(lane_off, src + ((goff + q * 256u) & 0x1fffffu), ring + ((slot + q * 256u) % 49152u));
                goff += 16384u; slot = (slot + 4096u) % 49152u;
            }
            if (READS) { const u32x4 v = *(const u32x4 *)(lds + ((slot + j * 4096u + lane * 16u) % 49152u)); acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
            if (MF) { mm32(a, b, c1); }
        }
        if (P > 0 && W != 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < (P > 0 ? P : 1); ++i) acc[0] ^= stage[i][0] ^ stage[i][3];
    float s = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int P, int W, bool MF, bool READS>
static double run(const char *d_src, float *d_sink, unsigned long long *d_cyc) {
    const int iters = 4000, blocks = 256;
    CK(hipFuncSetAttribute((const void *)probe<P, W, MF, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<P, W, MF, READS>), dim3(blocks), dim3(256), 49152, 0, d_src, d_sink, d_cyc, iters);
        CK(hipDeviceSynchronize());
    }
    unsigned long long c[256];
    CK(hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)c[i];
    return m / blocks / iters;
}

int main() {
    char *d_src; float *d_sink; unsigned long long *d_cyc;
    CK(hipMalloc(&d_src, (2u << 20) + 8192)); // + slack: the last piece starts at 2 MiB - 256
    CK(hipMemset(d_src, 0, (2u << 20) + 8192));
    CK(hipMalloc(&d_sink, 256 * 256 * sizeof(float)));
    CK(hipMalloc(&d_cyc, 256 * sizeof(unsigned long long)));
    const double base = run<0, 16, true, false>(d_src, d_sink, d_cyc);
    printf("{\"variant\": \"16 MFMAs, no DMA\", \"cycles_per_iteration\": %.1f, \"ideal\": 512}\n", base);
#define ROW(P, W, MF, RD, name) { const double v = run<P, W, MF, RD>(d_src, d_sink, d_cyc); \
    printf("{\"variant\": \"%s\", \"pieces_per_iteration\": %d, \"cycles_per_iteration\": %.1f, \"extra_cycles_per_KiB_piece\": %.1f}\n", name, P, v, (v - (MF ? (RD ? base_rd : base) : 0.0)) / P); }
    const double base_rd = run<0, 16, true, true>(d_src, d_sink, d_cyc);
    printf("{\"variant\": \"16 MFMAs + 8 ds_read_b128, no DMA\", \"cycles_per_iteration\": %.1f}\n", base_rd);
    ROW(1, 16, true, false, "16 MFMAs + dwordx4 pieces");
    ROW(2, 16, true, false, "16 MFMAs + dwordx4 pieces");
    ROW(4, 16, true, false, "16 MFMAs + dwordx4 pieces");
    ROW(8, 16, true, false, "16 MFMAs + dwordx4 pieces");
    ROW(2, 4, true, false, "16 MFMAs + 4 x dword per piece");
    ROW(2, 16, true, true, "16 MFMAs + 8 ds_read_b128 + dwordx4 pieces");
    ROW(4, 16, true, true, "16 MFMAs + 8 ds_read_b128 + dwordx4 pieces");
    ROW(1, 0, true, false, "16 MFMAs + register-staged pieces (global_load_dwordx4 + ds_write_b128)");
    ROW(2, 0, true, false, "16 MFMAs + register-staged pieces (global_load_dwordx4 + ds_write_b128)");
    ROW(4, 0, true, false, "16 MFMAs + register-staged pieces (global_load_dwordx4 + ds_write_b128)");
    ROW(8, 0, false, false, "register-staged pieces only (no MFMA)");
    ROW(8, 16, false, false, "dwordx4 pieces only (no MFMA)");
    ROW(8, 4, false, false, "4 x dword per piece only (no MFMA)");
    return 0;
}
