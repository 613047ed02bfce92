// lds_dma_stagger_probe.hip -- is the cost of an LDS-DMA piece beside MFMAs (lds_dma_probe.hip: 31-48 cycles per 1-KiB piece) the
// instruction's own, or the four waves of a workgroup meeting in the CU's one vector-memory path?  The bf16 kernel's waves run in
// lockstep (one s_barrier per 16-KiB chunk) and all four issue their pieces behind the same MFMAs (steps 9, 11, 13, 15 of a chunk).
// Here: one workgroup of 4 waves per CU, per iteration ("chunk") 16 steps x 2 MFMAs (v_mfma_f32_32x32x16_bf16, zero operands, 32 cycles
// each; ideal 1024 cycles), one s_barrier at step 8, 4 pieces per wave per chunk (16 KiB per workgroup, as the kernel), placed
// every half-slot of steps 8..15 carries the same instruction sequence in every wave; EXEC (all lanes / none) decides which are real pieces:
//   ALIGNED : every wave at steps 9, 11, 13, 15 behind the first MFMA (the kernel's schedule)
//   STAGGER : wave w, piece i in half-slot 4 i + w counted from step 8 (slot = step x 2 + which MFMA): no two waves behind the same MFMA
//   SOLO    : wave (chunk & 3) issues all 16 pieces of the chunk, one per step; the others none (the duty rotates)
//   NONE    : no pieces
// build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_stagger_probe lds_dma_stagger_probe.hip ; run: ./lds_dma_stagger_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void mm32(bf16x8 a, bf16x8 b, f32x16 &c) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

__device__ __forceinline__ void glds(uint32_t lane_off, const char *gsrc, uint32_t dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gsrc), "s"(dst) : "memory");
}

// the piece is issued with EXEC = all lanes if `on`, EXEC = 0 otherwise (a vector-memory instruction without lanes); M0 is not saved (hipcc
// uses none here)
__device__ __forceinline__ void glds_if(int on, uint32_t lane_off, const char *gsrc, uint32_t dst) {
    asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cselect_b64 exec, -1, 0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                 : : "s"(__builtin_amdgcn_readfirstlane(on)), "v"(lane_off), "s"(gsrc), "s"(dst) : "memory", "scc");
}

__device__ __forceinline__ void glds_nosave(uint32_t lane_off, const char *gsrc, uint32_t dst) { // M0 written, not restored
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds_nom0(uint32_t lane_off, const char *gsrc) { // M0 as it is (set once before the loop): timing only
    asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(gsrc) : "memory");
}
template <int OFF>
__device__ __forceinline__ void glds_nom0_off(uint32_t lane_off, const char *gsrc) { // M0 as it is, piece chosen by the instruction offset
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane_off), "s"(gsrc), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gload(uint32_t lane_off, const char *gsrc, uint4 &d) { // a plain load of the same bytes into VGPRs
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(lane_off), "s"(gsrc) : "memory");
}

enum { NONE = 0, ALIGNED = 1, STAGGER = 2, SOLO = 3, PLAIN_ALIGNED = 4, PLAIN_NOSAVE = 5, PLAIN_NOM0 = 6, PLAIN_NOM0_OFF = 7, PLAIN_VGPR = 8 };

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const char *src, float *sink, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bf16x8 a = {}, b = {};
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    const uint32_t lane_off = lane * 16;
    const uint32_t ring = (uint32_t)(uintptr_t)lds;
    uint32_t goff = blockIdx.x * 16384u, slot = 0;
    constexpr bool kBarrier = true;
    uint4 stage[4] = {};
    if (MODE == PLAIN_NOM0 || MODE == PLAIN_NOM0_OFF) asm volatile("s_mov_b32 m0, %0" : : "s"(ring + wave * 4096u));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int solo_wave = it & 3;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (st == 8) {
                if (MODE != NONE) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); // issue cost only: never waits for data
                if (kBarrier) asm volatile("s_barrier" ::: "memory");
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 0) mm32(a, b, c0); else mm32(a, b, c1);
                const int hs = (st - 8) * 2 + half; // half-slot from step 8
                if (MODE == PLAIN_ALIGNED) { // the kernel's schedule, nothing else in the instruction stream
                    if (half == 0 && st >= 9 && (st & 1)) glds(lane_off, src + ((goff + (wave * 4 + (st - 9) / 2) * 1024u) & 0x1fffffu), ring + ((slot + (wave * 4 + (st - 9) / 2) * 1024u) % 49152u));
                } else if (MODE >= PLAIN_NOSAVE) {
                    if (half == 0 && st >= 9 && (st & 1)) {
                        const char *g = src + ((goff + (wave * 4 + (st - 9) / 2) * 1024u) & 0x1fffffu);
                        if (MODE == PLAIN_NOSAVE) glds_nosave(lane_off, g, ring + ((slot + (wave * 4 + (st - 9) / 2) * 1024u) % 49152u));
                        if (MODE == PLAIN_NOM0) glds_nom0(lane_off, g);
                        if (MODE == PLAIN_NOM0_OFF) {
                            const char *gw = src + ((goff + wave * 4096u) & 0x1fffffu);
                            if (st == 9) glds_nom0_off<0>(lane_off, gw);
                            if (st == 11) glds_nom0_off<1024>(lane_off, gw);
                            if (st == 13) glds_nom0_off<2048>(lane_off, gw);
                            if (st == 15) glds_nom0_off<3072>(lane_off, gw);
                        }
                        if (MODE == PLAIN_VGPR) gload(lane_off, g, stage[(st - 9) / 2]);
                    }
                } else if (MODE != NONE && st >= 8) {
                    // the SAME instructions in all 16 half-slots of every wave (no branches: hipcc's code around wave-dependent branches
                    // moved MFMA operands about); EXEC decides whether the piece is real.  on = this wave's piece goes out here.
                    int on, piece;
                    if (MODE == ALIGNED) { on = (hs & 3) == 2; piece = wave * 4 + (hs >> 2); }           // steps 9, 11, 13, 15, first MFMA: all waves together
                    else if (MODE == STAGGER) { on = (hs & 3) == wave; piece = wave * 4 + (hs >> 2); }    // wave w behind MFMA 4 i + w
                    else { on = wave == solo_wave; piece = hs; }                                          // SOLO: one wave, every half-slot
                    glds_if(on, lane_off, src + ((goff + piece * 1024u) & 0x1fffffu), ring + ((slot + piece * 1024u) % 49152u));
                }
            }
        }
        goff += 16384u; slot = (slot + 16384u) % 49152u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int i = 0; i < 4; ++i) s += (float)(stage[i].x ^ stage[i].w);
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, const char *d_src, float *d_sink, unsigned long long *d_cyc, double base) {
    const int iters = 4000, blocks = 256;
    CK(hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 49152, 0, d_src, d_sink, d_cyc, iters);
        CK(hipDeviceSynchronize());
    }
    static unsigned long long c[1024];
    CK(hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
    double m = 0, mx = 0;
    for (int i = 0; i < blocks * 4; ++i) { m += (double)c[i]; if ((double)c[i] > mx) mx = (double)c[i]; }
    m = m / (blocks * 4) / iters; mx = mx / iters;
    printf("{\"variant\": \"%s\", \"cycles_per_chunk_mean\": %.1f, \"cycles_per_chunk_slowest_wave\": %.1f, \"ideal\": 1024, \"extra_cycles_per_piece_per_wave\": %.1f}\n",
           name, m, mx, base > 0 ? (m - base) / 4.0 : 0.0);
}

static double g_base = 0;

int main() {
    char *d_src; float *d_sink; unsigned long long *d_cyc;
    CK(hipMalloc(&d_src, (2u << 20) + 32768));
    CK(hipMemset(d_src, 0, (2u << 20) + 32768));
    CK(hipMalloc(&d_sink, 256 * 256 * sizeof(float)));
    CK(hipMalloc(&d_cyc, 1024 * sizeof(unsigned long long)));
    {   // base: no pieces
        const int iters = 4000;
        CK(hipFuncSetAttribute((const void *)probe<NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<NONE>), dim3(256), dim3(256), 49152, 0, d_src, d_sink, d_cyc, iters); CK(hipDeviceSynchronize()); }
        static unsigned long long c[1024];
        CK(hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
        double m = 0; for (int i = 0; i < 1024; ++i) m += (double)c[i];
        g_base = m / 1024 / iters;
    }
    run<NONE>("32 MFMAs + barrier, no pieces", d_src, d_sink, d_cyc, 0);
    run<ALIGNED>("aligned: all waves behind the same MFMAs (steps 9, 11, 13, 15)", d_src, d_sink, d_cyc, g_base);
    run<STAGGER>("staggered: no two waves behind the same MFMA", d_src, d_sink, d_cyc, g_base);
    run<SOLO>("solo: one wave issues the chunk's 16 pieces, duty rotates", d_src, d_sink, d_cyc, g_base);
    run<PLAIN_ALIGNED>("the kernel's schedule alone (4 pieces per wave, no lane-less instructions)", d_src, d_sink, d_cyc, g_base);
    run<PLAIN_NOSAVE>("the kernel's schedule, M0 written but not saved / restored", d_src, d_sink, d_cyc, g_base);
    run<PLAIN_NOM0>("the kernel's schedule, M0 not touched in the loop (timing only)", d_src, d_sink, d_cyc, g_base);
    run<PLAIN_NOM0_OFF>("the kernel's schedule, M0 not touched, pieces by instruction offset (timing only)", d_src, d_sink, d_cyc, g_base);
    run<PLAIN_VGPR>("the kernel's schedule with plain global_load_dwordx4 into VGPRs (no LDS)", d_src, d_sink, d_cyc, g_base);
    return 0;
}
