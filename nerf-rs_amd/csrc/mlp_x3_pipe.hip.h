// mlp_x3_pipe.hip.h -- the weight-stream pipeline shared by the two bf16x3 kernels (mlp_kernel_bf16x3.hip, mlp_kernel_bf16x3b.hip):
// 24-KiB chunks of eight 3-KiB units in an LDS ring filled by LDS-DMA, one s_waitcnt + s_barrier per chunk at unit 4, operand
// fragments prefetched kX3Ahead units ahead.  Included inside each kernel's anonymous namespace after kCB / kRS / kX3Ahead.
#pragma once

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct PipeX {
    const LDS_AS char *rd_base;   // LDS address (incl. lane * 16) of the chunk the next prefetched unit lives in
    const LDS_AS char *ring_lane;
    uint32_t rd_slot_off;
    u32x4 a[12];                  // A fragments (w1, w2, w3) of four consecutive units (slot = unit & 3): current, kX3Ahead in flight
    uint32_t ring_addr, wr_slot_off, next_off, stream_bytes;
    const char *gbase, *cur_src, *cur_src_hi; // cur_src_hi = cur_src + 4 KiB: pieces 4, 5 (instruction offsets reach 4095)
    uint32_t cur_dst, cur_dst_hi, lane16;
};

// LDS-DMA piece whose instruction offset OFF advances the global AND the LDS address (both = base + OFF + lane * 16)
template <int OFF>
__device__ __forceinline__ void glds_piece_off(uint32_t lane16, const char *gsrc, uint32_t dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane16), "s"(gsrc), "s"(dst), "n"(OFF)
                 : "memory");
}

__device__ __forceinline__ void pipe_next_chunk(PipeX &P) {
    uint32_t off = P.next_off, slot = P.wr_slot_off;
    asm volatile("" : "+s"(off), "+s"(slot));
    P.cur_src = P.gbase + off;
    P.cur_dst = P.ring_addr + slot;
    P.cur_src_hi = P.cur_src + 4096;
    P.cur_dst_hi = P.cur_dst + 4096;
    off += kCB;
    P.next_off = (off == P.stream_bytes) ? 0u : off;
    slot += kCB;
    P.wr_slot_off = (slot == kRS * kCB) ? 0u : slot;
}

// (Re)start at chunk 0 in the state a steady-state run is in there: chunks 0 .. kRS - 3 landed, chunk kRS - 2 selected with
// the four pieces that units 4..7 of "chunk -1" would have issued (units 0, 1 of chunk 0 issue pieces 4, 5), unit 0
// prefetched.  The caller guarantees that no wave still reads the ring.
__device__ __forceinline__ void pipe_start(PipeX &P) {
    P.next_off = 0;
    P.wr_slot_off = 0;
#pragma unroll
    for (int c = 0; c < kRS - 2; ++c) {
        pipe_next_chunk(P);
#pragma unroll
        for (int i = 0; i < 6; ++i) glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
    }
    pipe_next_chunk(P);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds_piece(P.lane16, P.cur_src + i * 1024, P.cur_dst + i * 1024);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    P.rd_slot_off = 0;
    P.rd_base = P.ring_lane;
#pragma unroll
    for (int s = 0; s < 3 * kX3Ahead; ++s) P.a[s] = *(const LDS_AS u32x4 *)(P.rd_base + s * 1024);
}

// Unit U (0..7 within its chunk) begins: hand out its three A fragments.  Unit 4: the chunk after this one must have landed
// (every wave waits for its own pieces, then the barrier) and the slot of the previous chunk is refilled with chunk c + kRS - 1.
template <int U>
__device__ __forceinline__ void pipe_take(PipeX &P, bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) {
    if constexpr (U == 4) {
        // chunk c + 1 must have landed; the six pieces of each of the kRS - 3 chunks issued after it may still be in flight
        // (VMEM returns in order; a compiler-issued access in between only makes this wait longer)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(6 * (kRS - 3)) : "memory");
        pipe_next_chunk(P);
    }
    constexpr int cur = (U & 3) * 3; // the unit's first MFMA takes the fragment that was read LAST (w3): LDS returns in order, so
                                     // hipcc's one s_waitcnt for it covers all three
    a1 = __builtin_bit_cast(bf16x8, P.a[cur]); a2 = __builtin_bit_cast(bf16x8, P.a[cur + 1]); a3 = __builtin_bit_cast(bf16x8, P.a[cur + 2]);
}

// start fetching the fragments of unit U + kX3Ahead (issued behind the first MFMA of unit U)
template <int U>
__device__ __forceinline__ void pipe_prefetch(PipeX &P) {
    constexpr int nxt = ((U + kX3Ahead) & 3) * 3;
    if constexpr (U + kX3Ahead == 8) { // the unit to fetch opens the next chunk
        uint32_t off = P.rd_slot_off + kCB;
        off = (off == kRS * kCB) ? 0u : off;
        P.rd_slot_off = off;
        P.rd_base = P.ring_lane + off;
    }
    constexpr int nu = (U + kX3Ahead) & 7;
#pragma unroll
    for (int s = 0; s < 3; ++s) P.a[nxt + s] = *(const LDS_AS u32x4 *)(P.rd_base + (3 * nu + s) * 1024);
}

// the LDS-DMA piece issued behind unit U: pieces of the chunk selected at the last sync, in issue order
// U 4 -> 0, 5 -> 1, 6 -> 2, 7 -> 3, 0 -> 4, 1 -> 5
template <int U>
__device__ __forceinline__ void pipe_dma(PipeX &P) {
    if constexpr (U >= 4) glds_piece_off<(U - 4) * 1024>(P.lane16, P.cur_src, P.cur_dst);
    else if constexpr (U <= 1) glds_piece_off<U * 1024>(P.lane16, P.cur_src_hi, P.cur_dst_hi);
}

