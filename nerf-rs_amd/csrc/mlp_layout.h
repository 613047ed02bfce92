// mlp_layout.h -- shared (host packer + device kernel) description of the packed weight stream.
//
// The fused MLP kernel (mlp_kernel.hip) keeps a wave's activations in registers in the C/D layout
// of v_mfma_f32_32x32x2_f32: a wave owns 32 points; lane l = (p = l & 31, h = l >> 5); an activation
// "tile" is 16 registers, register r of tile t on lane (p,h) holds feature
//       F(t, r, h) = 32 t + (r & 3) + 8 (r >> 2) + 4 h            of point p.
// One MFMA k-step (K = 2) takes register r of input tile t as its B operand (B[k = h][j = p]), so
// the two k-rows of step s = 16 t + r are features F(t,r,0) and F(t,r,1).  The A operand of the MFMA
// for output tile nt is therefore  A[i = l & 31][k = l >> 5] = W[row(s, h)][32 nt + i].
//
// Weight stream: for each layer, for each k-step s, for each group g of 4 output tiles, one 1-KiB
// "piece":   piece[(l * 4 + q)] = W[row(s, l >> 5)][32 (4 g + q) + (l & 31)]      (l = lane 0..63)
// so a wave fetches its A operands for 4 MFMAs with one conflict-free ds_read_b128.  The stream is
// cut in 16-KiB chunks (= 64/NT k-steps) which the kernel DMA's into a 3-slot LDS ring in order.
#pragma once
#include <stdint.h>

namespace nerfmlp {

constexpr int kChunkBytes = 16384;
constexpr int kChunkFloats = kChunkBytes / 4;
#ifndef NERF_RING_SLOTS
#define NERF_RING_SLOTS 3
#endif
constexpr int kRingSlots = NERF_RING_SLOTS; // >= 3; chunk c + kRingSlots - 1 is DMA'd while chunk c is consumed
constexpr int kPointsPerWave = 32;
constexpr int kWavesPerBlock = 4;
constexpr int kPointsPerBlock = kPointsPerWave * kWavesPerBlock;

// k-steps per layer (K padded to the 32-slot tiles) and output tiles
constexpr int kStepsL0 = 32;    // 64 encoding slots (63 used)
constexpr int kStepsHid = 128;  // 256
constexpr int kStepsL5 = 160;   // 64 encoding slots + 256
constexpr int kStepsView = 144; // 256 + 32 dir slots (27 used)

constexpr int chunksOf(int steps, int nt) { return steps * nt * 256 / kChunkBytes; }

// layer order in the stream: dense0..dense7, [sigma-only kernels stop here], bottleneck, viewdirs
constexpr int kChunksSigma = chunksOf(kStepsL0, 8) + 4 * chunksOf(kStepsHid, 8) + chunksOf(kStepsL5, 8) +
                             2 * chunksOf(kStepsHid, 8);                                  // 120
constexpr int kChunksFull = kChunksSigma + chunksOf(kStepsHid, 8) + chunksOf(kStepsView, 4); // 145

// small parameters kept resident in LDS (floats)
constexpr int kBiasOff = 0;                     // 9 layers x [8 nt][2 h][16]  (dense0..7, bottleneck)
constexpr int kBiasViewOff = kBiasOff + 9 * 256; // [4 nt][2 h][16]
constexpr int kAlphaWOff = kBiasViewOff + 128;  // [2 h][128]
constexpr int kRgbWOff = kAlphaWOff + 256;      // [2 h][3 c][64]
constexpr int kMiscOff = kRgbWOff + 384;        // alpha bias, rgb bias r,g,b
constexpr int kSmallFloats = ((kMiscOff + 4 + 63) / 64) * 64; // 3136
constexpr int kSmallBytes = kSmallFloats * 4;

constexpr int kLdsBytes = kRingSlots * kChunkBytes + kSmallBytes;

// ---- bf16 stream (mlp_kernel_bf16.hip): macro-step = 8 KiB (8 pieces of 64 lanes x 8 bf16), 32-KiB chunks.
// Per input tile (32 k-rows) an 8-tile layer has 2 k-steps x 8 pieces = 2 macro-steps; the 4-tile viewdirs layer
// 2 k-steps x 4 pieces = 1 macro-step.  viewdirs (9 tiles = 72 KiB) is zero-padded to 3 chunks.
constexpr int kChunkBytesBf16 = 32768;
constexpr int kRingSlotsBf16 = 3;
constexpr int kChunksSigmaBf16 = 1 + 4 * 4 + 5 + 2 * 4;        // dense0 (2 tiles), dense1-4, dense5 (10 tiles), dense6-7 = 30
constexpr int kChunksFullBf16 = kChunksSigmaBf16 + 4 + 3;       // + bottleneck + viewdirs (padded) = 37
constexpr int kLdsBytesBf16 = kRingSlotsBf16 * kChunkBytesBf16 + kSmallBytes;

// ---- bf16 stream v2 (mlp_kernel_bf16v2.hip): output-tile-major.  Per layer, per output tile nt, per k-step (K = 16) one
// 1-KiB piece; 16-KiB chunks of 16 pieces.  Pieces per layer: dense0 8 x 4, hidden 8 x 16, dense5 8 x 20, viewdirs 4 x 18
// (+ 8 zero pieces so that the stream ends on a chunk boundary).  64 points per wave, 256 per workgroup.
constexpr int kChunkBytesBf16V2 = 16384;
#ifndef NERF_BV2_RING_SLOTS
#define NERF_BV2_RING_SLOTS 3
#endif
constexpr int kRingSlotsBf16V2 = NERF_BV2_RING_SLOTS; // 3..8: chunk c + kRingSlots - 1 is DMA'd while chunk c is consumed (a chunk lasts ~0.45 us)
constexpr int kChunksSigmaBf16V2 = (32 + 4 * 128 + 160 + 2 * 128) / 16; // 60
constexpr int kChunksFullBf16V2 = kChunksSigmaBf16V2 + (128 + 72 + 8) / 16; // 73
constexpr int kLdsBytesBf16V2 = kRingSlotsBf16V2 * kChunkBytesBf16V2 + kSmallBytes;
constexpr int kPointsPerBlockBf16V2 = 256;

// ---- bf16x3 stream (mlp_kernel_bf16x3.hip, f32 by three-way bf16 split): the first bf16 design's pieces in the same order
// (layer, input tile, k-step, output tile), each followed by the pieces of the second and third bf16 part of the same
// weights: a unit = 3 KiB, a k-step of an 8-tile layer = 24 KiB = one chunk, a k-step of viewdirs half a chunk.  No padding.
constexpr int kChunkBytesX3 = 24576;
#ifndef NERF_X3_RING_SLOTS
#define NERF_X3_RING_SLOTS 3
#endif
constexpr int kRingSlotsX3 = NERF_X3_RING_SLOTS; // 3..5 (measured equal: the kernel is power-limited, not latency-limited)
constexpr int kChunksSigmaX3 = 2 * 2 + 4 * 16 + 20 + 2 * 16;  // dense0 (2 tiles x 2 k-steps), dense1-4, dense5, dense6-7 = 120
constexpr int kChunksFullX3 = kChunksSigmaX3 + 16 + 9;        // + bottleneck + viewdirs (9 tiles, one chunk each) = 145
constexpr int kLdsBytesX3 = kRingSlotsX3 * kChunkBytesX3 + kSmallBytes;
constexpr int kPiecesV1 = 8 * (4 + 4 * 16 + 20 + 3 * 16) + 4 * 18;  // 1160 pieces of the first bf16 design, without its padding

// ---- f16x2 stream (mlp_kernel_f16x2.hip, f32 by two-way f16 split): the x3 stream's units with two pieces (the f16 parts w1, w2 of
// the same 32 x 16 weight block) instead of three: a unit = 2 KiB, a k-step of an 8-tile layer = 16 KiB = one chunk, a k-step of
// viewdirs half a chunk; the same chunk counts as the x3 stream.
constexpr int kChunkBytesF16X2 = 16384;
#ifndef NERF_F16X2_RING_SLOTS
#define NERF_F16X2_RING_SLOTS 4
#endif
constexpr int kRingSlotsF16X2 = NERF_F16X2_RING_SLOTS; // 3..6; a chunk lasts only 8 units x 3 MFMAs x 32 cycles = 0.35 us
constexpr int kChunksSigmaF16X2 = kChunksSigmaX3;
constexpr int kChunksFullF16X2 = kChunksFullX3;
constexpr int kLdsBytesF16X2 = kRingSlotsF16X2 * kChunkBytesF16X2 + kSmallBytes;

// feature held by register r (0..15) of a tile on lane-half h, relative to the tile's first feature
constexpr int regFeature(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Encoding slots.  Points: 2 tiles; lane-half h computes octaves 5h..5h+4 (idx 0..29 = 6*o + {sin xyz, cos xyz}),
// idx 30/31 = raw x,y (h=0) or raw z, zero pad (h=1).  Returns the reference feature row
// (src/network.rs:263-292: [x,y,z] then per octave sin xyz, cos xyz) or -1 for the pad.
constexpr int posSlotFeature(int idx, int h) {
    return idx < 30 ? 3 + 6 * (5 * h + idx / 6) + idx % 6 : (h == 0 ? idx - 30 : (idx == 30 ? 2 : -1));
}
// Directions: 1 tile; half h computes octaves 2h..2h+1 (idx 0..11); h=0 idx 12..14 = raw x,y,z; rest pad.
constexpr int dirSlotFeature(int idx, int h) {
    return idx < 12 ? 3 + 6 * (2 * h + idx / 6) + idx % 6 : ((h == 0 && idx < 15) ? idx - 12 : -1);
}

} // namespace nerfmlp
