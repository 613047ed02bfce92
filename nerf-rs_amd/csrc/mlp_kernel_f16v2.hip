// mlp_kernel_f16v2.hip -- the f16 twin of mlp_kernel_bf16v2.hip (same source, NERF_V2_F16 = 1): v_mfma_f32_32x32x16_f16 on f16-rounded weights and
// layer inputs, f32 accumulate, f32 heads.  Only the sigma-only forms are built; they are certify_zero's pre-filter: with 11 significand
// bits instead of 8 the pre-activation it predicts is 8 x closer to the exact one, so the margin below which a sample is certified a zero of the exact
// network can be 6-8 x tighter and the exact kernel evaluates fewer samples.  f16 overflows at 65 504: an overflow ends in a non-finite density
// pre-activation, which is never certified and is counted; the host then falls back to the bf16 pre-filter for that network (nerf_api.cpp).
#define NERF_V2_F16 1
#include "mlp_kernel_bf16v2.hip"
