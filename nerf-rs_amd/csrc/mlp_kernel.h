// mlp_kernel.h -- launch interface of the fused MLP kernel (internal to libnerf_mi355x.so).
#pragma once
#include <hip/hip_runtime.h>

enum { MLP_MODE_POINTS = 0, MLP_MODE_RAYS = 1, MLP_MODE_LIST = 2 }; // LIST (f32 and split kernels): ray mode over a device-side list of sample indices (bit 31 of an entry: audited certificate)

struct MlpArgs {
    const float *wstream;      // packed weight stream (mlp_layout.h), device
    const float *small_params; // biases + head weights, kSmallFloats floats, device
    int n_points;
    int mode;
    // MLP_MODE_POINTS: forward_batch layouts (src/network.rs:197): points 3 x n SoA, dirs n x 3 AoS
    const float *pts_soa;
    const float *dirs_aos;
    // MLP_MODE_RAYS: point i = sample (i % samples_per_ray) of ray (i / samples_per_ray)
    const float *ray_dirs; // n_rays x 3, unit
    const float *t;        // n_rays x samples_per_ray
    int samples_per_ray;
    float origin[3];
    // outputs
    float *sigma_out; // n
    float *rgb_out;   // n x 3 (full kernels only)
    int skip_empty;                   // full kernels: skip the colour head of tiles whose 128 sigmas are all 0 (exact)
    unsigned long long *skip_counter; // optional: number of skipped 128-point tiles (atomic)
    unsigned long long *clock_out; // optional diagnostic: per workgroup {shader cycles, 100 MHz ticks} of the tile loop
    unsigned int *nonfinite;       // optional (split arithmetics): += points whose density pre-activation is NaN / inf (f16 range overflow)
    // zero certification (nerf_render_opts.certify_zero; nerf_api.cpp cert_pass): the bf16 kernel stores the density PRE-activation (no ReLU) in sigma_out ...
    int raw_pre;
    // ... and the f32 kernel evaluates only the listed samples (MLP_MODE_LIST: slot i -> sample point_list[i] = ray * samples_per_ray + k;
    // outputs are scattered to the sample's own position; n_points = capacity of the list, *point_list_count = its length)
    const unsigned int *point_list;
    const unsigned int *point_list_count;
    // optional second part of the list, filled from the END of the same buffer downwards (entry k at n_points - 1 - k): certify_zero puts the
    // listed samples that are probably zeros there (and the audited certificates), so that their tiles are all-zero and skip_empty skips
    // their colour heads.  Slots [0, front) map to entries [0, front), slots [front, front + back) to the last `back` entries.
    const unsigned int *point_list_count_back;
};

// Sets the dynamic-LDS attribute of every kernel instantiation on the current device.
hipError_t nerf_mlp_init();
// full=false evaluates dense0..7 + alpha only (sigma); n_blocks = persistent workgroups (<= #CUs).
hipError_t nerf_mlp_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream);
// bf16-operand variant (mlp_kernel_bf16v2.hip): a.wstream is the output-tile-major stream (mlp_layout.h kChunks*Bf16V2)
hipError_t nerf_mlp_bf16v2_init();
hipError_t nerf_mlp_bf16v2_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream);
// mlp_kernel_f16v2.hip: the f16 twin of the bf16 kernel, sigma-only forms (certify_zero's pre-filter; MlpArgs / SeqArgs .nonfinite += tiles that left the f16 range)
hipError_t nerf_prefilter_f16v2_init();
hipError_t nerf_mlp_f16v2_launch(const MlpArgs &a, int n_blocks, hipStream_t stream);
#ifdef NERF_V2_F16_FULL // experiment (variant builds only): the full f16 kernel
hipError_t nerf_mlp_f16v2_full_launch(const MlpArgs &a, int n_blocks, hipStream_t stream);
#endif
// skip_dead in the bf16 arithmetic (same file): two ray cursors per wave; the trunk exports the bf16-packed relu(h8) of the live samples
// (512 B per sample; capacity nerf_seq_h8_bytes_bf16) for nerf_colour_bf16_launch.  SeqArgs / ColourArgs are declared below.
struct SeqArgs;
struct ColourArgs;
hipError_t nerf_seq_bf16_init();
size_t nerf_seq_h8_bytes_bf16(size_t n_samples);
hipError_t nerf_trunk_seq_bf16_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream);
hipError_t nerf_trunk_seq_f16v2_launch(const SeqArgs &a, int n_blocks, hipStream_t stream); // SeqArgs.prefilter only
hipError_t nerf_colour_bf16_launch(const ColourArgs &a, int n_blocks, hipStream_t stream);
// the same arithmetic on v_mfma_f32_16x16x32_bf16 (mlp_kernel_bf16v3.hip): its own piece contents, the same chunk counts
hipError_t nerf_mlp_bf16v3_init();
hipError_t nerf_mlp_bf16v3_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream);
// f32 by three-way bf16 split (mlp_kernel_bf16x3.hip): a.wstream is the three-part stream (mlp_layout.h kChunks*X3)
hipError_t nerf_mlp_bf16x3_init();
hipError_t nerf_mlp_bf16x3_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream);

// ---- exact dead-sample skipping (mlp_kernel_seq.hip for f32; mlp_split_kernels.hip.h for the split arithmetics) ------------
// Ray-sequential trunk: waves take rays from a device-side queue and walk each ray's samples front to back in chunks of 32,
// stopping at the reference's T < 1e-4 cut (src/lib.rs:276-279).  sigma_out must be zero-filled by the caller (samples behind
// the cut are never written).  With export_live the colour head runs on every sample with weight > 0:
//   f32 kernel            in the same launch (live samples compacted in LDS, colour passes of 32 columns): set rgb_out (zero-filled);
//   split arithmetics     trunk output to `h8` (compacted in HBM, 1 KiB per sample; capacity = all samples of the launch,
//                         nerf_seq_h8_bytes) for the second launch nerf_colour_*_launch;
//   bf16                  the same two-launch form with bf16-packed tiles (512 B per sample).
struct SeqArgs {
    const float *wstream;      // the f32 packed weight stream (sigma part is used)
    const float *small_params;
    int n_rays, samples_per_ray;
    const float *ray_dirs;     // n_rays x 3, unit
    const float *t;            // n_rays x samples_per_ray, ascending
    float origin[3];
    float far_;
    float *sigma_out;          // n_rays x samples_per_ray
    unsigned int *ray_counter; // zeroed before the launch
    unsigned int *live_count;  // zeroed before the launch (export_live): number of samples with weight > 0
    float *rgb_out;            // f32 kernel with export_live: n_rays x samples_per_ray x 3, zero-filled by the caller
    float *h8;
    unsigned int *slot_point;  // sample index (ray * samples_per_ray + k) of every exported slot
    unsigned long long *stats; // optional: += number of samples evaluated
    // hybrid sampling: evaluate only the rays of a device-side list (n_rays then only bounds the grid); retired rays zero-fill the
    // rest of their sigma row themselves (the row holds another arithmetic's values, the caller does not clear it)
    const unsigned int *ray_list;
    const unsigned int *ray_list_count;
    int zero_fill_after_cut;
    unsigned int *nonfinite;   // optional (split arithmetics): += points whose density pre-activation is NaN / inf
    // certify_zero's pre-filter (bf16 trunk only, export_live = false): store the density PRE-activation instead of sigma and retire a ray
    // once its bf16 transmittance falls below prefilter_cut_T (a prediction: kept below the reference's 1e-4 so that k_cert_plan's own
    // predicted cut always falls inside the evaluated samples).  The caller fills sigma_out with NaN (0xFF bytes): "not evaluated".
    int prefilter;
    float prefilter_cut_T;
};
struct ColourArgs {
    const float *wstream;      // the same stream (bottleneck + viewdirs part is used)
    const float *small_params;
    const unsigned int *live_count;
    const float *h8;
    const unsigned int *slot_point;
    const float *ray_dirs;
    int samples_per_ray;
    float *rgb_out;            // n x 3, scattered by sample index; zero-filled by the caller
    unsigned int *nonfinite;   // optional: += samples for which an operand left the arithmetic's range
};
hipError_t nerf_seq_init();
size_t nerf_seq_h8_bytes(size_t n_samples);
hipError_t nerf_trunk_seq_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream);
// the same two kernels in the bf16x3 arithmetic (mlp_kernel_bf16x3.hip): a.wstream is the three-part stream; the h8 tiles and
// the small parameters have the f32 kernels' layout
hipError_t nerf_seq_x3_init();
hipError_t nerf_trunk_seq_x3_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream);
hipError_t nerf_colour_x3_launch(const ColourArgs &a, int n_blocks, hipStream_t stream);
// f32 by two-way f16 split (mlp_kernel_f16x2.hip): a.wstream is the two-part f16 stream (mlp_layout.h kChunks*F16X2)
hipError_t nerf_mlp_f16x2_init();
hipError_t nerf_mlp_f16x2_launch(const MlpArgs &a, bool full, int n_blocks, hipStream_t stream);
hipError_t nerf_seq_f16x2_init();
hipError_t nerf_trunk_seq_f16x2_launch(const SeqArgs &a, bool export_live, int n_blocks, hipStream_t stream);
hipError_t nerf_colour_f16x2_launch(const ColourArgs &a, int n_blocks, hipStream_t stream);
