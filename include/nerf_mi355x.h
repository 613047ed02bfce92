/*
 * nerf_mi355x.h -- C ABI of libnerf_mi355x.so: the MI355X (gfx950) drop-in for the hot path of
 * elisabeth96/nerf-rs (ray-march + MLP forward + volume integration).
 *
 * The reference has no FFI on this path (it is a single Rust crate); the seams this ABI replaces are
 *   S2  Network::forward_batch(&self, points: &Matrix [3 x B, SoA], view_dirs: &[Vec3]) -> (Vec<Vec3>, Vec<f32>)
 *                                                                   reference src/network.rs:197-237
 *   S3  render_image(&Network, &Network, &Camera, fine_samples_per_ray) -> Vec<Vec3>
 *                                                                   reference src/lib.rs:474-565
 * plus the host-side pieces a caller needs around them (loader src/lib.rs:108-174, camera_from_samples
 * src/lib.rs:614-645, save_ppm src/lib.rs:567-580).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer of the reference would add to call these from src/lib.rs.
 *
 * Conventions: every function returns 0 on success or a negative nerf_status; nerf_last_error() gives the
 * message (the reference panics instead: src/lib.rs:36,118,127,483-501).  The caller owns all in/out buffers;
 * the context owns device memory.  Plain pointers and sizes only -- no C++ / torch types.  A context is bound to
 * one HIP device and is single-caller: one call at a time, and consecutive asynchronous calls (`*_device`) must use the
 * same stream or be separated by a stream synchronisation -- they share the context's pass workspace.  Create one
 * context per GPU / per thread.
 * All buffers at the boundary are f32 (the MLP arithmetic inside is selected by nerf_render_opts.mlp_dtype).  Without the HIP runtime or a gfx950 device nerf_create fails (there is no CPU fallback).
 */
#ifndef NERF_MI355X_H
#define NERF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nerf_ctx nerf_ctx;

typedef enum {
    NERF_OK = 0,
    NERF_ERR_INVALID = -1,   /* bad argument (reference: assert!/panic) */
    NERF_ERR_IO = -2,        /* "read shapes"/"read tensor" (src/lib.rs:36,65) */
    NERF_ERR_MISSING = -3,   /* "missing matrix parameter"/"missing bias parameter" (src/lib.rs:118,127) */
    NERF_ERR_SHAPE = -4,     /* dims mismatch (debug_assert in src/lib.rs:119-120,128-129) */
    NERF_ERR_HIP = -5,       /* HIP runtime / device error */
    NERF_ERR_STATE = -6,     /* network not loaded */
    NERF_ERR_PARSE = -7      /* camera JSON malformed (src/lib.rs:620-631 unwrap/expect) */
} nerf_status;

enum { NERF_NET_COARSE = 0, NERF_NET_FINE = 1 };
enum { NERF_MLP_F32 = 0, NERF_MLP_BF16 = 1, NERF_MLP_BF16X3 = 2, NERF_MLP_F16X2 = 3 };

/* Mirrors `struct Camera` (src/lib.rs:197-211); samples_per_ray lives in nerf_render_opts.n_coarse.
 * alpha_* are the half field-of-view angles (radians); dir/up need not be orthogonal (basis is rebuilt
 * per src/lib.rs:216-218). */
typedef struct {
    int32_t nx, ny;
    float alpha_width, alpha_height;
    float pos[3], dir[3], up[3];
    float near_, far_;
} nerf_camera;

/* render_image's remaining inputs (src/lib.rs:474-482, 603-612) + the extensions BASELINE.json's configs need.
 * Zero-initialise, then set n_coarse (> 0).  The reference behaviour is n_coarse=64, n_fine=128, everything else 0. */
typedef struct {
    int32_t n_coarse;     /* camera.samples_per_ray */
    int32_t n_fine;       /* fine_samples_per_ray; 0 => fine net evaluated on the coarse samples only (src/lib.rs:295) */
    int32_t coarse_only;  /* ext: composite the coarse net's own rgb/sigma, skip the fine pass */
    int32_t crop_x0, crop_y0, crop_w, crop_h; /* ext: output window in pixels; crop_w = crop_h = 0 => full frame */
    int32_t ssaa;         /* ext: s x s rays per pixel, box filter; 0 or 1 => off */
    uint64_t seed;        /* counter-RNG seed (reference: unseeded thread_rng, src/lib.rs:375,407) */
    int32_t mlp_dtype;    /* ext: NERF_MLP_F32 (0, default: exact-f32 MFMA, the parity path) or NERF_MLP_BF16 (1: bf16
                           * operands / f32 accumulate on the bf16 matrix cores -- BASELINE config C5; PSNR-level parity).
                           * NERF_MLP_BF16X3 (2, opt-in): f32-accurate arithmetic on the bf16 matrix cores -- every weight and
                           * activation is split into three bf16 parts (exact to 2^-27) and a product is the sum of the six
                           * significant bf16 x bf16 products, accumulated in f32.  In a hierarchical render the coarse
                           * (sampling) pass stays on the exact-f32 kernel so that the fine sample positions equal the f32
                           * path's bit for bit; the fine (colour) pass runs in bf16x3.  Meets the f32 path's tolerances.
                           * NERF_MLP_F16X2 (3, opt-in): the cheaper sibling -- two f16 parts per operand (exact to 2^-22), three
                           * products per f32 product; same f32 sampling pass, same tolerances (the error against float64 stays
                           * at the f32 kernel's level: accumulation rounding dominates), half the matrix work of bf16x3.
                           * Range: f16 overflows at 65 504 -- activations must stay below it (true for the lego networks inside
                           * the scene and well beyond: validated for |p| <= 16; bf16x3 has no such limit); a network with a
                           * weight beyond the f16 range makes this mode unavailable (NERF_ERR_STATE). */
    int32_t skip_empty;   /* ext: 1 = skip the colour head (bottleneck + viewdirs + rgb, 17 % of a full MLP
                           * evaluation) for every workgroup tile (128 samples in f32, 256 in bf16) whose densities are all 0.
                           * EXACT: such samples have alpha = 0 and weight 0, the image is bit-identical; only the work
                           * changes.  Default 0 so that timings are plain executed-FLOP figures. */
    int32_t skip_dead;    /* ext (the rest of it; every mlp_dtype): 1 = evaluate only what can reach a pixel.
                           * Rays are walked front to back in chunks of 32 samples; a ray is retired at the reference's
                           * T < 1e-4 cut (src/lib.rs:276-279: every later weight is exactly 0), and the colour head runs only
                           * on the samples whose weight is > 0 (F32: compacted in LDS, same launch; BF16 and the split arithmetics:
                           * compacted through an HBM buffer bounded by NERF_MAX_EXPORT_BYTES, second launch).  EXACT: the image is
                           * bit-identical to skip_dead = 0.  Takes precedence over skip_empty.  Default 0 so that timings are
                           * plain executed-FLOP figures; nerf_stats.n_exec_* report the evaluations actually executed. */
    int32_t hybrid_sampling; /* ext (needs skip_dead = 1, hierarchical render): 1 = run the SAMPLING (coarse) pass in a split
                           * arithmetic (the render's own; for an F32 render f16x2, or bf16x3 if the network exceeds the f16 range;
                           * the fine pass keeps mlp_dtype), then redo in exact f32 only the rays with an ill-conditioned
                           * hierarchical draw: one whose position is predicted to move by more than 1e-5 in t under the split
                           * arithmetic's density error (|dt| = bin width x |d cdf| / bin mass, |d cdf| bounded per bin edge to
                           * first order: light CDF bins, nearly empty rays),
                           * or whose transmittance passes within 0.1 % of the 1e-4 cut.  Those rays (18 % of the lego frame) get
                           * the f32 path's sample positions bit for bit; the others move by <= 1e-5 -- a bound under a measured
                           * model of the arithmetic's density error, not a proof: fuzzes of 700 M rays (tools/fuzz_hybrid_flags.py)
                           * found 0.9 unflagged rays per million beyond it, the largest at 2.8e-5.  Not bit-identical to
                           * hybrid_sampling = 0 (pixels differ by 2e-8 on average); held to the same Gate 1.
                           * nerf_stats.n_hybrid_rays = rays redone in f32. */
    int32_t certify_zero;  /* ext (ABI 4, reworked in ABI 5; mlp_dtype F32, BF16X3 or F16X2, no skip mode): 1 = a 16-bit pass over all samples (f16 operands where
                           * every weight and activation of the network fits the f16 range, else bf16; f32 accumulation either way) finds
                           * (Z) the samples whose density pre-activation is so far below 0 (per-network margin, see below) that the exact
                           * network's density is certainly 0 there as well, and predicts (C) where each ray's transmittance falls below the
                           * reference's 1e-4 cut (src/lib.rs:276-279: every later weight is zero-filled whatever its density).  The exact kernel
                           * -- for the fine pass of a split arithmetic that arithmetic's kernel -- evaluates only the other samples in front of
                           * the predicted cut (a device-side list: 7 % of the coarse, 15 % of the fine samples of the lego frame); the
                           * EXACT transmittance then confirms each cut, and where it does not (rare) the rest of that ray is evaluated in a
                           * second launch -- so (C) is exact by construction.  A certified sample has sigma = 0, weight 0 (src/lib.rs:271-272):
                           * the image is BIT-IDENTICAL to certify_zero = 0 as long as no certificate (Z) is wrong.
                           * (Z) rests on measurements, not on a proof, so it is AUDITED in every frame: a deterministic share of the certified samples (1 in 16 of those certified by
                           * less than twice the margin, 1 in 128 of the others) is evaluated
                           * exactly all the same; a positive density there (nerf_stats.n_certify_violations), or an audited sample on which the 16-bit
                           * pass was off by more than half the margin (nerf_stats.certify_max_error, certify_headroom), widens that network's margin for the life of
                           * the context (floors: 0.25 coarse, 0.5 fine with the f16 pass, 1.0 / 3.0 with the bf16 pass -- 2.5-9 x the largest difference to the
                           * exact pre-activation the audits see on certified samples of the lego networks; reset when a network is loaded) and the frame is rendered again (nerf_stats.n_certify_retries); if
                           * 8 widenings do not satisfy the audit the render fails with NERF_ERR_STATE.  An activation beyond 65 504 makes the f16 pass's
                           * pre-activations non-finite (never certified, counted): that network goes back to the bf16 pass for good and the frame is
                           * rendered again.  A network on which the 16-bit pass is less accurate
                           * thus calibrates itself, certifies nothing (random weights: pre-activations near 0), or fails loudly; what remains
                           * unobserved is a wrong certificate that is so rare that a 1-in-64 audit of ~1e8 certified samples per frame never
                           * meets one or its precursors.  Because of the audit a certify_zero render synchronises the stream before it returns
                           * (also nerf_render_image_device with stats == NULL).  nerf_stats.n_exec_* = samples the exact kernel evaluated. */
    /* ext (ABI 5): render only ONE BAND of the output window's rows (the whole window when band_count <= 1) -- what one GPU of several
     * renders (nerf_render_image_multi and nerf-rs_amd/distributed.py set these; reference counterpart: the rayon fan-out over blocks,
     * src/lib.rs:533-550, which balances by work stealing).  band_stripe_rows = 0: band band_index of band_count CONTIGUOUS bands (the
     * first rows % count bands one row longer) -- balanced when every ray costs the same.  band_stripe_rows = S > 0: the window's rows
     * are dealt out in stripes of S rows round-robin, this call renders the stripes band_index, band_index + band_count, ... -- balanced
     * also when rays differ in cost (skip_dead, certify_zero: the lego background, 75 % of the rays, is nearly free and sits in the top
     * rows).  rgb_out holds the band's rows PACKED, in frame order: nerf_band_rows(window rows, ...) x width x 3.  Every pixel is the same
     * bits as in a whole-window render (per-pixel counter RNG). */
    int32_t band_index, band_count, band_stripe_rows;
} nerf_render_opts;

/* Device-side timing of the last render (HIP events on the render stream). */
typedef struct {
    uint64_t n_rays, n_coarse_points, n_fine_points;
    double ms_total;       /* first kernel -> last kernel */
    double ms_coarse_mlp;  /* sum over passes */
    double ms_fine_mlp;
    double ms_other;       /* ray gen + sampling + compositing + downsample */
    uint32_t n_mlp_launches;
    uint32_t n_passes;
    uint64_t n_colour_skipped_points; /* samples whose colour head was skipped (skip_empty) */
    /* MLP evaluations actually executed (equal to the point counts above unless skip_empty / skip_dead removed work): */
    uint64_t n_exec_coarse_trunk;  /* coarse network, dense0..7 + alpha */
    uint64_t n_exec_fine_trunk;    /* fine network, dense0..7 + alpha */
    uint64_t n_exec_colour;        /* bottleneck + viewdirs + rgb (fine network, or the coarse one when coarse_only) */
    uint64_t n_hybrid_rays;        /* hybrid_sampling: rays whose coarse pass was redone in f32 (counted in n_exec_coarse_trunk too) */
    uint64_t n_nonfinite_points;   /* split arithmetics: evaluations in which an operand left the arithmetic's range (NERF_MLP_F16X2: an
                                    * activation beyond 65 504 -- every value is watched as it is split, one v_max3 per pair) or whose density
                                    * pre-activation was not finite.  0 in every validated configuration; non-zero means the frame is WRONG
                                    * there (f16 overflow yields finite garbage, not NaN): nerf_forward_batch_ex and every render that reads its
                                    * counters (nerf_render_image, nerf_render_image_multi, nerf_render_image_device with stats != NULL) fail
                                    * with NERF_ERR_STATE -- the stats are filled all the same. */
    /* certify_zero (ABI 5): the audit of the frame that was returned, see nerf_render_opts.certify_zero */
    uint64_t n_certify_audited;    /* certified samples that the exact kernel evaluated all the same (1 in 16 of those within twice the margin, 1 in 128 of the others) */
    uint64_t n_certify_violations; /* audited samples whose exact density was positive, summed over ALL renders of this frame (the last one had none) */
    uint32_t n_certify_retries;    /* times the frame was rendered again (margins widened after a failed audit, or the sample list enlarged) */
    uint32_t n_certify_fallback_rays; /* rays whose predicted cut the exact transmittance did not confirm (their remaining samples went through a second launch) */
    float certify_margin[2];       /* margins in force (coarse, fine network): a sample is certified iff its 16-bit (f16 / bf16 pass) pre-activation < -margin */
    float certify_headroom[2];     /* min over the audited samples of -(exact pre-activation): how far the closest one stood from a positive density (inf: none audited) */
    float certify_max_error[2];    /* max over the audited samples of |16-bit - exact pre-activation|: what the pre-filter got wrong on a sample it certified */
} nerf_stats;

/* ---- lifecycle ------------------------------------------------------------------------------------------ */
int nerf_create(int device_id, nerf_ctx **out);
void nerf_destroy(nerf_ctx *ctx);
/* Message of the last failing call on ctx (or of the last failing context-free call when ctx == NULL). */
const char *nerf_last_error(const nerf_ctx *ctx);
int nerf_device_info(const nerf_ctx *ctx, int *n_cus, char *arch_name, size_t arch_name_len);

/* ---- loader: load_network_from_dir (src/lib.rs:108-174), same directory format (shapes.txt + <name>.bin,
 * little-endian f32, kernels [in][out] row-major), same required tensor names, same failure cases ---------- */
int nerf_load_network_dir(nerf_ctx *ctx, int which, const char *dir);
/* For hosts that read the tensors themselves (Rust load_tensor): n named tensors, dims[2*i], dims[2*i+1]
 * (second = 0 for biases). */
int nerf_load_network_tensors(nerf_ctx *ctx, int which, int n, const char *const *names, const int64_t *dims,
                              const float *const *data);

/* Packed weight blob: the pre-padded, pre-permuted device image of one network in a single file, so
 * that start-up is one read + one memcpy.  nerf_pack_network_dir converts the reference's directory format (host-only);
 * nerf_load_network_blob uploads it.  The directory loader stays the compatibility path.  Blob layout: 16-byte header
 * {"NRFMI355", u32 version = 1, u32 n_floats} + the weight stream + the small-parameter block (mlp_layout.h). */
int nerf_pack_network_dir(const char *dir, const char *blob_path);
int nerf_load_network_blob(nerf_ctx *ctx, int which, const char *blob_path);

/* Host-only validation of a weight directory (same checks as nerf_load_network_dir, no device needed). */
int nerf_check_network_dir(const char *dir);
/* Host-only validation of a packed blob (same reader and checks as nerf_load_network_blob: magic, version, exact size). */
int nerf_check_network_blob(const char *blob_path);
/* Diagnostic: the packed device images of a weight directory (layout: nerf-rs_amd/csrc/mlp_layout.h).  Pass NULL
 * buffers to query the lengths (in floats).  Host-only. */
int nerf_debug_pack_network_dir(const char *dir, float *wstream, size_t wstream_cap, float *small, size_t small_cap,
                                size_t *wstream_len, size_t *small_len);
/* Diagnostic: the three bf16 parts (raw bit patterns) the NERF_MLP_BF16X3 packer stores for each of n f32 weights:
 * parts[3 i + k], k = 0..2, with v = p0 + p1 + p2 up to 2^-27 |v|.  Host-only. */
int nerf_debug_split_bf16x3(const float *values, size_t n, uint16_t *parts /* 3 n */);
/* Diagnostic: certify_zero's audit policy as the renderer applies it after every certified frame, per network -- given the margin in force and
 * what the audit found (audited certificates, violations among them, least headroom, largest |bf16 - exact| error: the nerf_stats fields),
 * returns 0 if the frame stands, else 1 (a violation: margin x 4, or 4 x the error), 2 (headroom below half the margin: x 2, or 4 x the
 * error) or 3 (error above half the margin: x 1.25, or 3 x the error) with the widened margin in *new_margin.  Host-only. */
int nerf_debug_certify_policy(float margin, uint64_t audited, uint64_t violations, float headroom, float max_error, float *new_margin);
/* The same for the NERF_MLP_F16X2 packer: two f16 parts (IEEE binary16 bit patterns), v = p0 + p1 up to 2^-22 |v|.  Host-only. */
int nerf_debug_split_f16x2(const float *values, size_t n, uint16_t *parts /* 2 n */);

/* ---- S2: Network::forward_batch (src/network.rs:197-237) -------------------------------------------------- */
/* host pointers, synchronous.  n == 0 is a no-op (src/network.rs:199-201).
 * Input domain: the positional encoding evaluates sin/cos(2^k p), k <= 9, with a branch-free three-constant Cody-Waite
 * reduction that is accurate (<= 1.2e-7 absolute) for |2^9 p| <= 2^20, i.e. |p| <= 2048 per coordinate -- three orders of
 * magnitude beyond the scene (|p| <= 2.42 in the lego frustum); beyond that the accuracy degrades gradually, nothing faults.
 * n is limited to INT32_MAX minus one grid stride of tiles (~2.1e9 points); larger batches return NERF_ERR_INVALID. */
int nerf_forward_batch(nerf_ctx *ctx, int which, const float *pts_soa /*3 x n*/, const float *dirs_aos /*n x 3*/,
                       size_t n, float *rgb_aos /*n x 3*/, float *sigma /*n*/);
/* same with an explicit MLP arithmetic (NERF_MLP_F32 / NERF_MLP_BF16 / NERF_MLP_BF16X3 / NERF_MLP_F16X2).  In the two split
 * arithmetics a density of 0 whose pre-activation lies within 4e-5 of 0 is returned as -0.0f (an "uncertain zero": the f32
 * kernel may see a tiny positive density there; numerically it IS 0 -- hybrid_sampling's flag reads the sign). */
int nerf_forward_batch_ex(nerf_ctx *ctx, int which, int mlp_dtype, const float *pts_soa, const float *dirs_aos, size_t n,
                          float *rgb_aos, float *sigma);
/* device pointers, asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream) */
int nerf_forward_batch_device(nerf_ctx *ctx, int which, const float *d_pts_soa, const float *d_dirs_aos, size_t n,
                              float *d_rgb_aos, float *d_sigma, void *stream);

/* ---- S3: render_image (src/lib.rs:474-565) ----------------------------------------------------------------- */
/* rgb_out: crop_h x crop_w x 3 (or ny x nx x 3) linear RGB f32, row-major, index (i*w + j)*3 as image[i*nx+j]
 * (src/lib.rs:552-557).  Unlike the reference (src/lib.rs:491-501) nx, ny need not be multiples of 8. */
int nerf_render_image(nerf_ctx *ctx, const nerf_camera *cam, const nerf_render_opts *opts, float *rgb_out,
                      nerf_stats *stats /* may be NULL */);
/* device output, asynchronous on `stream`; stats != NULL synchronises the stream before returning. */
int nerf_render_image_device(nerf_ctx *ctx, const nerf_camera *cam, const nerf_render_opts *opts, float *d_rgb_out,
                             void *stream, nerf_stats *stats);
/* ---- S3 over several GPUs of one node (reference: the rayon fan-out over blocks + scatter, src/lib.rs:533-557) ------
 * ctxs[i] is one context per device (nerf_create / nerf_create_multi), each with both networks loaded (weights are
 * replicated).  Context i renders band i of n of the output rows (nerf_render_opts.band_*, set here: the caller's values are
 * ignored) on its own host thread and stream: CONTIGUOUS bands -- first row i*(h/n) + min(i, h%n), h/n + (i < h%n) rows -- when every
 * ray costs the same, single rows dealt out round-robin (band_stripe_rows = 1) when opts->skip_dead, skip_empty or certify_zero make
 * the cost follow the scene; a band is bit-identical to the same rows of a single-context frame (per-pixel counter RNG).  `gather` selects how the bands meet in rgb_out (host, same layout as nerf_render_image):
 *   NERF_GATHER_HOST  each band is copied device -> host into its rows directly (no GPU-to-GPU traffic);
 *   NERF_GATHER_PEER  bands are copied GPU -> GPU over xGMI (hipMemcpyPeerAsync) into a frame on ctxs[0]'s device, then one D2H;
 *   NERF_GATHER_RCCL  ONE ncclAllGather of the bands (RCCL over xGMI; librccl is dlopen'ed on first use): the whole frame
 *                     ends up on every device, then one D2H from ctxs[0].  RCCL refuses two ranks on one device: when contexts
 *                     share a device (single-GPU test boxes) the collective step is rehearsed as device-to-device copies into the
 *                     same equal-slot buffers (same layout, stream ordering and ragged compaction); RCCL proper runs whenever
 *                     the devices are distinct.
 * Synchronous.  per_ctx (n entries) may be NULL.  Several contexts may share a device (tests; no speed-up).  Not re-entrant:
 * calls that share a context -- or, with NERF_GATHER_RCCL, a device (the communicators are cached per device list) -- must not
 * overlap.  Errors of any band are reported on ctxs[0]. */
enum { NERF_GATHER_HOST = 0, NERF_GATHER_PEER = 1, NERF_GATHER_RCCL = 2 };
int nerf_render_image_multi(nerf_ctx *const *ctxs, int n, const nerf_camera *cam, const nerf_render_opts *opts, int gather,
                            float *rgb_out, nerf_stats *per_ctx /* n entries or NULL */);
/* n contexts, device_ids[i] each (NULL => devices 0..n-1); all-or-nothing. */
int nerf_create_multi(const int *device_ids, int n, nerf_ctx **out /* n entries */);
/* Frees the cached RCCL communicators of NERF_GATHER_RCCL (optional; call after the contexts are idle). */
void nerf_multi_release(void);

/* Rows of band band_index when window_rows rows are split over band_count bands (nerf_render_opts.band_*); host-only.  Negative
 * (NERF_ERR_INVALID) on bad arguments. */
int nerf_band_rows(int window_rows, int band_index, int band_count, int band_stripe_rows);

/* Accumulated device time of the dominant (fine- or coarse-only-MLP) kernel since the last reset: blocks until the
 * recorded events have completed.  Used by bench.py for the roofline line. */
int nerf_kernel_time_query(nerf_ctx *ctx, double *ms_dominant_mlp, uint64_t *points_dominant_mlp, uint32_t *n_launches,
                           int reset);

/* Diagnostic: median in-kernel shader clock (MHz) of the last fine-network MLP launch, from s_memtime / s_memrealtime
 * stamps around the tile loop.  Only available when NERF_DEBUG_CLOCK=1 was set before nerf_create. */
int nerf_debug_shader_clock_mhz(nerf_ctx *ctx, double *mhz);

/* ---- host helpers around the path ------------------------------------------------------------------------ */
/* camera_from_samples (src/lib.rs:614-645): reads near, far, camera_origin, camera_forward, camera_up, hwf. */
int nerf_camera_from_json(const char *json_path, int width, int height, nerf_camera *out);
int nerf_camera_from_values(float near_, float far_, const float origin[3], const float forward[3],
                            const float up[3], const float hwf[3], int width, int height, nerf_camera *out);
/* Camera from a 3x4 camera-to-world pose (row-major; columns = right, up, -forward, origin -- the layout of
 * "camera_matrix" in tf_reference_samples.json, which the reference reads but never uses).  focal in pixels of a
 * ref_w x ref_h image (hwf); the result equals nerf_camera_from_values(origin, -col2, col1, ...). */
int nerf_camera_from_pose(const float c2w[12], float ref_h, float ref_w, float focal, float near_, float far_, int width,
                          int height, nerf_camera *out);
/* save_ppm (src/lib.rs:567-580): P6, (clamp(v,0,1)*255+0.5) as u8 */
int nerf_save_ppm(const char *path, int width, int height, const float *rgb);
void nerf_quantize_rgb8(const float *rgb, size_t n_pixels, uint8_t *out);
/* pixels_to_rgba (src/lib.rs:582-592; the reference's wasm canvas path): the same quantisation, alpha = 255 */
void nerf_quantize_rgba8(const float *rgb, size_t n_pixels, uint8_t *out /* 4 n_pixels */);

/* ---- stage entry points (device execution, host buffers): the individual functions of render_block, exposed so
 * that a host that owns ray setup can call them and so that each stage has its own parity test ------------- */
/* Camera::get_ray_dir (src/lib.rs:213-231) for the rectangle [y0,y0+h) x [x0,x0+w); normalize != 0 applies
 * Vec3::normalize (src/lib.rs:371).  out: h x w x 3 */
int nerf_stage_ray_dirs(nerf_ctx *ctx, const nerf_camera *cam, int x0, int y0, int w, int h, int normalize,
                        float *dirs_out);
/* stratified_samples (src/lib.rs:233-248) for the same rectangle; out: h x w x count */
int nerf_stage_stratified(nerf_ctx *ctx, const nerf_camera *cam, int x0, int y0, int w, int h, int count,
                          uint64_t seed, float *t_out);
/* compute_weights + sample_importance + merge/sort (src/lib.rs:250-351, 414-420) for n_rays rays.
 * u (n_rays x nf) may be NULL => Philox stream 1 of pixel_index[ray].  Outputs may be NULL except t_fine. */
int nerf_stage_resample(nerf_ctx *ctx, size_t n_rays, int nc, int nf, float far_, uint64_t seed,
                        const uint32_t *pixel_index, const float *t_coarse, const float *sigma_coarse, const float *u,
                        float *w_out, float *cdf_out, float *t_new_out, float *t_fine_out);
/* hybrid_sampling's per-ray decision (nerf_render_opts.hybrid_sampling), exposed so that its promise can be tested directly: the same
 * kernel, inputs and RNG as nerf_stage_resample; flags_out[r] = 1 iff ray r would be redone in f32 (a draw predicted to move by more
 * than tau in t under the split arithmetics' density error, or a transmittance within 0.1 % of the cut).  tau = 0 selects the
 * context's threshold (1e-5).  t_new_out (n_rays x nf, the unsorted draws) optional. */
int nerf_stage_hybrid_flags(nerf_ctx *ctx, size_t n_rays, int nc, int nf, float far_, uint64_t seed,
                            const uint32_t *pixel_index, const float *t_coarse, const float *sigma_coarse, const float *u,
                            float tau, uint8_t *flags_out, float *t_new_out);
/* integrate_ray (src/lib.rs:176-195); w_out (n_rays x n) optional */
int nerf_stage_integrate(nerf_ctx *ctx, size_t n_rays, int n, float far_, const float *rgb_aos, const float *sigma,
                         const float *t, float *rgb_out, float *w_out);

/* "" for the product build.  Tuning / timing-only builds (make variant: some of their switches make results WRONG on purpose)
 * report "NAME: compile definitions"; a host should refuse such a library outside experiments (the Python loader does). */
const char *nerf_build_variant(void);
/* ABI version (currently 5): bumped on any signature or struct change (2: multi-GPU entry points, skip_dead, n_exec_* statistics; 3: nerf_stats.
 * n_nonfinite_points, nerf_check_network_blob, nerf_stage_hybrid_flags, nerf_build_variant; 4: nerf_render_opts.certify_zero; 5: nerf_stats.
 * n_certify_* / certify_margin / certify_headroom / certify_max_error, renders fail on n_nonfinite_points != 0,
 * nerf_render_opts.band_*, nerf_band_rows, nerf_debug_certify_policy). */
int nerf_abi_version(void);
/* sizeof(nerf_camera), sizeof(nerf_render_opts), sizeof(nerf_stats) as this library was built: lets a binding written in
 * another language (the Rust `-sys` crate, ctypes) check its struct mirrors at start-up. */
void nerf_abi_struct_sizes(size_t *camera, size_t *render_opts, size_t *stats);

#ifdef __cplusplus
}
#endif
#endif
