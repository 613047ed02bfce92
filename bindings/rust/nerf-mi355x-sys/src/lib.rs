//! Raw bindings to libnerf_mi355x.so.  One-to-one with include/nerf_mi355x.h (ABI version 5): every function the header
//! declares is declared here with the same number of arguments (tests/test_host_logic.py parses both and compares);
//! layouts are `#[repr(C)]` mirrors of `nerf_camera`, `nerf_render_opts`, `nerf_stats` -- call `check_layouts()` once at
//! start-up to compare their sizes with the library's (`nerf_abi_struct_sizes`).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct nerf_ctx {
    _private: [u8; 0],
}

/// `struct Camera` of the reference (src/lib.rs:197-211); `samples_per_ray` lives in `nerf_render_opts::n_coarse`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct nerf_camera {
    pub nx: i32,
    pub ny: i32,
    pub alpha_width: f32,
    pub alpha_height: f32,
    pub pos: [f32; 3],
    pub dir: [f32; 3],
    pub up: [f32; 3],
    pub near: f32,
    pub far: f32,
}

/// `nerf_render_opts::mlp_dtype` / `nerf_forward_batch_ex`: the MLP arithmetic (include/nerf_mi355x.h).
pub const NERF_MLP_F32: i32 = 0;
pub const NERF_MLP_BF16: i32 = 1;
pub const NERF_MLP_BF16X3: i32 = 2;
pub const NERF_MLP_F16X2: i32 = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct nerf_render_opts {
    pub n_coarse: i32,
    pub n_fine: i32,
    pub coarse_only: i32,
    pub crop_x0: i32,
    pub crop_y0: i32,
    pub crop_w: i32,
    pub crop_h: i32,
    pub ssaa: i32,
    pub seed: u64,
    pub mlp_dtype: i32,
    pub skip_empty: i32,
    pub skip_dead: i32,
    pub hybrid_sampling: i32,
    pub certify_zero: i32,
    pub band_index: i32,
    pub band_count: i32,
    pub band_stripe_rows: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct nerf_stats {
    pub n_rays: u64,
    pub n_coarse_points: u64,
    pub n_fine_points: u64,
    pub ms_total: f64,
    pub ms_coarse_mlp: f64,
    pub ms_fine_mlp: f64,
    pub ms_other: f64,
    pub n_mlp_launches: u32,
    pub n_passes: u32,
    pub n_colour_skipped_points: u64,
    pub n_exec_coarse_trunk: u64,
    pub n_exec_fine_trunk: u64,
    pub n_exec_colour: u64,
    pub n_hybrid_rays: u64,
    pub n_nonfinite_points: u64,
    pub n_certify_audited: u64,
    pub n_certify_violations: u64,
    pub n_certify_retries: u32,
    pub n_certify_fallback_rays: u32,
    pub certify_margin: [f32; 2],
    pub certify_headroom: [f32; 2],
    pub certify_max_error: [f32; 2],
}

/// `gather` of `nerf_render_image_multi`
pub const NERF_GATHER_HOST: c_int = 0;
pub const NERF_GATHER_PEER: c_int = 1;
pub const NERF_GATHER_RCCL: c_int = 2;

pub const NERF_OK: c_int = 0;
pub const NERF_ERR_INVALID: c_int = -1;
pub const NERF_ERR_IO: c_int = -2;
pub const NERF_ERR_MISSING: c_int = -3;
pub const NERF_ERR_SHAPE: c_int = -4;
pub const NERF_ERR_HIP: c_int = -5;
pub const NERF_ERR_STATE: c_int = -6;
pub const NERF_ERR_PARSE: c_int = -7;
pub const NERF_NET_COARSE: c_int = 0;
pub const NERF_NET_FINE: c_int = 1;

extern "C" {
    pub fn nerf_abi_version() -> c_int;
    pub fn nerf_build_variant() -> *const c_char;
    pub fn nerf_abi_struct_sizes(camera: *mut usize, render_opts: *mut usize, stats: *mut usize);
    pub fn nerf_create(device_id: c_int, out: *mut *mut nerf_ctx) -> c_int;
    pub fn nerf_destroy(ctx: *mut nerf_ctx);
    pub fn nerf_last_error(ctx: *const nerf_ctx) -> *const c_char;
    pub fn nerf_device_info(ctx: *const nerf_ctx, n_cus: *mut c_int, arch_name: *mut c_char, len: usize) -> c_int;
    pub fn nerf_load_network_dir(ctx: *mut nerf_ctx, which: c_int, dir: *const c_char) -> c_int;
    pub fn nerf_load_network_tensors(ctx: *mut nerf_ctx, which: c_int, n: c_int, names: *const *const c_char,
                                     dims: *const i64, data: *const *const f32) -> c_int;
    pub fn nerf_check_network_dir(dir: *const c_char) -> c_int;
    pub fn nerf_check_network_blob(blob_path: *const c_char) -> c_int;
    pub fn nerf_pack_network_dir(dir: *const c_char, blob_path: *const c_char) -> c_int;
    pub fn nerf_load_network_blob(ctx: *mut nerf_ctx, which: c_int, blob_path: *const c_char) -> c_int;
    pub fn nerf_debug_pack_network_dir(dir: *const c_char, wstream: *mut f32, wstream_cap: usize, small: *mut f32,
                                       small_cap: usize, wstream_len: *mut usize, small_len: *mut usize) -> c_int;
    pub fn nerf_debug_split_bf16x3(values: *const f32, n: usize, parts: *mut u16) -> c_int;
    pub fn nerf_debug_split_f16x2(values: *const f32, n: usize, parts: *mut u16) -> c_int;
    pub fn nerf_debug_certify_policy(margin: f32, audited: u64, violations: u64, headroom: f32, max_error: f32, new_margin: *mut f32) -> c_int;
    pub fn nerf_debug_shader_clock_mhz(ctx: *mut nerf_ctx, mhz: *mut f64) -> c_int;
    pub fn nerf_forward_batch(ctx: *mut nerf_ctx, which: c_int, pts_soa: *const f32, dirs_aos: *const f32, n: usize,
                              rgb_aos: *mut f32, sigma: *mut f32) -> c_int;
    pub fn nerf_forward_batch_ex(ctx: *mut nerf_ctx, which: c_int, mlp_dtype: c_int, pts_soa: *const f32, dirs_aos: *const f32,
                                 n: usize, rgb_aos: *mut f32, sigma: *mut f32) -> c_int;
    pub fn nerf_forward_batch_device(ctx: *mut nerf_ctx, which: c_int, d_pts_soa: *const f32, d_dirs_aos: *const f32,
                                     n: usize, d_rgb_aos: *mut f32, d_sigma: *mut f32, stream: *mut c_void) -> c_int;
    pub fn nerf_render_image(ctx: *mut nerf_ctx, cam: *const nerf_camera, opts: *const nerf_render_opts,
                             rgb_out: *mut f32, stats: *mut nerf_stats) -> c_int;
    pub fn nerf_render_image_device(ctx: *mut nerf_ctx, cam: *const nerf_camera, opts: *const nerf_render_opts,
                                    d_rgb_out: *mut f32, stream: *mut c_void, stats: *mut nerf_stats) -> c_int;
    /// render_image over several GPUs: `ctxs[i]` = one context per device, row bands on per-context host threads + streams,
    /// gathered into `rgb_out` by `gather` (NERF_GATHER_*).  The reference's counterpart is the rayon fan-out, src/lib.rs:533-557.
    pub fn nerf_render_image_multi(ctxs: *const *mut nerf_ctx, n: c_int, cam: *const nerf_camera, opts: *const nerf_render_opts,
                                   gather: c_int, rgb_out: *mut f32, per_ctx: *mut nerf_stats) -> c_int;
    pub fn nerf_create_multi(device_ids: *const c_int, n: c_int, out: *mut *mut nerf_ctx) -> c_int;
    pub fn nerf_multi_release();
    pub fn nerf_band_rows(window_rows: c_int, band_index: c_int, band_count: c_int, band_stripe_rows: c_int) -> c_int;
    pub fn nerf_kernel_time_query(ctx: *mut nerf_ctx, ms: *mut f64, points: *mut u64, n_launches: *mut u32, reset: c_int) -> c_int;
    pub fn nerf_camera_from_json(json_path: *const c_char, width: c_int, height: c_int, out: *mut nerf_camera) -> c_int;
    pub fn nerf_camera_from_pose(c2w: *const f32, ref_h: f32, ref_w: f32, focal: f32, near: f32, far: f32, width: c_int,
                                 height: c_int, out: *mut nerf_camera) -> c_int;
    pub fn nerf_camera_from_values(near: f32, far: f32, origin: *const f32, forward: *const f32, up: *const f32,
                                   hwf: *const f32, width: c_int, height: c_int, out: *mut nerf_camera) -> c_int;
    pub fn nerf_save_ppm(path: *const c_char, width: c_int, height: c_int, rgb: *const f32) -> c_int;
    pub fn nerf_quantize_rgb8(rgb: *const f32, n_pixels: usize, out: *mut u8);
    pub fn nerf_quantize_rgba8(rgb: *const f32, n_pixels: usize, out: *mut u8);
    pub fn nerf_stage_ray_dirs(ctx: *mut nerf_ctx, cam: *const nerf_camera, x0: c_int, y0: c_int, w: c_int, h: c_int,
                               normalize: c_int, dirs_out: *mut f32) -> c_int;
    pub fn nerf_stage_stratified(ctx: *mut nerf_ctx, cam: *const nerf_camera, x0: c_int, y0: c_int, w: c_int, h: c_int,
                                 count: c_int, seed: u64, t_out: *mut f32) -> c_int;
    pub fn nerf_stage_resample(ctx: *mut nerf_ctx, n_rays: usize, nc: c_int, nf: c_int, far: f32, seed: u64,
                               pixel_index: *const u32, t_coarse: *const f32, sigma_coarse: *const f32, u: *const f32,
                               w_out: *mut f32, cdf_out: *mut f32, t_new_out: *mut f32, t_fine_out: *mut f32) -> c_int;
    pub fn nerf_stage_hybrid_flags(ctx: *mut nerf_ctx, n_rays: usize, nc: c_int, nf: c_int, far: f32, seed: u64,
                                   pixel_index: *const u32, t_coarse: *const f32, sigma_coarse: *const f32, u: *const f32,
                                   tau: f32, flags_out: *mut u8, t_new_out: *mut f32) -> c_int;
    pub fn nerf_stage_integrate(ctx: *mut nerf_ctx, n_rays: usize, n: c_int, far: f32, rgb_aos: *const f32,
                                sigma: *const f32, t: *const f32, rgb_out: *mut f32, w_out: *mut f32) -> c_int;
}

/// Compares the sizes of the `#[repr(C)]` mirrors above with the library's own structs; call once before anything else.
pub fn check_layouts() -> Result<(), String> {
    let (mut a, mut b, mut c) = (0usize, 0usize, 0usize);
    unsafe { nerf_abi_struct_sizes(&mut a, &mut b, &mut c) };
    let mine = (std::mem::size_of::<nerf_camera>(), std::mem::size_of::<nerf_render_opts>(), std::mem::size_of::<nerf_stats>());
    if (a, b, c) == mine && unsafe { nerf_abi_version() } == 5 {
        Ok(())
    } else {
        Err(format!("libnerf_mi355x: ABI {} with struct sizes {:?}, this crate expects ABI 5 with {:?}", unsafe { nerf_abi_version() }, (a, b, c), mine))
    }
}
