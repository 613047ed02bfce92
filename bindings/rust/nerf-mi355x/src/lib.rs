//! Safe wrapper over `nerf-mi355x-sys`: the two seams of the reference this library replaces, with the
//! reference's own shapes --
//!   * `Gpu::forward_batch`  = `Network::forward_batch(&self, points: &Matrix /*3 x B SoA*/, view_dirs: &[Vec3])`
//!                             (reference src/network.rs:197-237)
//!   * `Gpu::render_image` / `Node::render_image` = `render_image(&coarse, &fine, &camera, fine_samples_per_ray)`
//!                             (reference src/lib.rs:474-565; `Node` = its rayon fan-out, :533-557, over the GPUs of one node)
//! Errors are `Result<_, Error>` with the library's message (the reference panics with the same text, src/lib.rs:36,118,127,
//! 483-501); contexts are freed on drop.  NOT COMPILED in the build image (no Rust toolchain there): kept small on purpose.
use nerf_mi355x_sys as sys;
use std::ffi::{CStr, CString};
use std::path::Path;

pub use sys::{nerf_camera as Camera, nerf_render_opts as RenderOpts, nerf_stats as Stats};

#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "[{}] {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(ctx: *const sys::nerf_ctx, rc: i32) -> Result<(), Error> {
    if rc == sys::NERF_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::nerf_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

fn cpath(p: &Path) -> Result<CString, Error> {
    CString::new(p.to_string_lossy().as_bytes()).map_err(|_| Error { code: sys::NERF_ERR_INVALID, message: "path contains NUL".into() })
}

/// How `Node::render_image` brings the row bands together (include/nerf_mi355x.h, `NERF_GATHER_*`).
#[derive(Clone, Copy, Debug)]
pub enum Gather {
    Host = sys::NERF_GATHER_HOST as isize,
    Peer = sys::NERF_GATHER_PEER as isize,
    Rccl = sys::NERF_GATHER_RCCL as isize,
}

/// `camera_from_samples` (reference src/lib.rs:614-645) from the scene's tf_reference_samples.json.
pub fn camera_from_json(json: &Path, width: i32, height: i32) -> Result<Camera, Error> {
    let mut cam = Camera::default();
    check(std::ptr::null(), unsafe { sys::nerf_camera_from_json(cpath(json)?.as_ptr(), width, height, &mut cam) })?;
    Ok(cam)
}

/// `save_ppm` (reference src/lib.rs:567-580).
pub fn save_ppm(path: &Path, width: i32, height: i32, rgb: &[f32]) -> Result<(), Error> {
    assert_eq!(rgb.len(), (width * height * 3) as usize, "pixels.len() != width * height"); // src/lib.rs:569
    check(std::ptr::null(), unsafe { sys::nerf_save_ppm(cpath(path)?.as_ptr(), width, height, rgb.as_ptr()) })
}

/// One GPU with both networks resident (`coarse_network` / `fine_network` of render_cli_image, src/lib.rs:651-652).
pub struct Gpu {
    ctx: *mut sys::nerf_ctx,
}

// a context is single-caller but may move between threads
unsafe impl Send for Gpu {}

impl Gpu {
    /// `root` = the scene directory holding coarse/ and fine/ (`load_network_from_dir`, src/lib.rs:108-174).
    pub fn new(device: i32, root: &Path) -> Result<Self, Error> {
        sys::check_layouts().map_err(|m| Error { code: sys::NERF_ERR_STATE, message: m })?;
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { sys::nerf_create(device, &mut ctx) })?;
        let gpu = Gpu { ctx }; // dropped (context destroyed) if a load below fails
        for (which, sub) in [(sys::NERF_NET_COARSE, "coarse"), (sys::NERF_NET_FINE, "fine")] {
            check(gpu.ctx, unsafe { sys::nerf_load_network_dir(gpu.ctx, which, cpath(&root.join(sub))?.as_ptr()) })?;
        }
        Ok(gpu)
    }

    /// points: 3 x B, row-major SoA (the reference's `Matrix`); view_dirs: B x 3.  Returns (colours B x 3, sigma B).
    pub fn forward_batch(&self, fine: bool, points: &[f32], view_dirs: &[f32]) -> Result<(Vec<f32>, Vec<f32>), Error> {
        assert_eq!(points.len() % 3, 0);
        let n = points.len() / 3;
        assert_eq!(view_dirs.len(), 3 * n, "one view direction per column"); // debug_assert_eq!(batch, view_dirs.len())
        let (mut rgb, mut sigma) = (vec![0f32; 3 * n], vec![0f32; n]);
        let which = if fine { sys::NERF_NET_FINE } else { sys::NERF_NET_COARSE };
        check(self.ctx, unsafe {
            sys::nerf_forward_batch(self.ctx, which, points.as_ptr(), view_dirs.as_ptr(), n, rgb.as_mut_ptr(), sigma.as_mut_ptr())
        })?;
        Ok((rgb, sigma))
    }

    /// Linear RGB, pixel (i, j) at `(i * w + j) * 3` (image[i * nx + j], src/lib.rs:552-557); `opts.n_coarse` = camera.samples_per_ray.
    pub fn render_image(&self, cam: &Camera, opts: &RenderOpts) -> Result<(Vec<f32>, Stats), Error> {
        let (w, h) = if opts.crop_w > 0 || opts.crop_h > 0 { (opts.crop_w, opts.crop_h) } else { (cam.nx, cam.ny) };
        let mut image = vec![0f32; (w.max(0) as usize) * (h.max(0) as usize) * 3];
        let mut stats = Stats::default();
        check(self.ctx, unsafe { sys::nerf_render_image(self.ctx, cam, opts, image.as_mut_ptr(), &mut stats) })?;
        Ok((image, stats))
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { sys::nerf_destroy(self.ctx) }
    }
}

/// All GPUs of one node: one context per device, weights replicated, row bands gathered into one image.
pub struct Node {
    gpus: Vec<Gpu>,
}

impl Node {
    pub fn new(devices: &[i32], root: &Path) -> Result<Self, Error> {
        Ok(Node { gpus: devices.iter().map(|&d| Gpu::new(d, root)).collect::<Result<_, _>>()? })
    }

    pub fn render_image(&self, cam: &Camera, opts: &RenderOpts, gather: Gather) -> Result<(Vec<f32>, Vec<Stats>), Error> {
        let (w, h) = if opts.crop_w > 0 || opts.crop_h > 0 { (opts.crop_w, opts.crop_h) } else { (cam.nx, cam.ny) };
        let mut image = vec![0f32; (w.max(0) as usize) * (h.max(0) as usize) * 3];
        let mut stats = vec![Stats::default(); self.gpus.len()];
        let ctxs: Vec<*mut sys::nerf_ctx> = self.gpus.iter().map(|g| g.ctx).collect();
        let first = ctxs.first().copied().unwrap_or(std::ptr::null_mut());
        check(first, unsafe {
            sys::nerf_render_image_multi(ctxs.as_ptr(), ctxs.len() as i32, cam, opts, gather as i32, image.as_mut_ptr(), stats.as_mut_ptr())
        })?;
        Ok((image, stats))
    }
}
