//! The reference's `cargo run --release` (src/main.rs:1-3 -> render_cli_image, src/lib.rs:647-677) on the MI355X
//! library: same scene directory, same printed facts, same output.ppm.  Rust owns paths, camera and timing; the
//! networks live on the GPU.
use nerf_mi355x_sys as sys;
use std::ffi::{CStr, CString};
use std::path::Path;
use std::time::Instant;

fn check(ctx: *const sys::nerf_ctx, rc: i32) {
    if rc != sys::NERF_OK {
        // the reference panics in these situations (src/lib.rs:36,118,127,483-501)
        let msg = unsafe { CStr::from_ptr(sys::nerf_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{msg}");
    }
}

fn cstr(p: &Path) -> CString {
    CString::new(p.to_str().expect("utf-8 path")).unwrap()
}

fn main() {
    let root = std::env::var("NERF_SCENE_DIR").unwrap_or_else(|_| "lego_rust".to_string());
    let root = Path::new(&root);
    sys::check_layouts().expect("libnerf_mi355x.so does not match this crate");
    let mut ctx = std::ptr::null_mut();
    check(std::ptr::null(), unsafe { sys::nerf_create(0, &mut ctx) });
    check(ctx, unsafe { sys::nerf_load_network_dir(ctx, sys::NERF_NET_COARSE, cstr(&root.join("coarse")).as_ptr()) });
    check(ctx, unsafe { sys::nerf_load_network_dir(ctx, sys::NERF_NET_FINE, cstr(&root.join("fine")).as_ptr()) });

    let (coarse_samples_per_ray, fine_samples_per_ray) = (64, 128); // default_sample_counts, src/lib.rs:603-612
    let (width, height) = (256, 256);                                // src/lib.rs:657-658
    println!("Rendering with {} coarse samples and {} fine samples per ray", coarse_samples_per_ray, fine_samples_per_ray);

    let mut cam = sys::nerf_camera::default();
    check(std::ptr::null(), unsafe {
        sys::nerf_camera_from_json(cstr(&root.join("tf_reference_samples.json")).as_ptr(), width, height, &mut cam)
    });
    let opts = sys::nerf_render_opts { n_coarse: coarse_samples_per_ray, n_fine: fine_samples_per_ray, ..Default::default() };

    println!("Starting image rendering...");
    let render_start = Instant::now();
    let mut image = vec![0f32; (width * height * 3) as usize];
    let mut stats = sys::nerf_stats::default();
    check(ctx, unsafe { sys::nerf_render_image(ctx, &cam, &opts, image.as_mut_ptr(), &mut stats) });
    let render_duration = render_start.elapsed();
    println!("Rendering complete: {}/{} pixels (100.0%)", width * height, width * height);
    println!("Rendering completed in {:.2} seconds", render_duration.as_secs_f64());
    println!("{:.0} rays/s on the GPU (coarse MLP {:.1} ms, fine MLP {:.1} ms, other {:.1} ms)",
             stats.n_rays as f64 / (stats.ms_total * 1e-3), stats.ms_coarse_mlp, stats.ms_fine_mlp, stats.ms_other);
    check(std::ptr::null(), unsafe { sys::nerf_save_ppm(CString::new("output.ppm").unwrap().as_ptr(), width, height, image.as_ptr()) });
    unsafe { sys::nerf_destroy(ctx) };
}
