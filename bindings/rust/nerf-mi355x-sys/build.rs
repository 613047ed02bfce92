// Links libnerf_mi355x.so; NERF_MI355X_LIB_DIR is the directory that holds it (the repo's nerf-rs_amd/).
fn main() {
    let dir = std::env::var("NERF_MI355X_LIB_DIR").expect("set NERF_MI355X_LIB_DIR to the directory holding libnerf_mi355x.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=nerf_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=NERF_MI355X_LIB_DIR");
}
