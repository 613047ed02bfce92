// nerf_cli.cpp -- the MI355X counterpart of `cargo run --release` (reference src/main.rs:1-3 ->
// render_cli_image, src/lib.rs:647-677): load lego_rust/{coarse,fine}, build the camera from
// tf_reference_samples.json, render, print the same facts, write output.ppm.  Plain C++ over the C ABI
// (include/nerf_mi355x.h) -- exactly what a Rust main.rs would do through the extern "C" block of INTEGRATION.md.
//
// With no flags it reproduces the reference's run: 256x256, 64 coarse + 128 fine samples, ./output.ppm.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [--scene DIR] [--width W] [--height H] [--coarse N] [--fine N] [--seed S] [--ssaa S]\n"
            "          [--coarse-only] [--crop X0,Y0,W,H] [--dtype f32|bf16|bf16x3|f16x2] [--skip-empty] [--skip-dead] [--hybrid-sampling] [--certify-zero]\n"
            "          [--device ID | --gpus N | --devices ID,ID,... [--gather host|peer|rccl]] [--frames K] [--out FILE.ppm]\n"
            "defaults: --scene lego_rust --width 256 --height 256 --coarse 64 --fine 128 --out output.ppm\n",
            argv0);
}

int main(int argc, char **argv) {
    std::string scene = getenv("NERF_SCENE_DIR") ? getenv("NERF_SCENE_DIR") : "lego_rust";
    std::string out = "output.ppm";
    int width = 256, height = 256, device = 0, frames = 1, gpus = 1, gather = NERF_GATHER_HOST; // src/lib.rs:657-658
    std::vector<int> devices;
    nerf_render_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.n_coarse = 64; opts.n_fine = 128; // default_sample_counts, src/lib.rs:603-612
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(argv[0]); exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--width") width = atoi(next());
        else if (a == "--height") height = atoi(next());
        else if (a == "--coarse") opts.n_coarse = atoi(next());
        else if (a == "--fine") opts.n_fine = atoi(next());
        else if (a == "--seed") opts.seed = strtoull(next(), nullptr, 10);
        else if (a == "--ssaa") opts.ssaa = atoi(next());
        else if (a == "--coarse-only") opts.coarse_only = 1;
        else if (a == "--dtype") { const std::string d = next(); if (d == "bf16") opts.mlp_dtype = NERF_MLP_BF16; else if (d == "bf16x3") opts.mlp_dtype = NERF_MLP_BF16X3; else if (d == "f16x2") opts.mlp_dtype = NERF_MLP_F16X2; else if (d != "f32") { usage(argv[0]); return 2; } }
        else if (a == "--skip-empty") opts.skip_empty = 1;
        else if (a == "--skip-dead") opts.skip_dead = 1;
        else if (a == "--hybrid-sampling") opts.hybrid_sampling = 1;
        else if (a == "--certify-zero") opts.certify_zero = 1;
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--devices") { // explicit device list, one context each (ids may repeat: several contexts on one GPU)
            devices.clear();
            for (const char *p = next(); *p;) { devices.push_back((int)strtol(p, (char **)&p, 10)); if (*p == ',') ++p; else if (*p) { usage(argv[0]); return 2; } }
            gpus = (int)devices.size();
        }
        else if (a == "--gather") { const std::string g = next(); gather = g == "peer" ? NERF_GATHER_PEER : g == "rccl" ? NERF_GATHER_RCCL : NERF_GATHER_HOST; }
        else if (a == "--device") device = atoi(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--crop") {
            if (sscanf(next(), "%d,%d,%d,%d", &opts.crop_x0, &opts.crop_y0, &opts.crop_w, &opts.crop_h) != 4) { usage(argv[0]); return 2; }
        } else { usage(argv[0]); return a == "--help" || a == "-h" ? 0 : 2; }
    }

    // one context per GPU (--gpus N: devices 0..N-1, the rayon fan-out of src/lib.rs:533-550 becomes a fan-out over devices)
    if (gpus < 1) { usage(argv[0]); return 2; }
    std::vector<nerf_ctx *> ctxs(gpus, nullptr);
    if (gpus == 1 && devices.empty() ? nerf_create(device, &ctxs[0]) : nerf_create_multi(devices.empty() ? nullptr : devices.data(), gpus, ctxs.data())) { fprintf(stderr, "error: %s\n", nerf_last_error(nullptr)); return 1; }
    nerf_ctx *ctx = ctxs[0];
    for (nerf_ctx *c : ctxs)
        if (nerf_load_network_dir(c, NERF_NET_COARSE, (scene + "/coarse").c_str()) ||
            nerf_load_network_dir(c, NERF_NET_FINE, (scene + "/fine").c_str())) {
            fprintf(stderr, "error: %s\n", nerf_last_error(c));
            return 1;
        }
    nerf_camera cam;
    if (nerf_camera_from_json((scene + "/tf_reference_samples.json").c_str(), width, height, &cam)) {
        fprintf(stderr, "error: %s\n", nerf_last_error(nullptr));
        return 1;
    }
    printf("Rendering with %d coarse samples and %d fine samples per ray\n", opts.n_coarse, opts.n_fine); // :660-663
    const int ow = opts.crop_w > 0 ? opts.crop_w : width, oh = opts.crop_h > 0 ? opts.crop_h : height;
    std::vector<float> image((size_t)ow * oh * 3);
    printf("Starting image rendering...\n"); // :667
    nerf_stats st;
    std::vector<nerf_stats> per(gpus);
    double best = 1e30;
    for (int f = 0; f < frames; ++f) {
        const auto t0 = std::chrono::steady_clock::now(); // Instant::now() :668
        if (gpus == 1 ? nerf_render_image(ctx, &cam, &opts, image.data(), &st)
                      : nerf_render_image_multi(ctxs.data(), gpus, &cam, &opts, gather, image.data(), per.data())) {
            fprintf(stderr, "error: %s\n", nerf_last_error(ctx));
            return 1;
        }
        if (gpus > 1) { // whole-job view: rays add up, device time is the slowest band's
            st = per[0];
            for (int g = 1; g < gpus; ++g) {
                st.n_rays += per[g].n_rays;
                st.n_nonfinite_points += per[g].n_nonfinite_points;
                if (per[g].ms_total > st.ms_total) { st.ms_total = per[g].ms_total; st.ms_coarse_mlp = per[g].ms_coarse_mlp; st.ms_fine_mlp = per[g].ms_fine_mlp; st.ms_other = per[g].ms_other; }
            }
        }
        if (st.n_nonfinite_points) { // a split arithmetic left its range (f16x2: an activation beyond 65 504): the image is wrong there
            fprintf(stderr, "error: %llu evaluations left the range of the selected arithmetic (nerf_stats.n_nonfinite_points); use --dtype bf16x3 or f32\n",
                    (unsigned long long)st.n_nonfinite_points);
            return 1;
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        best = secs < best ? secs : best;
        printf("Rendering complete: %llu/%llu pixels (100.0%%)\n", (unsigned long long)ow * oh, (unsigned long long)ow * oh); // :559-562
        printf("Rendering completed in %.2f seconds\n", secs); // :672-675
    }
    int n_cus = 0; char arch[64] = {0};
    nerf_device_info(ctx, &n_cus, arch, sizeof arch);
    const double flop_ray = opts.coarse_only ? opts.n_coarse * 1186816.0
                                             : opts.n_coarse * 982528.0 + (double)(opts.n_coarse + opts.n_fine) * 1186816.0;
    const bool bf16 = opts.mlp_dtype != NERF_MLP_F32;
    const double mfma_flops = opts.mlp_dtype == NERF_MLP_BF16X3 ? 6.0 : opts.mlp_dtype == NERF_MLP_F16X2 ? 3.0 : 1.0; // executed bf16 MFMA flops per algorithmic f32 flop
    if (gpus > 1) printf("%d GPUs, %s gathered by %s\n", gpus, (opts.skip_dead || opts.skip_empty || opts.certify_zero) ? "rows dealt out round-robin," : "contiguous row bands", gather == NERF_GATHER_PEER ? "xGMI peer copies" : gather == NERF_GATHER_RCCL ? "one RCCL all-gather" : "direct D2H");
    const bool skips = opts.skip_dead || opts.skip_empty || opts.certify_zero; // then less than the algorithmic work is executed: not a roofline fraction
    printf("device %s (%d CUs): %.0f rays/s (best of %d, host wall incl. D2H); device %.1f ms = coarse MLP %.1f + fine MLP %.1f + other %.1f; "
           "%s%.1f%% of the %s MFMA roofline%s\n",
           arch, n_cus, (double)st.n_rays / best, frames, st.ms_total, st.ms_coarse_mlp, st.ms_fine_mlp, st.ms_other,
           skips ? "the ALGORITHMIC work of these rays per second = " : "",
           100.0 * mfma_flops * (double)st.n_rays * flop_ray / (st.ms_total * 1e-3) / (gpus * (bf16 ? 2500e12 : 157.3e12)),
           bf16 ? "2.5 PFLOP/s bf16" : "157.3 TFLOP/s fp32", skips ? " (work provably without effect is skipped: not a utilisation figure)" : "");
    if (opts.certify_zero) // the audit of the last frame (first context): what the certificates rested on
        printf("certify_zero audit: %llu certificates evaluated all the same, %llu violations, margins %.3g / %.3g, least headroom %.3g / %.3g, largest bf16 error %.3g / %.3g, "
               "frame rendered again %u time(s), %u rays beyond their predicted cut\n", (unsigned long long)st.n_certify_audited, (unsigned long long)st.n_certify_violations,
               (double)st.certify_margin[0], (double)st.certify_margin[1], (double)st.certify_headroom[0], (double)st.certify_headroom[1],
               (double)st.certify_max_error[0], (double)st.certify_max_error[1], st.n_certify_retries, st.n_certify_fallback_rays);
    if (nerf_save_ppm(out.c_str(), ow, oh, image.data())) { fprintf(stderr, "error: %s\n", nerf_last_error(nullptr)); return 1; } // :676
    for (nerf_ctx *c : ctxs) nerf_destroy(c);
    return 0;
}
