// host_asan_driver.cpp -- drives the HOST-ONLY entry points of the C ABI (nerf_host_api.cpp + host_util.cpp) in a plain g++ build with
// AddressSanitizer + UndefinedBehaviorSanitizer (`make host-asan`; GPU ASan is not available on this pool, and none of this code
// touches the device).  tests/test_host_asan.py feeds it valid, truncated, oversized and malformed weight directories, blobs and
// camera JSON files: every call must come back with a status code and a message -- a sanitizer report aborts with a non-zero exit.
//   host_asan_driver check_dir <dir> | pack_dir <dir> <blob> | check_blob <blob> | camera_json <json> <w> <h> |
//                    debug_pack <dir> | quantize | save_ppm <path> <w> <h> | split
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"
#include "host_util.h"

static int report(const char *what, int rc) {
    printf("%s rc=%d msg=%s\n", what, rc, rc ? nerfhost::last_error_noctx() : "");
    return 0; // an error CODE is a correct answer; only a crash / sanitizer report fails the run
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "check_dir" && argc == 3) return report("check_dir", nerf_check_network_dir(argv[2]));
    if (cmd == "pack_dir" && argc == 4) return report("pack_dir", nerf_pack_network_dir(argv[2], argv[3]));
    if (cmd == "check_blob" && argc == 3) return report("check_blob", nerf_check_network_blob(argv[2]));
    if (cmd == "camera_json" && argc == 5) {
        nerf_camera cam;
        memset(&cam, 0, sizeof cam);
        const int rc = nerf_camera_from_json(argv[2], atoi(argv[3]), atoi(argv[4]), &cam);
        if (!rc) printf("camera %d %d %.9g %.9g near %.9g far %.9g\n", cam.nx, cam.ny, cam.alpha_width, cam.alpha_height, cam.near_, cam.far_);
        return report("camera_json", rc);
    }
    if (cmd == "debug_pack" && argc == 3) {
        size_t nw = 0, ns = 0;
        int rc = nerf_debug_pack_network_dir(argv[2], nullptr, 0, nullptr, 0, &nw, &ns);
        if (!rc) {
            std::vector<float> ws(nw), sm(ns);
            rc = nerf_debug_pack_network_dir(argv[2], ws.data(), ws.size(), sm.data(), sm.size(), &nw, &ns);
            if (!rc) rc = nerf_debug_pack_network_dir(argv[2], ws.data(), ws.size() - 1, sm.data(), sm.size(), &nw, &ns) == NERF_ERR_INVALID ? 0 : 99; // short buffer refused
            double s = 0; for (float v : ws) s += v; for (float v : sm) s += v;
            printf("packed %zu + %zu floats, sum %.6f\n", nw, ns, s);
        }
        return report("debug_pack", rc);
    }
    if (cmd == "quantize") { // clamp + NaN/inf through the quantisers (src/lib.rs:573-577, :582-592)
        const float v[] = {-1.f, 0.f, 0.5f, 1.f, 2.f, NAN, INFINITY, -INFINITY, 1e-9f, 0.999999f, 0.25f, 0.75f};
        uint8_t a[12], b[16];
        nerf_quantize_rgb8(v, 4, a);
        nerf_quantize_rgba8(v, 4, b);
        for (int i = 0; i < 12; ++i) printf("%d ", a[i]);
        printf("| ");
        for (int i = 0; i < 16; ++i) printf("%d ", b[i]);
        printf("\n");
        nerf_quantize_rgb8(v, 0, a);
        return report("quantize", 0);
    }
    if (cmd == "save_ppm" && argc == 5) {
        const int w = atoi(argv[3]), h = atoi(argv[4]);
        std::vector<float> img(w > 0 && h > 0 ? (size_t)w * h * 3 : 0);
        for (size_t i = 0; i < img.size(); ++i) img[i] = (float)(i % 97) / 96.f - 0.01f;
        return report("save_ppm", nerf_save_ppm(argv[2], w, h, img.data()));
    }
    if (cmd == "split") {
        const float v[] = {0.f, -0.f, 1.f, -3.14159274f, 65504.f, 7e4f, 1e-8f, 6e-8f, 1e30f, -1e-30f, NAN, INFINITY};
        uint16_t p3[3 * 12], p2[2 * 12];
        int rc = nerf_debug_split_bf16x3(v, 12, p3);
        if (!rc) rc = nerf_debug_split_f16x2(v, 12, p2);
        if (!rc) rc = nerf_debug_split_f16x2(nullptr, 0, nullptr);
        for (int i = 0; i < 36; ++i) printf("%04x ", p3[i]);
        printf("| ");
        for (int i = 0; i < 24; ++i) printf("%04x ", p2[i]);
        printf("\n");
        return report("split", rc);
    }
    fprintf(stderr, "usage: see the header comment\n");
    return 2;
}
