// nerf_host_api.cpp -- the HOST-ONLY entry points of the C ABI (include/nerf_mi355x.h): everything that parses files from disk or
// converts host buffers and never touches the device -- weight-directory validation and packing, the packed-blob reader, camera
// construction (incl. the hand-written JSON reader), PPM writer, quantisers, the split diagnostics.  No HIP header is included,
// so this file and host_util.cpp also build with plain g++ under AddressSanitizer + UBSan (`make host-asan`; driven by
// tests/test_host_asan.py on the CPU box with truncated / oversized / malformed inputs).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"
#include "host_util.h"
#include "mlp_layout.h"

using namespace nerfhost;

namespace nerfhost {

static thread_local std::string g_err; // message of the last failing context-free call on this thread

int fail_noctx(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

const char *last_error_noctx() { return g_err.c_str(); }

static const char kBlobMagic[8] = {'N', 'R', 'F', 'M', 'I', '3', '5', '5'};

// Blob layout (include/nerf_mi355x.h): 16-byte header {"NRFMI355", u32 version = 1, u32 n_floats} + weight stream + small block.
// The sizes are fixed by this build's layout; anything else -- short file, trailing bytes, other version -- is refused.
int read_blob_file(const std::string &path, std::vector<float> &ws, std::vector<float> &sm, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "read blob: " + path; return NERF_ERR_IO; }
    char magic[8]; uint32_t hdr[2] = {0, 0};
    const size_t nw = (size_t)nerfmlp::kChunksFull * nerfmlp::kChunkFloats, ns = nerfmlp::kSmallFloats;
    ws.assign(nw, 0.f); sm.assign(ns, 0.f);
    const bool ok = fread(magic, 1, 8, f) == 8 && fread(hdr, 4, 2, f) == 2 && !memcmp(magic, kBlobMagic, 8) && hdr[0] == 1u &&
                    hdr[1] == nw + ns && fread(ws.data(), 4, nw, f) == nw && fread(sm.data(), 4, ns, f) == ns && fgetc(f) == EOF;
    fclose(f);
    if (!ok) { err = "not a version-1 packed network blob for this build: " + path; return NERF_ERR_SHAPE; }
    return NERF_OK;
}

} // namespace nerfhost

// No C++ exception may cross the C ABI (a Rust caller would abort).
#define NERF_HOST_CATCH                                                                                             \
    catch (const std::bad_alloc &) { return fail_noctx(NERF_ERR_IO, "out of host memory"); }                        \
    catch (const std::exception &e) { return fail_noctx(NERF_ERR_INVALID, std::string("internal error: ") + e.what()); } \
    catch (...) { return fail_noctx(NERF_ERR_INVALID, "internal error"); }

#ifndef NERF_BUILD_VARIANT
#define NERF_BUILD_VARIANT ""
#endif

extern "C" {

const char *nerf_build_variant(void) { return NERF_BUILD_VARIANT; }

int nerf_pack_network_dir(const char *dir, const char *blob_path) try {
    if (!dir || !blob_path) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    std::map<std::string, Tensor> params;
    std::string err;
    int rc = read_tensor_dir(dir, params, err);
    if (rc) return fail_noctx(rc, err);
    HostNet hn;
    rc = assemble_net(params, hn, err);
    if (rc) return fail_noctx(rc, err);
    std::vector<float> ws, sm;
    pack_network(hn, ws, sm);
    FILE *f = fopen(blob_path, "wb");
    if (!f) return fail_noctx(NERF_ERR_IO, std::string("cannot create ") + blob_path);
    const uint32_t hdr[2] = {1u, (uint32_t)(ws.size() + sm.size())};
    const bool ok = fwrite(kBlobMagic, 1, 8, f) == 8 && fwrite(hdr, 4, 2, f) == 2 &&
                    fwrite(ws.data(), 4, ws.size(), f) == ws.size() && fwrite(sm.data(), 4, sm.size(), f) == sm.size();
    fclose(f);
    return ok ? NERF_OK : fail_noctx(NERF_ERR_IO, std::string("short write ") + blob_path);
} NERF_HOST_CATCH

int nerf_camera_from_pose(const float c2w[12], float ref_h, float ref_w, float focal, float near_, float far_, int width,
                          int height, nerf_camera *out) try {
    if (!c2w || !out) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    const float origin[3] = {c2w[3], c2w[7], c2w[11]};
    const float forward[3] = {-c2w[2], -c2w[6], -c2w[10]};
    const float up[3] = {c2w[1], c2w[5], c2w[9]};
    const float hwf[3] = {ref_h, ref_w, focal};
    camera_from_values(near_, far_, origin, forward, up, hwf, width, height, out);
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_check_network_dir(const char *dir) try {
    if (!dir) return fail_noctx(NERF_ERR_INVALID, "dir is NULL");
    std::map<std::string, Tensor> params;
    std::string err;
    int rc = read_tensor_dir(dir, params, err);
    if (rc) return fail_noctx(rc, err);
    HostNet hn;
    rc = assemble_net(params, hn, err);
    if (rc) return fail_noctx(rc, err);
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_debug_split_bf16x3(const float *values, size_t n, uint16_t *parts) try {
    if ((!values || !parts) && n) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < n; ++i) split_bf16x3(values[i], parts + 3 * i);
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_debug_certify_policy(float margin, uint64_t audited, uint64_t violations, float headroom, float max_error, float *new_margin) {
    if (!(margin > 0.0f)) return fail_noctx(NERF_ERR_INVALID, "margin must be > 0");
    return certify_policy(margin, audited, violations, headroom, max_error, new_margin);
}

int nerf_debug_split_f16x2(const float *values, size_t n, uint16_t *parts) try {
    if ((!values || !parts) && n) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < n; ++i) split_f16x2(values[i], parts + 2 * i);
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_debug_pack_network_dir(const char *dir, float *wstream, size_t wstream_cap, float *small, size_t small_cap,
                                size_t *wstream_len, size_t *small_len) try {
    if (!dir) return fail_noctx(NERF_ERR_INVALID, "dir is NULL");
    std::map<std::string, Tensor> params;
    std::string err;
    int rc = read_tensor_dir(dir, params, err);
    if (rc) return fail_noctx(rc, err);
    HostNet hn;
    rc = assemble_net(params, hn, err);
    if (rc) return fail_noctx(rc, err);
    std::vector<float> ws, sm;
    pack_network(hn, ws, sm);
    if (wstream_len) *wstream_len = ws.size();
    if (small_len) *small_len = sm.size();
    if (wstream) { if (wstream_cap < ws.size()) return fail_noctx(NERF_ERR_INVALID, "wstream buffer too small"); memcpy(wstream, ws.data(), ws.size() * sizeof(float)); }
    if (small) { if (small_cap < sm.size()) return fail_noctx(NERF_ERR_INVALID, "small buffer too small"); memcpy(small, sm.data(), sm.size() * sizeof(float)); }
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_camera_from_json(const char *path, int width, int height, nerf_camera *out) try {
    if (!path || !out) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    std::string err;
    const int rc = camera_from_json(path, width, height, out, err);
    return rc ? fail_noctx(rc, err) : NERF_OK;
} NERF_HOST_CATCH

int nerf_camera_from_values(float near_, float far_, const float origin[3], const float forward[3], const float up[3],
                            const float hwf[3], int width, int height, nerf_camera *out) try {
    if (!origin || !forward || !up || !hwf || !out) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    camera_from_values(near_, far_, origin, forward, up, hwf, width, height, out);
    return NERF_OK;
} NERF_HOST_CATCH

int nerf_save_ppm(const char *path, int width, int height, const float *rgb) try {
    if (!path || !rgb) return fail_noctx(NERF_ERR_INVALID, "NULL argument");
    std::string err;
    const int rc = save_ppm(path, width, height, rgb, err);
    return rc ? fail_noctx(rc, err) : NERF_OK;
} NERF_HOST_CATCH

void nerf_quantize_rgb8(const float *rgb, size_t n_pixels, uint8_t *out) { quantize_rgb8(rgb, n_pixels, out); }

void nerf_quantize_rgba8(const float *rgb, size_t n_pixels, uint8_t *out) {
    for (size_t i = 0; i < n_pixels; ++i) { // no allocation: nothing here can throw across the ABI
        quantize_rgb8(rgb + 3 * i, 1, out + 4 * i);
        out[4 * i + 3] = 255;
    }
}

int nerf_check_network_blob(const char *blob_path) try {
    if (!blob_path) return fail_noctx(NERF_ERR_INVALID, "blob_path is NULL");
    std::vector<float> ws, sm;
    std::string err;
    const int rc = read_blob_file(blob_path, ws, sm, err);
    return rc ? fail_noctx(rc, err) : NERF_OK;
} NERF_HOST_CATCH

} // extern "C"
