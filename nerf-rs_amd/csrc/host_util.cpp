// host_util.cpp -- loader, packer, camera, JSON, PPM.  Build with -ffp-contract=off: the camera math must round
// exactly like the reference's (rustc never fuses a*b+c).
#include "host_util.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <sstream>

#include "mlp_layout.h"

namespace nerfhost {

// ------------------------------------------------------------------------------------------------
// Loader (reference src/lib.rs:34-42, 62-74, 108-174)
// ------------------------------------------------------------------------------------------------
int read_tensor_dir(const std::string &dir, std::map<std::string, Tensor> &out, std::string &err) {
    std::ifstream shapes(dir + "/shapes.txt");
    if (!shapes) { err = "read shapes: " + dir + "/shapes.txt"; return NERF_ERR_IO; }
    std::string line;
    while (std::getline(shapes, line)) {
        std::istringstream ls(line);
        std::string name;
        if (!(ls >> name)) continue; // blank line (the reference would panic on unwrap; be lenient)
        Tensor t;
        std::string tok;
        while (ls >> tok) {
            char *end = nullptr;
            const long long v = strtoll(tok.c_str(), &end, 10);
            if (*end != 0 || v < 0) { err = "shapes.txt: bad dimension '" + tok + "' for " + name; return NERF_ERR_PARSE; }
            t.dims.push_back(v);
        }
        const std::string path = dir + "/" + name + ".bin";
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) { err = "read tensor: " + path; return NERF_ERR_IO; }
        // file size: a directory or special file named <name>.bin gives -1 / nonsense -- an I/O error, not a 2^62-element resize
        long sz = -1;
        if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
        if (sz < 0 || fseek(f, 0, SEEK_SET) != 0 || (unsigned long)sz > (1ul << 32)) {
            fclose(f);
            err = "read tensor: " + path;
            return NERF_ERR_IO;
        }
        t.data.resize((size_t)sz / 4); // chunks_exact(4)
        if (!t.data.empty() && fread(t.data.data(), 4, t.data.size(), f) != t.data.size()) {
            fclose(f);
            err = "read tensor: " + path;
            return NERF_ERR_IO;
        }
        fclose(f);
        out[name] = std::move(t);
    }
    return NERF_OK;
}

static int take(std::map<std::string, Tensor> &params, const std::string &name, bool bias, int K, int N,
                HostNet::L &L, std::string &err) {
    auto it = params.find(name);
    if (it == params.end()) {
        err = std::string("missing ") + (bias ? "bias" : "matrix") + " parameter: " + name; // src/lib.rs:118,127
        return NERF_ERR_MISSING;
    }
    Tensor &t = it->second;
    char buf[256];
    if (bias) {
        if (t.dims.size() != 1 || (size_t)t.dims[0] != t.data.size() || t.dims[0] != N) {
            snprintf(buf, sizeof buf, "bias dims mismatch for %s: expected [%d], file has %zu values", name.c_str(), N, t.data.size());
            err = buf;
            return NERF_ERR_SHAPE;
        }
        L.b = std::move(t.data);
    } else {
        // compare the dimensions first: their product is only formed for the expected (small) values -- a shapes.txt line
        // "dense0_kernel 99999999999999 99999999999999" overflowed int64 here (found by the host-asan build, tests/test_host_asan.py)
        if (t.dims.size() != 2 || t.dims[0] != K || t.dims[1] != N || (size_t)K * (size_t)N != t.data.size()) {
            snprintf(buf, sizeof buf, "matrix dims mismatch for %s: expected [%d, %d], file has %zu values", name.c_str(), K, N, t.data.size());
            err = buf;
            return NERF_ERR_SHAPE;
        }
        L.K = K; L.N = N;
        L.w = std::move(t.data);
    }
    params.erase(it);
    return NERF_OK;
}

int assemble_net(std::map<std::string, Tensor> &params, HostNet &net, std::string &err) {
    // The fused kernel is specialised for the reference's architecture (lego_rust/*/shapes.txt).
    static const int dK[8] = {63, 256, 256, 256, 256, 319, 256, 256};
    int rc;
    for (int i = 0; i < 8; ++i) {
        const std::string b = "dense" + std::to_string(i);
        if ((rc = take(params, b + "_kernel", false, dK[i], 256, net.dense[i], err))) return rc;
        if ((rc = take(params, b + "_bias", true, dK[i], 256, net.dense[i], err))) return rc;
    }
    if ((rc = take(params, "bottleneck_kernel", false, 256, 256, net.bottleneck, err))) return rc;
    if ((rc = take(params, "bottleneck_bias", true, 256, 256, net.bottleneck, err))) return rc;
    if ((rc = take(params, "viewdirs_kernel", false, 283, 128, net.viewdirs, err))) return rc;
    if ((rc = take(params, "viewdirs_bias", true, 283, 128, net.viewdirs, err))) return rc;
    if ((rc = take(params, "rgb_kernel", false, 128, 3, net.rgb, err))) return rc;
    if ((rc = take(params, "rgb_bias", true, 128, 3, net.rgb, err))) return rc;
    if ((rc = take(params, "alpha_kernel", false, 256, 1, net.alpha, err))) return rc;
    if ((rc = take(params, "alpha_bias", true, 256, 1, net.alpha, err))) return rc;
    return NERF_OK; // leftovers only trip a debug_assert in the reference (src/lib.rs:171)
}

// ------------------------------------------------------------------------------------------------
// Weight packer (layout: mlp_layout.h)
// ------------------------------------------------------------------------------------------------
using namespace nerfmlp;

template <class RowFn>
static void pack_layer(std::vector<float> &s, const HostNet::L &L, int n_steps, int NT, RowFn row) {
    for (int st = 0; st < n_steps; ++st)
        for (int g = 0; g < NT / 4; ++g)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q) {
                    const int r = row(st, l >> 5);
                    const int n = 32 * (4 * g + q) + (l & 31);
                    s.push_back((r >= 0 && r < L.K && n < L.N) ? L.w[(size_t)r * L.N + n] : 0.0f);
                }
}

static int hidden_row(int st, int h) { return 32 * (st >> 4) + regFeature(st & 15, h); }

static void pack_bias(float *dst, const HostNet::L &L, int NT) {
    for (int nt = 0; nt < NT; ++nt)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) dst[(nt * 2 + h) * 16 + r] = L.b[32 * nt + regFeature(r, h)];
}

void pack_network(const HostNet &net, std::vector<float> &ws, std::vector<float> &sm) {
    ws.clear();
    ws.reserve((size_t)kChunksFull * kChunkFloats);
    pack_layer(ws, net.dense[0], kStepsL0, 8, [](int st, int h) { return posSlotFeature(st, h); });
    for (int i = 1; i < 5; ++i) pack_layer(ws, net.dense[i], kStepsHid, 8, hidden_row);
    pack_layer(ws, net.dense[5], kStepsL5, 8, [](int st, int h) { // rows 0..62 encoding, 63..318 h4 (src/network.rs:210)
        return st < 32 ? posSlotFeature(st, h) : 63 + hidden_row(st - 32, h);
    });
    for (int i = 6; i < 8; ++i) pack_layer(ws, net.dense[i], kStepsHid, 8, hidden_row);
    pack_layer(ws, net.bottleneck, kStepsHid, 8, hidden_row);
    pack_layer(ws, net.viewdirs, kStepsView, 4, [](int st, int h) { // rows 0..255 bottleneck, 256..282 dirs (src/network.rs:220)
        if (st < 128) return hidden_row(st, h);
        const int f = dirSlotFeature(st - 128, h);
        return f < 0 ? -1 : 256 + f;
    });
    sm.assign(kSmallFloats, 0.0f);
    for (int i = 0; i < 8; ++i) pack_bias(&sm[kBiasOff + i * 256], net.dense[i], 8);
    pack_bias(&sm[kBiasOff + 8 * 256], net.bottleneck, 8);
    pack_bias(&sm[kBiasViewOff], net.viewdirs, 4);
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 8; ++t)
            for (int r = 0; r < 16; ++r) sm[kAlphaWOff + h * 128 + t * 16 + r] = net.alpha.w[32 * t + regFeature(r, h)];
    for (int h = 0; h < 2; ++h)
        for (int c = 0; c < 3; ++c)
            for (int t = 0; t < 4; ++t)
                for (int r = 0; r < 16; ++r)
                    sm[kRgbWOff + (h * 3 + c) * 64 + t * 16 + r] = net.rgb.w[(size_t)(32 * t + regFeature(r, h)) * 3 + c];
    sm[kMiscOff + 0] = net.alpha.b[0];
    for (int c = 0; c < 3; ++c) sm[kMiscOff + 1 + c] = net.rgb.b[c];
}

// ---- bf16 stream -------------------------------------------------------------------------------
uint16_t f32_to_bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// rowfn(tile, r, h): weight row held by register r (0..15) of input tile `tile` on lane-half h, or -1.
// Emits the layer's 1-KiB pieces (64 lanes x 8 elements) in the order (input tile, k-step, output tile) as f32 values.
template <class RowFn>
static void pack_layer_v1order(std::vector<float> &s, const HostNet::L &L, int n_tiles, int NT, RowFn row) {
    for (int tt = 0; tt < n_tiles; ++tt)
        for (int ks = 0; ks < 2; ++ks)
            for (int nt = 0; nt < NT; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) { // B-fragment element j of k-step ks = register 8 ks + j of the tile
                        const int r = row(tt, 8 * ks + j, l >> 5);
                        const int n = 32 * nt + (l & 31);
                        s.push_back((r >= 0 && r < L.K && n < L.N) ? L.w[(size_t)r * L.N + n] : 0.0f);
                    }
}

void pack_network_v1order_f32(const HostNet &net, std::vector<float> &ws) {
    ws.clear();
    ws.reserve((size_t)kPiecesV1 * 512);
    auto hid = [](int tt, int r, int h) { return 32 * tt + regFeature(r, h); };
    pack_layer_v1order(ws, net.dense[0], 2, 8, [](int tt, int r, int h) { return posSlotFeature(16 * tt + r, h); });
    for (int i = 1; i < 5; ++i) pack_layer_v1order(ws, net.dense[i], 8, 8, hid);
    pack_layer_v1order(ws, net.dense[5], 10, 8, [](int tt, int r, int h) {
        return tt < 2 ? posSlotFeature(16 * tt + r, h) : 63 + 32 * (tt - 2) + regFeature(r, h);
    });
    for (int i = 6; i < 8; ++i) pack_layer_v1order(ws, net.dense[i], 8, 8, hid);
    pack_layer_v1order(ws, net.bottleneck, 8, 8, hid);
    pack_layer_v1order(ws, net.viewdirs, 9, 4, [](int tt, int r, int h) {
        if (tt < 8) return 32 * tt + regFeature(r, h);
        const int f = dirSlotFeature(r, h);
        return f < 0 ? -1 : 256 + f;
    });
}

void bf16_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &ws) {
    ws.resize(v1f.size());
    for (size_t i = 0; i < v1f.size(); ++i) ws[i] = f32_to_bf16_rne(v1f[i]);
    ws.resize((size_t)kChunksFullBf16 * kChunkBytesBf16 / 2, (uint16_t)0); // pad viewdirs to whole chunks
}

void pack_network_bf16(const HostNet &net, std::vector<uint16_t> &ws) {
    std::vector<float> v1f;
    pack_network_v1order_f32(net, v1f);
    bf16_stream_from_v1order(v1f, ws);
}

static float bf16_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// w = w1 + w2 + w3 (+ at most 2^-27 |w|): w1 = bf16(w), w2 = bf16(w - w1), w3 = bf16(w - w1 - w2); the subtractions are exact.
void split_bf16x3(float v, uint16_t out[3]) {
    out[0] = f32_to_bf16_rne(v);
    const float r1 = v - bf16_to_f32(out[0]);
    out[1] = f32_to_bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(out[1]);
    out[2] = f32_to_bf16_rne(r2);
}

// x3 stream: every piece of the v1 order followed by the pieces of the second and third part of the same weights
void x3_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &ws) {
    const size_t n_pieces = v1f.size() / 512;
    ws.assign(n_pieces * 3 * 512, (uint16_t)0);
    for (size_t pc = 0; pc < n_pieces; ++pc)
        for (int e = 0; e < 512; ++e) {
            uint16_t parts[3];
            split_bf16x3(v1f[pc * 512 + e], parts);
            for (int s3 = 0; s3 < 3; ++s3) ws[(pc * 3 + s3) * 512 + e] = parts[s3];
        }
}

// w = w1 + w2 (+ at most 2^-22 |w|): w1 = f16(w), w2 = f16(w - w1), round to nearest even; the subtraction is exact in f32.
// Values beyond the f16 range (|w| > 65504) would become infinite: the loader rejects such networks for this arithmetic.
void split_f16x2(float v, uint16_t out[2]) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    memcpy(&out[0], &h, 2);
    memcpy(&out[1], &l, 2);
}

// f16x2 stream: every piece of the v1 order followed by the piece of the second part of the same weights
// The f16 twin of bf16_stream_from_v1order (mlp_kernel_f16v2.hip: certify_zero's pre-filter): every weight rounded to f16 (RNE).  Returns false
// if a weight is outside the f16 range (the pre-filter then stays bf16 for that network).
bool f16_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &ws) {
    ws.resize(v1f.size());
    bool in_range = true;
    for (size_t i = 0; i < v1f.size(); ++i) {
        const float v = v1f[i];
        if (!(fabsf(v) <= 65504.0f)) in_range = false;
        const _Float16 h = (_Float16)v;
        memcpy(&ws[i], &h, sizeof(uint16_t));
    }
    ws.resize((size_t)kChunksFullBf16 * kChunkBytesBf16 / 2, (uint16_t)0); // pad viewdirs to whole chunks
    return in_range;
}

bool x2_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &ws) {
    const size_t n_pieces = v1f.size() / 512;
    ws.assign(n_pieces * 2 * 512, (uint16_t)0);
    bool in_range = true;
    for (size_t pc = 0; pc < n_pieces; ++pc)
        for (int e = 0; e < 512; ++e) {
            const float v = v1f[pc * 512 + e];
            if (!(fabsf(v) <= 65504.0f)) in_range = false;
            uint16_t parts[2];
            split_f16x2(v, parts);
            ws[(pc * 2 + 0) * 512 + e] = parts[0];
            ws[(pc * 2 + 1) * 512 + e] = parts[1];
        }
    return in_range;
}

// ------------------------------------------------------------------------------------------------
// Camera (reference src/lib.rs:614-645, 213-231; src/vec3.rs:19-34)
// ------------------------------------------------------------------------------------------------
static void normalize3(const float v[3], float o[3]) {
    const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    o[0] = v[0] / len; o[1] = v[1] / len; o[2] = v[2] / len;
}

static void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

void camera_from_values(float near_, float far_, const float origin[3], const float forward[3], const float up[3],
                        const float hwf[3], int width, int height, nerf_camera *c) {
    c->nx = width; c->ny = height;
    c->near_ = near_; c->far_ = far_;
    memcpy(c->pos, origin, sizeof c->pos);
    normalize3(forward, c->dir);
    normalize3(up, c->up);
    const float hw = hwf[1], hh = hwf[0], focal = hwf[2];
    c->alpha_width = atanf((0.5f * hw) / focal);
    c->alpha_height = atanf((0.5f * hh) / focal);
}

void camera_basis(const nerf_camera &cam, float r[3], float u[3], float f[3], float *sx, float *sy) {
    float t[3];
    normalize3(cam.dir, f);
    cross3(f, cam.up, t); normalize3(t, r);
    cross3(r, f, t); normalize3(t, u);
    *sx = tanf(cam.alpha_width);
    *sy = tanf(cam.alpha_height);
}

// ------------------------------------------------------------------------------------------------
// Minimal JSON reader: enough for tf_reference_samples.json (objects, arrays, numbers, strings, literals).
// Only top-level keys are retained; numeric arrays are flattened in document order.
// ------------------------------------------------------------------------------------------------
namespace {
struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool lit(const char *s) { const size_t n = strlen(s); if ((size_t)(end - p) >= n && !strncmp(p, s, n)) { p += n; return true; } return false; }
    bool string(std::string &out) {
        if (p >= end || *p != '"') { err = "expected string"; return false; }
        ++p; out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) { ++p; out.push_back(*p == 'n' ? '\n' : *p == 't' ? '\t' : *p); ++p; }
            else out.push_back(*p++);
        }
        if (p >= end) { err = "unterminated string"; return false; }
        ++p;
        return true;
    }
    // parse any value; numbers encountered (at any depth) are appended to *nums when nums != NULL
    bool value(std::vector<double> *nums, int depth) {
        if (depth > 64) { err = "nesting too deep"; return false; }
        ws();
        if (p >= end) { err = "unexpected end"; return false; }
        if (*p == '{') {
            ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                std::string k; ws();
                if (!string(k)) return false;
                ws(); if (p >= end || *p != ':') { err = "expected ':'"; return false; } ++p;
                if (!value(nums, depth + 1)) return false;
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                err = "expected ',' or '}'"; return false;
            }
        }
        if (*p == '[') {
            ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                if (!value(nums, depth + 1)) return false;
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                err = "expected ',' or ']'"; return false;
            }
        }
        if (*p == '"') { std::string s; return string(s); }
        if (lit("true") || lit("false") || lit("null")) return true;
        char *e = nullptr;
        const double v = strtod(p, &e);
        if (e == p) { err = "unexpected character"; return false; }
        p = e;
        if (nums) nums->push_back(v);
        return true;
    }
};
} // namespace

static int json_top_level_numbers(const std::string &text, std::map<std::string, std::vector<double>> &out, std::string &err) {
    JParser J{text.data(), text.data() + text.size(), {}};
    J.ws();
    if (J.p >= J.end || *J.p != '{') { err = "camera JSON: top level is not an object"; return NERF_ERR_PARSE; }
    ++J.p; J.ws();
    if (J.p < J.end && *J.p == '}') return NERF_OK;
    for (;;) {
        std::string k; J.ws();
        if (!J.string(k)) { err = "camera JSON: " + J.err; return NERF_ERR_PARSE; }
        J.ws(); if (J.p >= J.end || *J.p != ':') { err = "camera JSON: expected ':'"; return NERF_ERR_PARSE; } ++J.p;
        std::vector<double> nums;
        if (!J.value(&nums, 0)) { err = "camera JSON: " + J.err; return NERF_ERR_PARSE; }
        out[k] = std::move(nums);
        J.ws();
        if (J.p < J.end && *J.p == ',') { ++J.p; continue; }
        if (J.p < J.end && *J.p == '}') return NERF_OK;
        err = "camera JSON: expected ',' or '}'"; return NERF_ERR_PARSE;
    }
}

int camera_from_json(const std::string &path, int width, int height, nerf_camera *out, std::string &err) {
    std::ifstream f(path);
    if (!f) { err = "read camera JSON: " + path; return NERF_ERR_IO; }
    std::stringstream ss; ss << f.rdbuf();
    std::map<std::string, std::vector<double>> kv;
    const int rc = json_top_level_numbers(ss.str(), kv, err);
    if (rc) return rc;
    auto need = [&](const char *k, size_t n) -> const std::vector<double> * {
        auto it = kv.find(k);
        if (it == kv.end() || it->second.size() < n) { err = std::string("camera JSON: missing or short key '") + k + "'"; return nullptr; }
        return &it->second;
    };
    const auto *nr = need("near", 1), *fr = need("far", 1), *o = need("camera_origin", 3), *fw = need("camera_forward", 3),
               *up = need("camera_up", 3), *hwf = need("hwf", 3);
    if (!nr || !fr || !o || !fw || !up || !hwf) return NERF_ERR_PARSE;
    // `as_f64().unwrap() as f32` (src/lib.rs:597-599, 620-629)
    const float O[3] = {(float)(*o)[0], (float)(*o)[1], (float)(*o)[2]};
    const float F[3] = {(float)(*fw)[0], (float)(*fw)[1], (float)(*fw)[2]};
    const float U[3] = {(float)(*up)[0], (float)(*up)[1], (float)(*up)[2]};
    const float H[3] = {(float)(*hwf)[0], (float)(*hwf)[1], (float)(*hwf)[2]};
    camera_from_values((float)(*nr)[0], (float)(*fr)[0], O, F, U, H, width, height, out);
    return NERF_OK;
}

// ------------------------------------------------------------------------------------------------
// save_ppm (reference src/lib.rs:567-580)
// ------------------------------------------------------------------------------------------------
void quantize_rgb8(const float *rgb, size_t n_pixels, uint8_t *out) {
    for (size_t i = 0; i < 3 * n_pixels; ++i) {
        float v = rgb[i];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        const float q = v * 255.0f + 0.5f;
        out[i] = (q != q) ? 0 : (uint8_t)q; // NaN `as u8` == 0 in Rust
    }
}

int save_ppm(const std::string &path, int width, int height, const float *rgb, std::string &err) {
    if (width <= 0 || height <= 0) { err = "save_ppm: bad size"; return NERF_ERR_INVALID; }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "save_ppm: cannot create " + path; return NERF_ERR_IO; }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> buf((size_t)width * height * 3);
    quantize_rgb8(rgb, (size_t)width * height, buf.data());
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (!ok) { err = "save_ppm: short write " + path; return NERF_ERR_IO; }
    return NERF_OK;
}

int certify_policy(float margin, uint64_t audited, uint64_t violations, float headroom, float max_error, float *new_margin) {
    if (new_margin) *new_margin = margin;
    if (!audited) return 0; // nothing was certified in front of a predicted cut: nothing to judge
    const float m = margin, err = std::max(max_error, m - headroom);
    float widened = m;
    int rule = 0;
    if (violations) { widened = std::max(4.0f * m, 4.0f * err); rule = 1; }
    else if (!(headroom >= 0.5f * m)) { widened = std::max(2.0f * m, 4.0f * err); rule = 2; }
    else if (!(err <= 0.5f * m)) { widened = std::max(1.25f * m, 3.0f * err); rule = 3; }
    if (rule && new_margin) *new_margin = widened;
    return rule;
}

} // namespace nerfhost
