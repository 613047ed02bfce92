// nerf_api.cpp -- context, device memory, render scheduler and the extern "C" boundary (include/nerf_mi355x.h).
//
// Host counterpart of render_image / render_block (reference src/lib.rs:353-565): where the reference cuts the
// frame in 8x8 blocks for rayon workers, this scheduler cuts it in "passes" of whole ray rows (<= max_rays_per_pass
// rays, all resident in HBM) and runs per pass
//     ray dirs -> stratified t -> coarse MLP (sigma) -> resample/sort -> fine MLP -> composite
// as six launches on one stream with no host round trip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"
#include "host_util.h"
#include "mlp_kernel.h"
#include "mlp_layout.h"
#include "sampling_kernels.h"

using namespace nerfhost;

#include "nerf_internal.h"

using namespace nerfint;

namespace nerfint {

int fail(nerf_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return nerfhost::fail_noctx(code, msg); // also recorded for nerf_last_error(NULL) of this thread
}

} // namespace nerfint

namespace {

hipEvent_t get_event(nerf_ctx *c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Event pairs of kind 1 (dominant MLP kernel) are owned by ctx->dominant once a render has finished; every other
// pair is owned by ctx->last_render.
void recycle_render(nerf_ctx *c) {
    for (auto &p : c->last_render) if (p.kind != 1) { c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b); }
    c->last_render.clear();
}

void recycle_dominant(nerf_ctx *c, size_t keep) {
    while (c->dominant.size() > keep) {
        c->ev_pool.push_back(c->dominant.front().a); c->ev_pool.push_back(c->dominant.front().b);
        c->dominant.erase(c->dominant.begin());
    }
}

} // namespace

int nerfint::ensure_bytes(nerf_ctx *c, void **p, size_t *cur, size_t need) {
    if (*cur >= need) return NERF_OK;
    if (*p) { HIP_TRY(c, hipDeviceSynchronize()); HIP_TRY(c, hipFree(*p)); *p = nullptr; *cur = 0; }
    HIP_TRY(c, hipMalloc(p, need));
    *cur = need;
    return NERF_OK;
}

namespace {

// The coarse network's colours exist only in a coarse_only render (the reference discards them otherwise, src/lib.rs:404):
// their buffer (rays x nc x 3 floats, 492 MB for an 800x800 pass) is allocated on first such use only.
int ensure_workspace(nerf_ctx *c, size_t rays, size_t nc, size_t m, bool need_rgbc) {
    if (rays <= c->ws_rays && nc <= c->ws_nc && m <= c->ws_m && (!need_rgbc || c->d_rgbc)) return NERF_OK;
    rays = std::max(rays, c->ws_rays); nc = std::max(nc, c->ws_nc); m = std::max(m, c->ws_m);
    need_rgbc = need_rgbc || c->d_rgbc != nullptr;
    HIP_TRY(c, hipDeviceSynchronize());
    float **ptrs[] = {&c->d_dirs, &c->d_tc, &c->d_sc, &c->d_rgbc, &c->d_tf, &c->d_sf, &c->d_rgbf};
    for (auto p : ptrs) if (*p) { HIP_TRY(c, hipFree(*p)); *p = nullptr; }
    c->ws_rays = c->ws_nc = c->ws_m = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_dirs, rays * 3 * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_tc, rays * nc * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_sc, rays * nc * sizeof(float)));
    if (need_rgbc) HIP_TRY(c, hipMalloc((void **)&c->d_rgbc, rays * nc * 3 * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_tf, rays * m * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_sf, rays * m * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_rgbf, rays * m * 3 * sizeof(float)));
    c->ws_rays = rays; c->ws_nc = nc; c->ws_m = m;
    return NERF_OK;
}

int upload_packed(nerf_ctx *c, int which, const std::vector<float> &ws, const std::vector<float> &sm);

// The bf16 kernel: mlp_kernel_bf16v2.hip (32x32x16).  A variant build with -DNERF_BF16_V3=1 (make variant; round-3 experiment, see
// DESIGN 4.3) links mlp_kernel_bf16v3.hip (16x16x32) instead: same arithmetic, same results, measured 6 % slower on the full kernel.
#ifndef NERF_BF16_V3
#define NERF_BF16_V3 0
#endif
static constexpr bool g_bf16_v2 = !NERF_BF16_V3;

// weight stream + launcher of the selected arithmetic
static bool valid_dtype(int d) { return d == NERF_MLP_F32 || d == NERF_MLP_BF16 || d == NERF_MLP_BF16X3 || d == NERF_MLP_F16X2; }
static bool split_dtype(int d) { return d == NERF_MLP_BF16X3 || d == NERF_MLP_F16X2; } // f32-accurate operand-splitting arithmetics
static const float *stream_of(const DevNet &n, int dtype) {
    if (dtype == NERF_MLP_BF16X3) return (const float *)n.wstream_x3;
    if (dtype == NERF_MLP_F16X2) return (const float *)n.wstream_x2;
    if (dtype == NERF_MLP_F32) return n.wstream;
    return g_bf16_v2 ? (const float *)n.wstream_bf16v2 : (const float *)n.wstream_bf16v3;
}
// NERF_V2_F16_FULL (experiment, variant builds): NERF_MLP_BF16 renders run the f16 twin of the bf16 kernel (fused ray-mode launches only)
static hipError_t launch_mlp(const nerf_ctx *c, int dtype, const MlpArgs &a, bool full, hipStream_t st) {
#ifdef NERF_V2_F16_FULL
    if (dtype == NERF_MLP_BF16 && a.mode == MLP_MODE_RAYS && !a.raw_pre) {
        MlpArgs b = a;
        for (const DevNet &n : c->net) if ((const void *)n.wstream_bf16v2 == (const void *)a.wstream && n.wstream_f16v2) b.wstream = (const float *)n.wstream_f16v2;
        return full ? nerf_mlp_f16v2_full_launch(b, c->n_cus, st) : nerf_mlp_f16v2_launch(b, c->n_cus, st);
    }
#endif
    if (dtype == NERF_MLP_F32) return nerf_mlp_launch(a, full, c->n_cus, st);
    if (dtype == NERF_MLP_BF16X3) return nerf_mlp_bf16x3_launch(a, full, c->n_cus, st);
    if (dtype == NERF_MLP_F16X2) return nerf_mlp_f16x2_launch(a, full, c->n_cus, st);
#if NERF_BF16_V3
    return nerf_mlp_bf16v3_launch(a, full, c->n_cus, st);
#else
    return nerf_mlp_bf16v2_launch(a, full, c->n_cus, st);
#endif
}

// v1 stream -> v2 stream: the 1-KiB pieces are identical (lane l, element j = W[row(tile, 8 ks + j, l >> 5)][32 nt + (l & 31)]);
// v1 orders a layer's pieces (k-step, nt), v2 orders them (nt, k-step) and pads viewdirs' 72 pieces to 80.
static void bf16_v2_from_v1(const std::vector<uint16_t> &v1, std::vector<uint16_t> &v2) {
    using namespace nerfmlp;
    v2.clear();
    v2.reserve((size_t)kChunksFullBf16V2 * kChunkBytesBf16V2 / 2);
    size_t base = 0; // uint16 elements
    auto layer = [&](int KS, int NT) {
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks) {
                const uint16_t *src = &v1[base + ((size_t)ks * NT + nt) * 512];
                v2.insert(v2.end(), src, src + 512);
            }
        base += (size_t)KS * NT * 512;
    };
    layer(4, 8);
    for (int i = 0; i < 4; ++i) layer(16, 8);
    layer(20, 8);
    for (int i = 0; i < 3; ++i) layer(16, 8);
    layer(18, 4);
    v2.resize((size_t)kChunksFullBf16V2 * kChunkBytesBf16V2 / 2, (uint16_t)0);
}

// v1 order (f32 values) -> the stream of mlp_kernel_bf16v3.hip (v_mfma_f32_16x16x32_bf16): per layer, per output tile nt, per k-step ks
// (K = 32), per 16-feature half fh one 1-KiB piece: lane l = (i = l & 15, g = l >> 4), element e = W[row(ks, g, e)][32 nt + 16 fh + i].
// Hidden k-steps: k-slot (g, e) = feature 4 g + e (e < 4) or 16 + 4 g + e - 4 of input tile ks (the C/D layout of two 16x16 output
// halves packed pairwise); encoding k-steps: slot 32 ks + 8 g + e = the reference's feature index (src/network.rs:263-330).
static void bf16_v3_from_v1(const std::vector<float> &v1f, std::vector<uint16_t> &v3) {
    using namespace nerfmlp;
    v3.clear();
    v3.reserve((size_t)kChunksFullBf16V2 * kChunkBytesBf16V2 / 2);
    int pos_inv[64][3], dir_inv[32][3]; // reference feature -> (input tile, register, lane-half) of the 32x32 layouts
    for (auto &x : pos_inv) x[0] = -1;
    for (auto &x : dir_inv) x[0] = -1;
    for (int tt = 0; tt < 2; ++tt)
        for (int r = 0; r < 16; ++r)
            for (int h = 0; h < 2; ++h) {
                const int f = posSlotFeature(16 * tt + r, h);
                if (f >= 0) { pos_inv[f][0] = tt; pos_inv[f][1] = r; pos_inv[f][2] = h; }
            }
    for (int r = 0; r < 16; ++r)
        for (int h = 0; h < 2; ++h) {
            const int f = dirSlotFeature(r, h);
            if (f >= 0) { dir_inv[f][0] = 0; dir_inv[f][1] = r; dir_inv[f][2] = h; }
        }
    size_t base = 0; // v1 pieces
    // kind of k-step ks: 0 = activation tile `tile`, 1 = position-encoding slots 32 * tile .., 2 = direction-encoding slots
    auto layer = [&](int n_tiles_v1, int NT, int KS, auto kind_of) {
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks)
                for (int fh = 0; fh < 2; ++fh)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int i = l & 15, g = l >> 4;
                            int kind, tile;
                            kind_of(ks, kind, tile);
                            int tt = -1, r = 0, h = 0;
                            if (kind == 0) {
                                const int f = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
                                tt = tile; h = (f >> 2) & 1; r = (f & 3) + 4 * (f >> 3);
                            } else if (kind == 1) {
                                const int s = 32 * tile + 8 * g + e;
                                if (s < 63 && pos_inv[s][0] >= 0) { tt = pos_inv[s][0]; r = pos_inv[s][1]; h = pos_inv[s][2]; }
                            } else {
                                const int s = 8 * g + e;
                                if (s < 27 && dir_inv[s][0] >= 0) { tt = n_tiles_v1 - 1; r = dir_inv[s][1]; h = dir_inv[s][2]; }
                            }
                            float v = 0.0f;
                            if (tt >= 0) {
                                const size_t piece = base + ((size_t)(tt * 2 + (r >> 3)) * NT + nt);
                                v = v1f[(piece * 64 + (size_t)(16 * fh + i + 32 * h)) * 8 + (r & 7)];
                            }
                            v3.push_back(f32_to_bf16_rne(v));
                        }
        base += (size_t)n_tiles_v1 * 2 * NT;
    };
    layer(2, 8, 2, [](int ks, int &kind, int &tile) { kind = 1; tile = ks; });
    for (int i = 0; i < 4; ++i) layer(8, 8, 8, [](int ks, int &kind, int &tile) { kind = 0; tile = ks; });
    layer(10, 8, 10, [](int ks, int &kind, int &tile) { if (ks < 2) { kind = 1; tile = ks; } else { kind = 0; tile = ks; } });
    for (int i = 0; i < 3; ++i) layer(8, 8, 8, [](int ks, int &kind, int &tile) { kind = 0; tile = ks; });
    layer(9, 4, 9, [](int ks, int &kind, int &tile) { if (ks < 8) { kind = 0; tile = ks; } else { kind = 2; tile = 0; } });
    v3.resize((size_t)kChunksFullBf16V2 * kChunkBytesBf16V2 / 2, (uint16_t)0);
}

// All bf16-family streams come from one f32 array in the first bf16 design's piece order (host_util.h).
int upload_bf16_family(nerf_ctx *c, int which, const std::vector<float> &v1f) {
    if (v1f.size() != (size_t)nerfmlp::kPiecesV1 * 512) return fail(c, NERF_ERR_SHAPE, "internal: bf16 piece array has the wrong size");
    DevNet &d = c->net[which];
    std::vector<uint16_t> wb, v2, x3;
    bf16_stream_from_v1order(v1f, wb);
    bf16_v2_from_v1(wb, v2);
    x3_stream_from_v1order(v1f, x3);
    if (x3.size() != (size_t)nerfmlp::kChunksFullX3 * nerfmlp::kChunkBytesX3 / 2) return fail(c, NERF_ERR_SHAPE, "internal: bf16x3 stream size");
    if (!d.wstream_bf16v2) HIP_TRY(c, hipMalloc((void **)&d.wstream_bf16v2, v2.size() * sizeof(uint16_t)));
    HIP_TRY(c, hipMemcpy(d.wstream_bf16v2, v2.data(), v2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (!g_bf16_v2) {
        std::vector<uint16_t> v3;
        bf16_v3_from_v1(v1f, v3);
        if (v3.size() != v2.size()) return fail(c, NERF_ERR_SHAPE, "internal: bf16 v3 stream size");
        if (!d.wstream_bf16v3) HIP_TRY(c, hipMalloc((void **)&d.wstream_bf16v3, v3.size() * sizeof(uint16_t)));
        HIP_TRY(c, hipMemcpy(d.wstream_bf16v3, v3.data(), v3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (!d.wstream_x3) HIP_TRY(c, hipMalloc((void **)&d.wstream_x3, x3.size() * sizeof(uint16_t)));
    HIP_TRY(c, hipMemcpy(d.wstream_x3, x3.data(), x3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    {   // the f16 twin of the bf16 v2 stream (certify_zero's pre-filter): the same pieces in the same order, f16-rounded
        std::vector<uint16_t> hb, h2;
        const bool fits = f16_stream_from_v1order(v1f, hb);
        if (fits) {
            bf16_v2_from_v1(hb, h2); // a permutation of 16-bit elements: the element type does not matter
            if (h2.size() != v2.size()) return fail(c, NERF_ERR_SHAPE, "internal: f16 v2 stream size");
            if (!d.wstream_f16v2) HIP_TRY(c, hipMalloc((void **)&d.wstream_f16v2, h2.size() * sizeof(uint16_t)));
            HIP_TRY(c, hipMemcpy(d.wstream_f16v2, h2.data(), h2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        } else if (d.wstream_f16v2) {
            HIP_TRY(c, hipFree(d.wstream_f16v2));
            d.wstream_f16v2 = nullptr;
        }
        // certify_zero calibrates itself per network: margins start at the floors of the pre-filter this network gets
        c->cert_prefilter_f16[which] = g_bf16_v2 && c->cert_allow_f16 && d.wstream_f16v2 != nullptr;
        c->cert_margin[which] = c->cert_prefilter_f16[which] ? c->cert_margin_floor_f16[which] : c->cert_margin_floor[which];
    }
    std::vector<uint16_t> x2;
    if (x2_stream_from_v1order(v1f, x2)) { // every weight inside the f16 range (the lego networks: |w| <= 8.3)
        if (x2.size() != (size_t)nerfmlp::kChunksFullF16X2 * nerfmlp::kChunkBytesF16X2 / 2) return fail(c, NERF_ERR_SHAPE, "internal: f16x2 stream size");
        if (!d.wstream_x2) HIP_TRY(c, hipMalloc((void **)&d.wstream_x2, x2.size() * sizeof(uint16_t)));
        HIP_TRY(c, hipMemcpy(d.wstream_x2, x2.data(), x2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    } else if (d.wstream_x2) { // a reload with out-of-range weights: NERF_MLP_F16X2 becomes unavailable for this network
        HIP_TRY(c, hipFree(d.wstream_x2));
        d.wstream_x2 = nullptr;
    }
    return NERF_OK;
}

int upload_net(nerf_ctx *c, int which, const HostNet &hn) {
    std::vector<float> ws, sm;
    pack_network(hn, ws, sm);
    int rc = upload_packed(c, which, ws, sm);
    if (rc) return rc;
    std::vector<float> v1f;
    pack_network_v1order_f32(hn, v1f);
    return upload_bf16_family(c, which, v1f);
}

// The bf16-family streams hold the same weights in another order; for networks that arrive as a packed f32 blob they are
// rebuilt from the f32 stream's pieces (piece (s, g): lane l, q -> W[row(s, l >> 5)][32 (4 g + q) + (l & 31)]).
int bf16_from_f32_stream(nerf_ctx *c, int which, const std::vector<float> &ws) {
    using namespace nerfmlp;
    std::vector<float> v1f;
    v1f.reserve((size_t)kPiecesV1 * 512);
    size_t layer_base = 0; // floats
    auto layer = [&](int n_tiles, int NT) {
        for (int tt = 0; tt < n_tiles; ++tt)
            for (int ks = 0; ks < 2; ++ks)
                for (int nt = 0; nt < NT; ++nt)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int st = 16 * tt + 8 * ks + j; // f32 k-step that holds register 8 ks + j of tile tt
                            const size_t piece = layer_base + ((size_t)st * (NT / 4) + nt / 4) * 256;
                            v1f.push_back(ws[piece + (size_t)l * 4 + (nt & 3)]);
                        }
        layer_base += (size_t)n_tiles * 16 * NT * 64;
    };
    layer(2, 8);
    for (int i = 0; i < 4; ++i) layer(8, 8);
    layer(10, 8);
    for (int i = 0; i < 3; ++i) layer(8, 8);
    layer(9, 4);
    return upload_bf16_family(c, which, v1f);
}

int upload_packed(nerf_ctx *c, int which, const std::vector<float> &ws, const std::vector<float> &sm) {
    if (ws.size() != (size_t)nerfmlp::kChunksFull * nerfmlp::kChunkFloats || sm.size() != (size_t)nerfmlp::kSmallFloats)
        return fail(c, NERF_ERR_SHAPE, "packed network image has the wrong size for this library build");
    DevNet &d = c->net[which];
    if (!d.wstream) HIP_TRY(c, hipMalloc((void **)&d.wstream, ws.size() * sizeof(float)));
    if (!d.small) HIP_TRY(c, hipMalloc((void **)&d.small, sm.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(d.wstream, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d.small, sm.data(), sm.size() * sizeof(float), hipMemcpyHostToDevice));
    d.loaded = true;
    c->cert_margin[which] = c->cert_margin_floor[which]; // certify_zero calibrates itself per network (upload_bf16_family, which follows, decides the pre-filter)
    return NERF_OK;
}

RayGenArgs make_raygen(const nerf_camera &cam, int s) {
    RayGenArgs g{};
    g.rnx = cam.nx * s; g.rny = cam.ny * s;
    g.half = 0.5f; g.normalize = 1;
    camera_basis(cam, g.r, g.u, g.f, &g.sx, &g.sy);
    return g;
}

int check_camera(nerf_ctx *c, const nerf_camera *cam) {
    if (!cam) return fail(c, NERF_ERR_INVALID, "camera is NULL");
    if (cam->nx <= 0 || cam->ny <= 0) return fail(c, NERF_ERR_INVALID, "width and height must be greater than zero"); // src/lib.rs:705-709
    return NERF_OK;
}

// Record kernel(s) between two events of the given kind.  If done() is never reached (a launch failed and the caller
// returned early) the destructor hands the events back to the pool.
struct Timed {
    nerf_ctx *c; hipStream_t st; EvPair p; bool on;
    Timed(nerf_ctx *c_, hipStream_t st_, int kind, uint64_t points, bool enable) : c(c_), st(st_), on(enable) {
        p.kind = kind; p.points = points; p.a = get_event(c); p.b = get_event(c);
        if (!p.a || !p.b) on = false;
        if (on) (void)hipEventRecord(p.a, st);
    }
    Timed(const Timed &) = delete;
    Timed &operator=(const Timed &) = delete;
    void done(std::vector<EvPair> &dst) {
        if (on) { (void)hipEventRecord(p.b, st); dst.push_back(p); p.a = p.b = nullptr; }
    }
    ~Timed() {
        if (p.a) c->ev_pool.push_back(p.a);
        if (p.b) c->ev_pool.push_back(p.b);
    }
};

// What the audit of a certify_zero frame found (per network: 0 coarse, 1 fine), read from the device after the frame.
struct CertOutcome {
    uint64_t audited[2] = {0, 0}, violations[2] = {0, 0};
    float headroom[2] = {INFINITY, INFINITY}; // min over the audited certificates of -(exact pre-activation)
    float max_err[2] = {0.0f, 0.0f};          // max over them of |bf16 - exact pre-activation|
    uint64_t listed[2] = {0, 0};      // samples the exact kernel evaluated (both launches, audited certificates included)
    uint64_t fallback_rays = 0;       // rays whose predicted cut the exact transmittance did not confirm
    uint64_t range_left[2] = {0, 0};  // f16 pre-filter: wave tiles in which an activation left the f16 range (the frame's certificates are void)
    uint64_t list_need = 0;           // largest list any pass wanted ...
    uint64_t list_capacity = 0;       // ... and what it had
    uint64_t pass_samples = 0;        // samples of the largest network pass (the list's worst case)
    bool used[2] = {false, false};
};

int render_once(nerf_ctx *c, const nerf_camera *cam, const nerf_render_opts *o, float *d_out, hipStream_t st, nerf_stats *stats,
                CertOutcome *cert) {
    int rc;
    if ((rc = check_camera(c, cam))) return rc;
    if (!o) return fail(c, NERF_ERR_INVALID, "opts is NULL");
    if (!d_out) return fail(c, NERF_ERR_INVALID, "output pointer is NULL");
    if (o->n_coarse <= 0) return fail(c, NERF_ERR_INVALID, "coarse samples per ray must be greater than 0"); // src/lib.rs:483-486
    if (o->n_fine < 0) return fail(c, NERF_ERR_INVALID, "fine samples per ray must be >= 0");
    if (!valid_dtype(o->mlp_dtype)) return fail(c, NERF_ERR_INVALID, "mlp_dtype must be NERF_MLP_F32, NERF_MLP_BF16, NERF_MLP_BF16X3 or NERF_MLP_F16X2");
    const int dtype = o->mlp_dtype;
    // bf16x3 / f16x2: the sampling pass (coarse sigma -> CDF -> fine sample positions) stays on the exact-f32 MFMA kernel, so the fine
    // samples sit where the f32 path puts them bit for bit (a 1e-5 density difference can move a CDF entry across a fixed
    // uniform draw and relocate a sample -- a discontinuity, not an accuracy problem); only the colour-producing pass runs in
    // the three-way split arithmetic.
    // hybrid_sampling: the sampling pass itself runs in a split arithmetic (the render's own, or -- for an f32 render -- f16x2 where
    // the network fits the f16 range, else bf16x3) and only the ill-conditioned rays are redone in f32.
    const int dtype_coarse = o->hybrid_sampling ? (split_dtype(dtype) ? dtype : c->net[NERF_NET_COARSE].wstream_x2 ? NERF_MLP_F16X2 : NERF_MLP_BF16X3)
                                                : (split_dtype(dtype) && !o->coarse_only) ? NERF_MLP_F32 : dtype;
    if (o->skip_empty != 0 && o->skip_empty != 1) return fail(c, NERF_ERR_INVALID, "skip_empty must be 0 or 1");
    if (o->skip_dead != 0 && o->skip_dead != 1) return fail(c, NERF_ERR_INVALID, "skip_dead must be 0 or 1");
    const bool seq = o->skip_dead != 0;
    if (seq && dtype == NERF_MLP_BF16 && !g_bf16_v2) return fail(c, NERF_ERR_INVALID, "skip_dead with NERF_MLP_BF16 is not part of the NERF_BF16_V3 variant build");
    if (o->hybrid_sampling != 0 && o->hybrid_sampling != 1) return fail(c, NERF_ERR_INVALID, "hybrid_sampling must be 0 or 1");
    if (o->certify_zero != 0 && o->certify_zero != 1) return fail(c, NERF_ERR_INVALID, "certify_zero must be 0 or 1");
    if (o->certify_zero && (dtype == NERF_MLP_BF16 || seq || o->skip_empty || !g_bf16_v2))
        return fail(c, NERF_ERR_INVALID, "certify_zero needs mlp_dtype F32, BF16X3 or F16X2 and neither skip_empty nor skip_dead");
    if (o->hybrid_sampling && !(seq && !o->coarse_only && dtype != NERF_MLP_BF16)) // bf16 is its own arithmetic: there are no f32 sample positions to protect
        return fail(c, NERF_ERR_INVALID, "hybrid_sampling needs skip_dead = 1, mlp_dtype F32, BF16X3 or F16X2, and a hierarchical render");
    if (!c->net[NERF_NET_COARSE].loaded) return fail(c, NERF_ERR_STATE, "coarse network not loaded");
    if (!o->coarse_only && !c->net[NERF_NET_FINE].loaded) return fail(c, NERF_ERR_STATE, "fine network not loaded");
    if (dtype == NERF_MLP_F16X2 && (!c->net[o->coarse_only ? NERF_NET_COARSE : NERF_NET_FINE].wstream_x2 ||
                                    (o->hybrid_sampling && !c->net[NERF_NET_COARSE].wstream_x2)))
        return fail(c, NERF_ERR_STATE, "NERF_MLP_F16X2 is unavailable for this network: a weight exceeds the f16 range");
    const int s = o->ssaa > 1 ? o->ssaa : 1;
    int x0 = 0, y0 = 0, cw = cam->nx, ch = cam->ny;
    if (o->crop_w > 0 || o->crop_h > 0) { x0 = o->crop_x0; y0 = o->crop_y0; cw = o->crop_w; ch = o->crop_h; }
    if (cw <= 0 || ch <= 0 || x0 < 0 || y0 < 0 || x0 + cw > cam->nx || y0 + ch > cam->ny)
        return fail(c, NERF_ERR_INVALID, "crop window outside the frame");
    // one band of the window's rows (multi-GPU): contiguous bands are just a shorter window; striped bands keep the window and map
    // band-local rows to window rows in the ray generator (sampling_kernels.hip ray_row)
    int stripe = 0;
    if (o->band_count > 1) {
        if (o->band_index < 0 || o->band_index >= o->band_count || o->band_stripe_rows < 0) return fail(c, NERF_ERR_INVALID, "band_index / band_count / band_stripe_rows out of range");
        const int rows = band_rows(ch, o->band_index, o->band_count, o->band_stripe_rows);
        if (rows <= 0) return fail(c, NERF_ERR_INVALID, "this band has no rows (more bands than rows)");
        if (o->band_stripe_rows == 0) y0 += band_first_row(ch, o->band_index, o->band_count);
        else stripe = o->band_stripe_rows;
        ch = rows;
    } else if (o->band_count < 0) return fail(c, NERF_ERR_INVALID, "band_count must be >= 0");
    const int nc = o->n_coarse;
    // sample_importance returns nothing for count == 0 or < 3 coarse samples (src/lib.rs:295-297): the fine net
    // then runs on the coarse samples only
    const int nf = (o->coarse_only || o->n_fine == 0 || nc < 3) ? 0 : o->n_fine;
    const int M = nc + nf;
    const bool hybrid = o->hybrid_sampling != 0 && nf > 0; // without resampling there is nothing to protect
    if (composite_lds_bytes(M) > 160 * 1024 || (nf > 0 && resample_lds_bytes(nc, nf) > 160 * 1024))
        return fail(c, NERF_ERR_INVALID, "too many samples per ray for the sampling / compositing kernels (one ray per wave in LDS)");
    const int RW = cw * s, RH = ch * s, RX0 = x0 * s, RY0 = y0 * s;
    // a pass must keep rays * samples within int32 (kernel indices) as well as within the configured budget
    size_t pass_cap = std::min<size_t>(c->max_rays_per_pass, (size_t)0x3fffffff / (size_t)M);
    // skip_dead in a split arithmetic or in bf16 (two launches): the compacted trunk outputs are sized for the worst case (every sample
    // of a pass live; 1 KiB each as f32 tiles, 512 B as packed bf16).  The f32 kernel compacts in LDS and runs its colour passes in the
    // same launch: no buffer, no extra passes.
    const bool h8_export = seq && (split_dtype(dtype) || dtype == NERF_MLP_BF16);
    const size_t h8_sample_bytes = dtype == NERF_MLP_BF16 ? 512 : 1024;
    // Budget and allocation of that buffer are ONE critical section per process: contexts that share a device (nerf_render_image_multi's
    // worker threads) would otherwise each claim half of the same free memory and fail in hipMalloc.  A context keeps its buffer while
    // later renders fit; a render that needs less than a quarter of it gives it back (shrinks).
    static std::mutex h8_mu;
    std::unique_lock<std::mutex> h8_lock(h8_mu, std::defer_lock);
    if (h8_export) {
        h8_lock.lock();
        size_t budget = c->max_export_bytes, free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) // never more than half of what the device can still give (plus what we already hold)
            budget = std::min(budget, (free_b + c->h8_bytes) / 2);
        const size_t row_bytes = (size_t)RW * M * h8_sample_bytes;
        if (row_bytes > budget) {
            char msg[256];
            snprintf(msg, sizeof msg, "skip_dead in this arithmetic exports %zu B per sample: one ray row of this render (%d rays x %d samples) needs "
                     "%zu MiB, the budget is %zu MiB (NERF_MAX_EXPORT_BYTES, free device memory / 2); render a narrower crop or use NERF_MLP_F32",
                     h8_sample_bytes, RW, M, row_bytes >> 20, budget >> 20);
            return fail(c, NERF_ERR_INVALID, msg);
        }
        pass_cap = std::min<size_t>(pass_cap, budget / ((size_t)M * h8_sample_bytes));
    }
    if ((size_t)RW > pass_cap && (size_t)RW * M > (size_t)0x3fffffff) return fail(c, NERF_ERR_INVALID, "ray row too wide for one pass");
    const size_t rows_per_pass = std::max<size_t>(1, std::min<size_t>(RH, pass_cap / (size_t)RW));
    if ((rc = ensure_workspace(c, rows_per_pass * RW, nc, M, o->coarse_only != 0))) return rc;
    float *ray_out = d_out;
    if (s > 1) {
        if ((rc = ensure_bytes(c, (void **)&c->d_rayfb, &c->rayfb_floats, (size_t)RW * RH * 3 * sizeof(float)))) return rc;
        ray_out = c->d_rayfb;
    }
    const uint32_t n_passes_total = (uint32_t)((RH + rows_per_pass - 1) / rows_per_pass);
    if (seq) { // per MLP launch: {u32 ray queue head, u32 live-sample count, u64 evaluated samples}
        const size_t slots = (size_t)n_passes_total * 3; // per pass: coarse, fine, hybrid f32 redo of the coarse pass
        if (slots > c->seq_slots) {
            size_t bytes = c->seq_slots * 16;
            if ((rc = ensure_bytes(c, (void **)&c->d_seq, &bytes, slots * 16))) return rc;
            c->seq_slots = slots;
        }
        HIP_TRY(c, hipMemsetAsync(c->d_seq, 0, slots * 16, st));
        const size_t pass_samples = rows_per_pass * RW * (size_t)M;
        if (h8_export) {
            const size_t need = dtype == NERF_MLP_BF16 ? nerf_seq_h8_bytes_bf16(pass_samples) : nerf_seq_h8_bytes(pass_samples);
            if (c->d_h8 && need < c->h8_bytes / 4) { // a much smaller render: give the big buffer back
                HIP_TRY(c, hipDeviceSynchronize());
                HIP_TRY(c, hipFree(c->d_h8));
                c->d_h8 = nullptr; c->h8_bytes = 0;
            }
            if ((rc = ensure_bytes(c, (void **)&c->d_h8, &c->h8_bytes, need))) return rc;
            if ((rc = ensure_bytes(c, (void **)&c->d_slot_point, &c->slot_point_bytes, (pass_samples + 256) * sizeof(unsigned int)))) return rc;
            h8_lock.unlock();
        }
        if (hybrid && (rc = ensure_bytes(c, (void **)&c->d_flag_list, &c->flag_list_bytes, rows_per_pass * RW * sizeof(unsigned int)))) return rc;
    }
    // Zero certification (nerf_render_opts.certify_zero; DESIGN 4.9, protocol: sampling_kernels.hip k_cert_*): a 16-bit pass (f16 or bf16 operands, cert_prefilter_f16) over all samples
    // finds (Z) the samples whose density pre-activation is so far below 0 (margin per network, c->cert_margin: audited every frame and
    // widened by render_device when the audit's headroom shrinks) that the exact network's density is certainly 0 there too, and predicts
    // (C) where each ray's transmittance falls below the reference's 1e-4 cut; the exact kernel evaluates only the remaining samples in
    // front of the predicted cut (a device-side list), the exact transmittance confirms the cut (or a second launch evaluates the rest).
    // A certified sample has sigma = 0 => weight 0, a sample behind the cut has weight 0 whatever its density: the frame is the plain
    // frame, bit for bit, as long as no certificate is wrong.
    const bool certify = o->certify_zero != 0;
    // per (pass, network): {list 1 length, list 2 length, audited, violations, headroom code, max-error bits, fallback rays, audit-record count,
    //                       pre-filter ray queue head, list 1 back-part length, u64 samples the pre-filter evaluated, ...}
    constexpr int kCertSlots = 16;
    size_t aux_cap = 0;
    size_t cert_cap = 0;
    if (certify) {
        if (cert_plan_lds_bytes(M, nullptr) > 159 * 1024) return fail(c, NERF_ERR_INVALID, "certify_zero: too many samples per ray");
        const size_t n_pass = ((size_t)RH + rows_per_pass - 1) / rows_per_pass;
        const size_t pass_samples = rows_per_pass * RW * (size_t)M;
        // the list is sized from what earlier frames needed (c->cert_list_frac of the samples, +25 %): if a pass wants more, the entries
        // beyond the capacity are counted, not stored, and render_device renders the frame again with a list that fits
        // (up to 64 MiB the list simply holds every sample: small renders -- a central crop lists 70 % of its samples -- never take that path)
        cert_cap = pass_samples <= ((size_t)16 << 20) ? pass_samples : std::min<size_t>(pass_samples, (size_t)((double)pass_samples * c->cert_list_frac) + 4096);
        if ((rc = ensure_bytes(c, (void **)&c->d_point_list, &c->point_list_bytes, cert_cap * sizeof(unsigned int)))) return rc;
        cert_cap = std::min<size_t>(pass_samples, c->point_list_bytes / sizeof(unsigned int)); // an earlier, larger allocation is kept
        if ((rc = ensure_bytes(c, (void **)&c->d_jstar, &c->jstar_bytes, rows_per_pass * RW * sizeof(int)))) return rc;
        aux_cap = 2 * (pass_samples / ((size_t)std::min(c->cert_audit_mask, c->cert_audit_mask_near) + 1)) + 4096; // more than the audit can select: a certificate that does not fit is not audited
        if ((rc = ensure_bytes(c, (void **)&c->d_cert_aux, &c->cert_aux_bytes, aux_cap * 2 * sizeof(unsigned int)))) return rc;
        if ((rc = ensure_bytes(c, (void **)&c->d_cert, &c->cert_bytes, kCertSlots * 2 * n_pass * sizeof(unsigned int)))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_cert, 0, kCertSlots * 2 * n_pass * sizeof(unsigned int), st));
        if (cert) { cert->list_capacity = cert_cap; cert->pass_samples = pass_samples; }
    }
    recycle_render(c);
    recycle_dominant(c, 4096); // bound the backlog if the caller never queries
    // If this render returns early (a launch failed), its dominant-kernel event pairs never reach c->dominant: hand them back
    // to last_render's ownership so that the next recycle_render returns them to the pool.
    struct RenderScope {
        nerf_ctx *c; bool ok = false;
        ~RenderScope() { if (!ok) for (auto &p : c->last_render) if (p.kind == 1) p.kind = 2; }
    } scope{c};
    if ((o->skip_empty || certify) && c->d_skip) HIP_TRY(c, hipMemsetAsync(c->d_skip, 0, sizeof(unsigned long long), st));
    const bool watch_range = c->d_nonfinite && (split_dtype(dtype) || split_dtype(dtype_coarse));
    if (watch_range) HIP_TRY(c, hipMemsetAsync(c->d_nonfinite, 0, sizeof(unsigned int), st));
    const bool timing = true;
    RayGenArgs g = make_raygen(*cam, s);
    if (stripe > 0) { g.stripe = stripe * s; g.stripe_n = o->band_count; g.stripe_i = o->band_index; g.stripe_y0 = RY0; }
    const DevNet &NC = c->net[NERF_NET_COARSE], &NF = c->net[NERF_NET_FINE];
    uint32_t passes = 0;
    for (int row = 0; row < RH; row += (int)rows_per_pass, ++passes) {
        const int rows = std::min<int>((int)rows_per_pass, RH - row);
        const int n_rays = rows * RW;
        g.n_rays = n_rays; g.rx0 = RX0; g.ry0 = (stripe > 0 ? 0 : RY0) + row; g.rw = RW;
        {
            Timed t(c, st, 2, 0, timing);
            HIP_TRY(c, launch_ray_dirs(g, c->d_dirs, st));
            HIP_TRY(c, launch_stratified(g, nc, cam->near_, cam->far_, o->seed, c->d_tc, st));
            t.done(c->last_render);
        }
        // skip_dead: one network over the rays of this pass as ray-sequential trunk [+ colour head on the live samples]
        auto seq_pass = [&](const DevNet &net, int dt, int spr, const float *t_in, float *sigma_out, float *rgb_out, int slot, int kind_trunk) -> int {
            unsigned int *ctr = c->d_seq + 4 * (size_t)slot;
            HIP_TRY(c, hipMemsetAsync(sigma_out, 0, (size_t)n_rays * spr * sizeof(float), st)); // samples behind the cut stay 0
            if (rgb_out) HIP_TRY(c, hipMemsetAsync(rgb_out, 0, (size_t)n_rays * spr * 3 * sizeof(float), st)); // weight-0 samples: 0 * 0
            SeqArgs q{};
            q.wstream = stream_of(net, dt); q.small_params = net.small; q.n_rays = n_rays; q.samples_per_ray = spr;
            q.ray_dirs = c->d_dirs; q.t = t_in; q.far_ = cam->far_;
            q.origin[0] = cam->pos[0]; q.origin[1] = cam->pos[1]; q.origin[2] = cam->pos[2];
            q.sigma_out = sigma_out; q.ray_counter = ctr; q.live_count = ctr + 1; q.h8 = c->d_h8; q.slot_point = c->d_slot_point;
            q.rgb_out = rgb_out; // f32: colour passes inside the trunk launch
            q.stats = (unsigned long long *)(ctr + 2);
            q.nonfinite = split_dtype(dt) ? c->d_nonfinite : nullptr;
            {
                Timed t(c, st, kind_trunk, (uint64_t)n_rays * spr, timing);
                HIP_TRY(c, dt == NERF_MLP_BF16X3 ? nerf_trunk_seq_x3_launch(q, rgb_out != nullptr, c->n_cus, st)
                           : dt == NERF_MLP_F16X2 ? nerf_trunk_seq_f16x2_launch(q, rgb_out != nullptr, c->n_cus, st)
                           : dt == NERF_MLP_BF16  ? nerf_trunk_seq_bf16_launch(q, rgb_out != nullptr, c->n_cus, st)
                                                  : nerf_trunk_seq_launch(q, rgb_out != nullptr, c->n_cus, st));
                t.done(c->last_render);
            }
            if (rgb_out && dt != NERF_MLP_F32) {
                ColourArgs k{};
                k.wstream = stream_of(net, dt); k.small_params = net.small; k.live_count = ctr + 1; k.h8 = c->d_h8; k.slot_point = c->d_slot_point;
                k.ray_dirs = c->d_dirs; k.samples_per_ray = spr; k.rgb_out = rgb_out; k.nonfinite = split_dtype(dt) ? c->d_nonfinite : nullptr;
                Timed t(c, st, 4, 0, timing);
                HIP_TRY(c, dt == NERF_MLP_BF16X3 ? nerf_colour_x3_launch(k, c->n_cus, st)
                           : dt == NERF_MLP_F16X2 ? nerf_colour_f16x2_launch(k, c->n_cus, st)
                                                  : nerf_colour_bf16_launch(k, c->n_cus, st));
                t.done(c->last_render);
            }
            return NERF_OK;
        };
        MlpArgs a{};
        a.mode = MLP_MODE_RAYS;
        a.ray_dirs = c->d_dirs;
        a.origin[0] = cam->pos[0]; a.origin[1] = cam->pos[1]; a.origin[2] = cam->pos[2];
        // zero certification: 16-bit pre-activations of all samples -> plan (list 1, predicted cuts) -> exact kernel on list 1 -> audit ->
        // exact transmittance confirms the cuts (list 2 = what is left of the rays it does not) -> exact kernel on list 2
        auto cert_pass = [&](const DevNet &net, int which, int dt, int spr, const float *t_in, float *sigma_out, float *rgb_out, int kind) -> int {
            const int n_pts = n_rays * spr;
            unsigned int *slots = c->d_cert + kCertSlots * (2 * (size_t)passes + which);
            const unsigned cap = (unsigned)std::min<size_t>(cert_cap, (size_t)n_pts);
            // the pre-filter's arithmetic: f16 where the network fits its range (8 x closer to the exact pre-activations: tighter margins, a shorter
            // list), else bf16; a tile of the f16 pass that saw an activation leave the range counts in slots[12] and render_device goes back to bf16
            const bool pf16 = c->cert_prefilter_f16[which];
            const float *pf_stream = pf16 ? (const float *)net.wstream_f16v2 : stream_of(net, NERF_MLP_BF16);
            MlpArgs b = a;
            b.wstream = pf_stream; b.small_params = net.small; b.n_points = n_pts; b.samples_per_ray = spr; b.t = t_in;
            b.sigma_out = sigma_out; b.rgb_out = nullptr; b.raw_pre = 1; b.skip_empty = 0; b.skip_counter = nullptr; b.nonfinite = pf16 ? slots + 12 : nullptr;
            const int kind_side = which == 0 && !rgb_out ? 0 : 4;
            {
                Timed t(c, st, kind_side, 0, timing);
                if (c->cert_seq_prefilter) {
                    // the pre-filter walks each ray front to back and stops where its own (bf16) transmittance predicts the cut, a little
                    // later than k_cert_plan will (depth + 0.5): samples it never reaches stay NaN = "not evaluated" = uncertain
                    HIP_TRY(c, hipMemsetAsync(sigma_out, 0xFF, (size_t)n_pts * sizeof(float), st));
                    SeqArgs q{};
                    q.wstream = pf_stream; q.small_params = net.small; q.n_rays = n_rays; q.samples_per_ray = spr;
                    q.ray_dirs = c->d_dirs; q.t = t_in; q.far_ = cam->far_;
                    q.origin[0] = cam->pos[0]; q.origin[1] = cam->pos[1]; q.origin[2] = cam->pos[2];
                    q.sigma_out = sigma_out; q.ray_counter = slots + 8; q.stats = (unsigned long long *)(slots + 10);
                    q.prefilter = 1; q.prefilter_cut_T = expf(-(c->cert_depth_limit + 0.5f));
                    q.nonfinite = pf16 ? slots + 12 : nullptr;
                    HIP_TRY(c, pf16 ? nerf_trunk_seq_f16v2_launch(q, c->n_cus, st) : nerf_trunk_seq_bf16_launch(q, false, c->n_cus, st));
                } else HIP_TRY(c, pf16 ? nerf_mlp_f16v2_launch(b, c->n_cus, st) : launch_mlp(c, NERF_MLP_BF16, b, false, st));
                CertPlanArgs p{};
                p.pre = sigma_out; p.t = t_in; p.n_rays = n_rays; p.spr = spr; p.far_ = cam->far_;
                p.margin = c->cert_margin[which]; p.depth_limit = c->cert_depth_limit;
                p.audit_mask = c->cert_audit_mask; p.audit_mask_near = c->cert_audit_mask_near; p.audit_salt = (unsigned)(o->seed * 0x9E3779B97F4A7C15ull >> 32) + 0x632BE5ABu * passes + (unsigned)which;
                p.list = c->d_point_list; p.count = slots; p.capacity = cap; p.jstar = c->d_jstar;
                // full evaluations: probable zeros (pre-filter pre-activation below -margin x cert_zero_frac: several times the largest pre-filter error the lego audits see
                // is not needed for a skip -- a positive density among them only keeps its tile's colour heads) and the audited certificates go
                // to the back part of the list, whose all-zero tiles skip the colour head (exact: skip_empty)
                if (rgb_out && c->cert_zero_tiles) { p.count_back = slots + 9; p.zero_threshold = c->cert_margin[which] * c->cert_zero_frac; }
                p.aux = c->d_cert_aux; p.aux_count = slots + 7; p.aux_capacity = (unsigned)aux_cap;
                HIP_TRY(c, launch_cert_plan(p, st));
                if (rgb_out) HIP_TRY(c, hipMemsetAsync(rgb_out, 0, (size_t)n_pts * 3 * sizeof(float), st)); // weight 0 either way: 0 * 0
                t.done(c->last_render);
            }
            MlpArgs l = b;
            l.wstream = stream_of(net, dt); l.raw_pre = 0; l.mode = MLP_MODE_LIST; l.rgb_out = rgb_out; l.n_points = (int)cap;
            l.point_list = c->d_point_list; l.point_list_count = slots;
            if (rgb_out && c->cert_zero_tiles) { l.point_list_count_back = slots + 9; l.skip_empty = 1; l.skip_counter = c->d_skip; }
            l.nonfinite = split_dtype(dt) ? c->d_nonfinite : nullptr;
            if (cap > 0) {
                // points = 0: the list's length is known on the device only (nerf_stats.n_exec_* report it after the frame)
                Timed t(c, st, kind, 0, timing);
                HIP_TRY(c, launch_mlp(c, dt, l, rgb_out != nullptr, st));
                t.done(c->last_render);
            }
            Timed t(c, st, kind_side, 0, timing);
            HIP_TRY(c, launch_cert_audit(c->d_cert_aux, slots + 7, (unsigned)aux_cap, sigma_out, slots + 2, c->n_cus, st));
            CertVerifyArgs v{};
            v.t = t_in; v.sigma = sigma_out; v.n_rays = n_rays; v.spr = spr; v.far_ = cam->far_; v.jstar = c->d_jstar;
            v.list = c->d_point_list; v.count = slots + 1; v.capacity = cap; v.fallback_rays = slots + 6;
            HIP_TRY(c, launch_cert_verify(v, st));
            l.point_list_count = slots + 1; l.point_list_count_back = nullptr;
            if (cap > 0) HIP_TRY(c, launch_mlp(c, dt, l, rgb_out != nullptr, st)); // usually an empty list: the launch returns at once
            t.done(c->last_render);
            return NERF_OK;
        };
        // coarse network: sigma only unless its colours are composited (reference discards them, src/lib.rs:404)
        a.wstream = stream_of(NC, dtype_coarse); a.small_params = NC.small;
        a.n_points = n_rays * nc; a.samples_per_ray = nc; a.t = c->d_tc;
        a.sigma_out = c->d_sc; a.rgb_out = o->coarse_only ? c->d_rgbc : nullptr; // sigma-only launch otherwise
        a.skip_empty = o->skip_empty; a.skip_counter = o->skip_empty ? c->d_skip : nullptr; // only full kernels look at it
        a.nonfinite = split_dtype(dtype_coarse) ? c->d_nonfinite : nullptr;
        if (seq) {
            if ((rc = seq_pass(NC, dtype_coarse, nc, c->d_tc, c->d_sc, o->coarse_only ? c->d_rgbc : nullptr, 3 * (int)passes, o->coarse_only ? 1 : 0))) return rc;
        } else if (certify) {
            if ((rc = cert_pass(NC, 0, dtype_coarse, nc, c->d_tc, c->d_sc, o->coarse_only ? c->d_rgbc : nullptr, o->coarse_only ? 1 : 0))) return rc;
        } else {
            Timed t(c, st, o->coarse_only ? 1 : 0, (uint64_t)a.n_points, timing);
            HIP_TRY(c, launch_mlp(c, dtype_coarse, a, o->coarse_only != 0, st));
            t.done(c->last_render);
        }
        float *pass_out = ray_out + (size_t)row * RW * 3;
        CompositeArgs ca{};
        ca.n_rays = n_rays; ca.far_ = cam->far_; ca.out = pass_out;
        if (o->coarse_only) {
            ca.n = nc; ca.t = c->d_tc; ca.sigma = c->d_sc; ca.rgb = c->d_rgbc;
            Timed t(c, st, 2, 0, timing);
            HIP_TRY(c, launch_composite(ca, st));
            t.done(c->last_render);
            continue;
        }
        const float *t_fine = c->d_tc;
        if (nf > 0) {
            ResampleArgs ra{};
            ra.g = g; ra.n_rays = n_rays; ra.nc = nc; ra.nf = nf; ra.far_ = cam->far_;
            ra.seed_lo = (uint32_t)o->seed; ra.seed_hi = (uint32_t)(o->seed >> 32);
            ra.t_coarse = c->d_tc; ra.sigma_coarse = c->d_sc; ra.t_fine = c->d_tf;
            unsigned int *hctr = seq ? c->d_seq + 4 * (size_t)(3 * passes + 2) : nullptr; // {queue head, flagged-ray count, u64 chunks}
            if (hybrid) { ra.flag_tau = c->hybrid_tau; ra.flag_count = hctr + 1; ra.flag_list = c->d_flag_list; }
            {
                Timed t(c, st, 2, 0, timing);
                HIP_TRY(c, launch_resample(ra, st));
                t.done(c->last_render);
            }
            if (hybrid) {
                // The coarse densities came from the split arithmetic.  Redo, in exact f32, the rays with a draw in a light CDF
                // bin (their sample positions are the ill-conditioned ones) and resample just those: same positions as the f32 path.
                SeqArgs q{};
                q.wstream = NC.wstream; q.small_params = NC.small; q.n_rays = n_rays; q.samples_per_ray = nc;
                q.ray_dirs = c->d_dirs; q.t = c->d_tc; q.far_ = cam->far_;
                q.origin[0] = cam->pos[0]; q.origin[1] = cam->pos[1]; q.origin[2] = cam->pos[2];
                q.sigma_out = c->d_sc; q.ray_counter = hctr; q.stats = (unsigned long long *)(hctr + 2);
                q.live_count = nullptr; // hctr[1] is the LENGTH of the ray list below: this launch never exports (no colour head)
                q.ray_list = c->d_flag_list; q.ray_list_count = hctr + 1; q.zero_fill_after_cut = 1;
                {
                    Timed t(c, st, 0, 0, timing);
                    HIP_TRY(c, nerf_trunk_seq_launch(q, false, c->n_cus, st));
                    t.done(c->last_render);
                }
                ResampleArgs rb = ra;
                rb.flag_tau = 0.0f; rb.flag_count = nullptr; rb.flag_list = nullptr;
                rb.ray_list = c->d_flag_list; rb.ray_list_count = hctr + 1;
                Timed t(c, st, 2, 0, timing);
                HIP_TRY(c, launch_resample(rb, st));
                t.done(c->last_render);
            }
            t_fine = c->d_tf;
        }
        a.wstream = stream_of(NF, dtype); a.small_params = NF.small;
        a.n_points = n_rays * M; a.samples_per_ray = M; a.t = t_fine;
        a.sigma_out = c->d_sf; a.rgb_out = c->d_rgbf;
        a.clock_out = c->d_clock; // NULL unless NERF_DEBUG_CLOCK=1
        a.nonfinite = split_dtype(dtype) ? c->d_nonfinite : nullptr;
        c->clock_valid = c->d_clock != nullptr;
        if (seq) {
            if ((rc = seq_pass(NF, dtype, M, t_fine, c->d_sf, c->d_rgbf, 3 * (int)passes + 1, 1))) return rc;
        } else if (certify) {
            if ((rc = cert_pass(NF, 1, dtype, M, t_fine, c->d_sf, c->d_rgbf, 1))) return rc;
        } else {
            Timed t(c, st, 1, (uint64_t)a.n_points, timing);
            HIP_TRY(c, launch_mlp(c, dtype, a, true, st));
            t.done(c->last_render);
        }
        ca.n = M; ca.t = t_fine; ca.sigma = c->d_sf; ca.rgb = c->d_rgbf;
        {
            Timed t(c, st, 2, 0, timing);
            HIP_TRY(c, launch_composite(ca, st));
            t.done(c->last_render);
        }
    }
    if (s > 1) {
        Timed t(c, st, 2, 0, timing);
        HIP_TRY(c, launch_box_downsample(c->d_rayfb, d_out, cw, ch, s, st));
        t.done(c->last_render);
    }
    // the dominant-kernel events also feed nerf_kernel_time_query (ownership: see recycle_render)
    for (const auto &p : c->last_render)
        if (p.kind == 1) c->dominant.push_back(p);
    scope.ok = true;
    if (cert) { // the audit's outcome decides whether this frame stands (render_device): certify_zero renders synchronise the stream
        HIP_TRY(c, hipStreamSynchronize(st));
        std::vector<unsigned int> h((size_t)passes * 2 * kCertSlots);
        HIP_TRY(c, hipMemcpy(h.data(), c->d_cert, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < passes * 2; ++k) {
            const unsigned int *q = &h[(size_t)k * kCertSlots];
            const int w = (int)(k & 1);
            if (w == 1 && o->coarse_only) continue;
            cert->used[w] = true;
            cert->listed[w] += std::min<uint64_t>((uint64_t)q[0] + q[9], cert->list_capacity) + std::min<uint64_t>(q[1], cert->list_capacity);
            cert->list_need = std::max<uint64_t>(cert->list_need, std::max<uint64_t>((uint64_t)q[0] + q[9], q[1]));
            cert->audited[w] += q[2]; cert->violations[w] += q[3];
            if (q[2]) {
                const unsigned bits = 0x7f800000u - q[4];
                float hr, er;
                memcpy(&hr, &bits, sizeof hr); memcpy(&er, &q[5], sizeof er);
                cert->headroom[w] = std::min(cert->headroom[w], hr); cert->max_err[w] = std::max(cert->max_err[w], er);
            }
            cert->fallback_rays += q[6];
            cert->range_left[w] += q[12];
        }
    }
    if (stats) {
        HIP_TRY(c, hipStreamSynchronize(st));
        memset(stats, 0, sizeof *stats);
        stats->n_rays = (uint64_t)RW * RH;
        stats->n_passes = passes;
        stats->n_coarse_points = stats->n_rays * (uint64_t)nc;                       // nominal: every sample of every ray
        stats->n_fine_points = o->coarse_only ? 0 : stats->n_rays * (uint64_t)M;
        for (const auto &p : c->last_render) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
            if (p.kind == 0) { stats->ms_coarse_mlp += ms; stats->n_mlp_launches++; }
            else if (p.kind == 1) {
                stats->n_mlp_launches++;
                if (o->coarse_only) stats->ms_coarse_mlp += ms; else stats->ms_fine_mlp += ms;
            } else if (p.kind == 4) { // colour head on the compacted live samples (skip_dead)
                stats->n_mlp_launches++;
                if (o->coarse_only) stats->ms_coarse_mlp += ms; else stats->ms_fine_mlp += ms;
            } else stats->ms_other += ms;
        }
        if (!c->last_render.empty()) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->last_render.front().a, c->last_render.back().b));
            stats->ms_total = ms;
        }
        if (watch_range) {
            unsigned int bad = 0;
            HIP_TRY(c, hipMemcpy(&bad, c->d_nonfinite, sizeof bad, hipMemcpyDeviceToHost));
            stats->n_nonfinite_points = bad;
        }
        if ((o->skip_empty || certify) && c->d_skip) { // certify_zero: all-zero tiles of the list's back part (probable zeros, audited certificates)
            unsigned long long tiles = 0;
            HIP_TRY(c, hipMemcpy(&tiles, c->d_skip, sizeof tiles, hipMemcpyDeviceToHost));
            stats->n_colour_skipped_points = (uint64_t)tiles * nerfmlp::kPointsPerBlock;
        }
        // evaluations actually executed
        stats->n_exec_coarse_trunk = stats->n_coarse_points;
        stats->n_exec_fine_trunk = stats->n_fine_points;
        stats->n_exec_colour = (o->coarse_only ? stats->n_coarse_points : stats->n_fine_points) - stats->n_colour_skipped_points;
        if (certify && cert) { // samples the exact kernel evaluated = the lengths of the lists (audited certificates included)
            stats->n_exec_coarse_trunk = cert->listed[0];
            stats->n_exec_fine_trunk = o->coarse_only ? 0 : cert->listed[1];
            stats->n_exec_colour = (o->coarse_only ? cert->listed[0] : cert->listed[1]) - std::min<uint64_t>(stats->n_colour_skipped_points, o->coarse_only ? cert->listed[0] : cert->listed[1]);
            stats->n_certify_audited = cert->audited[0] + cert->audited[1];
            stats->n_certify_violations = cert->violations[0] + cert->violations[1];
            stats->n_certify_fallback_rays = (uint32_t)std::min<uint64_t>(cert->fallback_rays, 0xffffffffu);
            for (int w = 0; w < 2; ++w) { stats->certify_margin[w] = c->cert_margin[w]; stats->certify_headroom[w] = cert->headroom[w]; stats->certify_max_error[w] = cert->max_err[w]; }
        }
        if (seq) {
            std::vector<unsigned int> h((size_t)passes * 3 * 4);
            HIP_TRY(c, hipMemcpy(h.data(), c->d_seq, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
            uint64_t evaluated[3] = {0, 0, 0}, live = 0; // samples the trunk launches evaluated (a ray's last chunk may be partial)
            for (uint32_t k = 0; k < passes * 3; ++k) {
                unsigned long long ch64;
                memcpy(&ch64, &h[4 * (size_t)k + 2], sizeof ch64);
                evaluated[k % 3] += ch64;
                if (k % 3 == 2) stats->n_hybrid_rays += h[4 * (size_t)k + 1]; // flagged rays redone in f32
                else live += h[4 * (size_t)k + 1];
            }
            stats->n_exec_coarse_trunk = evaluated[0] + evaluated[2];
            stats->n_exec_fine_trunk = o->coarse_only ? 0 : evaluated[1];
            stats->n_exec_colour = live;
            stats->n_colour_skipped_points = (o->coarse_only ? stats->n_coarse_points : stats->n_fine_points) - live;
        }
        if (stats->n_nonfinite_points) { // the frame is WRONG there (f16 overflow yields finite garbage, not NaN): never return it as a success
            char msg[320];
            snprintf(msg, sizeof msg, "%llu evaluations: an operand left the range of the split arithmetic (NERF_MLP_F16X2: |activation| <= 65504) or a "
                     "density was not finite; the frame is not usable -- use NERF_MLP_BF16X3 or NERF_MLP_F32", (unsigned long long)stats->n_nonfinite_points);
            return fail(c, NERF_ERR_STATE, msg);
        }
    }
    return NERF_OK;
}

} // namespace

// certify_zero frames are AUDITED (k_cert_audit: 1 in 16 of the samples certified by less than twice the margin and 1 in 128 of the others are evaluated exactly all the same).  A frame stands only if no
// audited certificate was wrong and the closest audited sample kept at least half the margin between itself and a positive density;
// otherwise the network's margin is widened -- for good: c->cert_margin is per context and network, reset when a network is loaded --
// and the frame is rendered again.  The same loop grows the sample list when a pass wanted more entries than it had.  Margins are
// measurements, not proofs: what this buys is that a network on which bf16 is less accurate than on the lego scene
// calibrates itself or fails loudly (NERF_ERR_STATE) instead of returning a silently different frame.
int nerfint::render_device(nerf_ctx *c, const nerf_camera *cam, const nerf_render_opts *o, float *d_out, hipStream_t st,
                           nerf_stats *stats) {
    if (!o || !o->certify_zero) return render_once(c, cam, o, d_out, st, stats, nullptr);
    constexpr int kMaxRetries = 8;
    uint64_t violations = 0;
    for (int attempt = 0;; ++attempt) {
        CertOutcome oc;
        const int rc = render_once(c, cam, o, d_out, st, stats, &oc);
        if (rc) return rc;
        bool again = false;
        std::string why;
        if (oc.list_need > oc.list_capacity) { // (entries beyond the capacity were counted, not stored)
            c->cert_list_frac = std::min(1.0, 1.25 * (double)oc.list_need / (double)std::max<uint64_t>(oc.pass_samples, 1));
            again = true;
            why = "the sample list was too short";
        }
        for (int w = 0; w < 2; ++w)
            if (oc.range_left[w]) { // the f16 pre-filter met an activation beyond 65 504: its pre-activations are void -- bf16 (f32's range) from now on
                c->cert_prefilter_f16[w] = false;
                c->cert_margin[w] = std::max(c->cert_margin[w], c->cert_margin_floor[w]);
                again = true;
                why = "an activation left the range of the f16 pre-filter";
            }
        const bool overflow = again; // an attempt whose list was too short evaluated only a part of it (or whose pre-filter left its range): its audit says nothing
        if (!overflow) violations += oc.violations[0] + oc.violations[1];
        for (int w = 0; w < 2 && !overflow; ++w) {
            if (!oc.used[w] || !oc.audited[w]) continue;
            // the rules live in host_util.cpp (certify_policy: tested on the CPU through nerf_debug_certify_policy)
            float widened = c->cert_margin[w];
            const int rule = certify_policy(c->cert_margin[w], oc.audited[w], oc.violations[w], oc.headroom[w], oc.max_err[w], &widened);
            if (rule) {
                c->cert_margin[w] = widened; again = true;
                why = rule == 1 ? "an audited certificate was wrong" : rule == 2 ? "an audited certificate came closer to a positive density than half the margin"
                                                                                 : "the 16-bit pass was off by more than half the margin on an audited certificate";
            }
        }
        if (stats) { stats->n_certify_retries = (uint32_t)attempt; stats->n_certify_violations = violations; }
        if (!again) return NERF_OK;
        if (attempt == kMaxRetries) {
            char msg[320];
            snprintf(msg, sizeof msg, "certify_zero: %s after %d renders of this frame (margins now %g / %g): the 16-bit pass cannot certify this network's "
                     "zero densities; render with certify_zero = 0", why.c_str(), attempt + 1, (double)c->cert_margin[0], (double)c->cert_margin[1]);
            return fail(c, NERF_ERR_STATE, msg);
        }
    }
}

// ================================================================================================
// extern "C"
// ================================================================================================
// No C++ exception may cross the C ABI (a Rust caller would abort): every entry point that can allocate is a function-try-block.
#define NERF_CATCH(c)                                                                                              \
    catch (const std::bad_alloc &) { return fail((c), NERF_ERR_IO, "out of host memory"); }                         \
    catch (const std::exception &e) { return fail((c), NERF_ERR_INVALID, std::string("internal error: ") + e.what()); } \
    catch (...) { return fail((c), NERF_ERR_INVALID, "internal error"); }

extern "C" {

int nerf_abi_version(void) { return 5; }

void nerf_abi_struct_sizes(size_t *camera, size_t *render_opts, size_t *stats) {
    if (camera) *camera = sizeof(nerf_camera);
    if (render_opts) *render_opts = sizeof(nerf_render_opts);
    if (stats) *stats = sizeof(nerf_stats);
}

const char *nerf_last_error(const nerf_ctx *ctx) { return ctx ? ctx->err.c_str() : nerfhost::last_error_noctx(); }

int nerf_create(int device_id, nerf_ctx **out) try {
    if (!out) return fail(nullptr, NERF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, NERF_ERR_HIP, std::string("no HIP device available (") + hipGetErrorString(e) + "); libnerf_mi355x has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(nullptr, NERF_ERR_INVALID, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    std::string arch = prop.gcnArchName;
    if (arch.rfind("gfx950", 0) != 0)
        return fail(nullptr, NERF_ERR_HIP, "device is " + arch + ", this library is built for gfx950 (MI355X) only");
    nerf_ctx *c = new nerf_ctx();
    c->device = device_id;
    c->n_cus = prop.multiProcessorCount;
    c->arch = arch;
    if (const char *env = getenv("NERF_MAX_RAYS_PER_PASS")) {
        const long long v = atoll(env);
        if (v > 0) c->max_rays_per_pass = (size_t)v;
    }
    if (const char *env = getenv("NERF_MAX_EXPORT_BYTES")) {
        const long long v = atoll(env);
        if (v > 0) c->max_export_bytes = (size_t)v;
    }
    if (const char *env = getenv("NERF_HYBRID_TAU")) { // experiments only: the predicted sample displacement (in t) above which a ray is redone in f32
        const double v = atof(env);
        if (v >= 0.0 && v <= 1.0) c->hybrid_tau = (float)v;
    }
#ifdef NERF_CERT_TUNING // variant builds only (make variant DEFS=-DNERF_CERT_TUNING=1): the product's certificates are not configurable from the environment
    if (const char *env = getenv("NERF_CERTIFY_SEQ_PREFILTER")) c->cert_seq_prefilter = atoi(env) != 0;
    if (const char *env = getenv("NERF_CERTIFY_PREFILTER_F16")) c->cert_allow_f16 = atoi(env) != 0;
    if (const char *env = getenv("NERF_CERTIFY_MARGINS_F16")) { float m0 = 0.f, m1 = 0.f; if (sscanf(env, "%f,%f", &m0, &m1) == 2 && m0 > 0.f && m1 > 0.f) { c->cert_margin_floor_f16[0] = m0; c->cert_margin_floor_f16[1] = m1; } }
    if (const char *env = getenv("NERF_CERTIFY_ZERO_TILES")) c->cert_zero_tiles = atoi(env) != 0;
    if (const char *env = getenv("NERF_CERTIFY_ZERO_FRAC")) { const double v = atof(env); if (v > 0.0 && v <= 1.0) c->cert_zero_frac = (float)v; }
    if (const char *env = getenv("NERF_CERTIFY_CUT_DEPTH")) { const double v = atof(env); if (v > 0.0) c->cert_depth_limit = (float)v; }
    if (const char *env = getenv("NERF_CERTIFY_AUDIT_MASK")) { const long v = atol(env); if (v >= 0 && ((v + 1) & v) == 0) c->cert_audit_mask = (unsigned)v; }
    if (const char *env = getenv("NERF_CERTIFY_AUDIT_MASK_NEAR")) { const long v = atol(env); if (v >= 0 && ((v + 1) & v) == 0) c->cert_audit_mask_near = (unsigned)v; }
    if (const char *env = getenv("NERF_CERTIFY_MARGINS")) {
        float m0 = 0.f, m1 = 0.f;
        if (sscanf(env, "%f,%f", &m0, &m1) == 2 && m0 > 0.f && m1 > 0.f) { c->cert_margin_floor[0] = m0; c->cert_margin_floor[1] = m1; c->cert_margin[0] = m0; c->cert_margin[1] = m1; }
    }
#endif
    if (const char *env = getenv("NERF_DEBUG_CLOCK")) {
        if (atoi(env) > 0 && hipMalloc((void **)&c->d_clock, (size_t)c->n_cus * 2 * sizeof(unsigned long long)) != hipSuccess) c->d_clock = nullptr;
    }
    if (hipMalloc((void **)&c->d_skip, sizeof(unsigned long long)) != hipSuccess) c->d_skip = nullptr;
    if (hipMalloc((void **)&c->d_nonfinite, sizeof(unsigned int)) != hipSuccess) c->d_nonfinite = nullptr;
    // kernel attributes (dynamic LDS sizes) are per device, not per context: set them once per device and process
    static std::mutex init_mu;
    static std::set<int> init_done;
    hipError_t e1 = hipSuccess, e2 = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(init_mu);
        if (!init_done.count(device_id)) {
            e1 = nerf_mlp_init();
            if (e1 == hipSuccess) e1 = nerf_mlp_bf16v2_init();
            if (e1 == hipSuccess) e1 = nerf_prefilter_f16v2_init();
#if NERF_BF16_V3
            if (e1 == hipSuccess) e1 = nerf_mlp_bf16v3_init();
#endif
            if (e1 == hipSuccess) e1 = nerf_mlp_bf16x3_init();
            if (e1 == hipSuccess) e1 = nerf_seq_init();
            if (e1 == hipSuccess) e1 = nerf_seq_bf16_init();
            if (e1 == hipSuccess) e1 = nerf_seq_x3_init();
            if (e1 == hipSuccess) e1 = nerf_mlp_f16x2_init();
            if (e1 == hipSuccess) e1 = nerf_seq_f16x2_init();
            e2 = e1 == hipSuccess ? sampling_init() : e1;
            if (e2 == hipSuccess) init_done.insert(device_id);
        }
    }
    hipError_t e3 = e2 == hipSuccess ? hipStreamCreate(&c->stream) : e2;
    if (e3 != hipSuccess) {
        const std::string m = std::string("context initialisation failed: ") + hipGetErrorString(e3);
        nerf_destroy(c); // frees whatever was allocated so far (counters, clock buffer, stream)
        return fail(nullptr, NERF_ERR_HIP, m);
    }
    *out = c;
    return NERF_OK;
} NERF_CATCH(nullptr)

void nerf_destroy(nerf_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c->device);
    (void)hipDeviceSynchronize();
    for (auto &n : c->net) { if (n.wstream) (void)hipFree(n.wstream); if (n.small) (void)hipFree(n.small); if (n.wstream_bf16v2) (void)hipFree(n.wstream_bf16v2); if (n.wstream_bf16v3) (void)hipFree(n.wstream_bf16v3); if (n.wstream_x3) (void)hipFree(n.wstream_x3); if (n.wstream_x2) (void)hipFree(n.wstream_x2); if (n.wstream_f16v2) (void)hipFree(n.wstream_f16v2); }
    float *ptrs[] = {c->d_dirs, c->d_tc, c->d_sc, c->d_rgbc, c->d_tf, c->d_sf, c->d_rgbf, c->d_rayfb, c->d_out};
    for (float *p : ptrs) if (p) (void)hipFree(p);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_clock) (void)hipFree(c->d_clock);
    if (c->d_skip) (void)hipFree(c->d_skip);
    if (c->d_nonfinite) (void)hipFree(c->d_nonfinite);
    if (c->d_seq) (void)hipFree(c->d_seq);
    if (c->d_h8) (void)hipFree(c->d_h8);
    if (c->d_slot_point) (void)hipFree(c->d_slot_point);
    if (c->d_flag_list) (void)hipFree(c->d_flag_list);
    if (c->d_point_list) (void)hipFree(c->d_point_list);
    if (c->d_jstar) (void)hipFree(c->d_jstar);
    if (c->d_cert_aux) (void)hipFree(c->d_cert_aux);
    if (c->d_cert) (void)hipFree(c->d_cert);
    recycle_render(c);
    recycle_dominant(c, 0);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int nerf_device_info(const nerf_ctx *c, int *n_cus, char *arch_name, size_t len) {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (n_cus) *n_cus = c->n_cus;
    if (arch_name && len) snprintf(arch_name, len, "%s", c->arch.c_str());
    return NERF_OK;
}

int nerf_load_network_dir(nerf_ctx *c, int which, const char *dir) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail(c, NERF_ERR_INVALID, "which must be NERF_NET_COARSE or NERF_NET_FINE");
    if (!dir) return fail(c, NERF_ERR_INVALID, "dir is NULL");
    DeviceGuard dg(c->device);
    std::map<std::string, Tensor> params;
    std::string err;
    int rc = read_tensor_dir(dir, params, err);
    if (rc) return fail(c, rc, err);
    HostNet hn;
    rc = assemble_net(params, hn, err);
    if (rc) return fail(c, rc, err);
    return upload_net(c, which, hn);
} NERF_CATCH(c)

int nerf_load_network_tensors(nerf_ctx *c, int which, int n, const char *const *names, const int64_t *dims,
                              const float *const *data) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail(c, NERF_ERR_INVALID, "which must be NERF_NET_COARSE or NERF_NET_FINE");
    if (n < 0 || (n > 0 && (!names || !dims || !data))) return fail(c, NERF_ERR_INVALID, "bad tensor table");
    DeviceGuard dg(c->device);
    std::map<std::string, Tensor> params;
    for (int i = 0; i < n; ++i) {
        if (!names[i] || !data[i] || dims[2 * i] < 0 || dims[2 * i + 1] < 0) return fail(c, NERF_ERR_INVALID, "bad tensor table entry");
        Tensor t;
        size_t cnt = (size_t)dims[2 * i];
        t.dims.push_back(dims[2 * i]);
        if (dims[2 * i + 1] > 0) { t.dims.push_back(dims[2 * i + 1]); cnt *= (size_t)dims[2 * i + 1]; }
        t.data.assign(data[i], data[i] + cnt);
        params[names[i]] = std::move(t);
    }
    HostNet hn;
    std::string err;
    const int rc = assemble_net(params, hn, err);
    if (rc) return fail(c, rc, err);
    return upload_net(c, which, hn);
} NERF_CATCH(c)

int nerf_load_network_blob(nerf_ctx *c, int which, const char *blob_path) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail(c, NERF_ERR_INVALID, "which must be NERF_NET_COARSE or NERF_NET_FINE");
    if (!blob_path) return fail(c, NERF_ERR_INVALID, "blob_path is NULL");
    DeviceGuard dg(c->device);
    std::vector<float> ws, sm;
    std::string err;
    const int rrc = read_blob_file(blob_path, ws, sm, err);
    if (rrc) return fail(c, rrc, err);
    const int rc = upload_packed(c, which, ws, sm);
    return rc ? rc : bf16_from_f32_stream(c, which, ws);
} NERF_CATCH(c)

static int forward_device(nerf_ctx *c, int which, int dtype, const float *d_pts, const float *d_dirs, size_t n, float *d_rgb,
                          float *d_sigma, void *stream);
// largest batch of nerf_forward_batch*: INT32_MAX minus (persistent workgroups + 1) tiles of the widest kernel (256 points)
static size_t max_batch_points(int n_cus) { return (size_t)0x7fffffff - ((size_t)n_cus + 1) * 256; }

int nerf_forward_batch_device(nerf_ctx *c, int which, const float *d_pts, const float *d_dirs, size_t n, float *d_rgb,
                              float *d_sigma, void *stream) try {
    return forward_device(c, which, NERF_MLP_F32, d_pts, d_dirs, n, d_rgb, d_sigma, stream);
} NERF_CATCH(c)

static int forward_device(nerf_ctx *c, int which, int dtype, const float *d_pts, const float *d_dirs, size_t n, float *d_rgb,
                          float *d_sigma, void *stream) {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail(c, NERF_ERR_INVALID, "which must be NERF_NET_COARSE or NERF_NET_FINE");
    if (n == 0) return NERF_OK; // src/network.rs:199-201
    if (!d_pts || !d_dirs || !d_rgb || !d_sigma) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    // the kernels index points in int32 and look one persistent-grid stride of tiles ahead (load_raw): keep that in range too
    if (n > max_batch_points(c->n_cus)) return fail(c, NERF_ERR_INVALID, "batch too large (n plus one grid stride of tiles must fit in int32)");
    if (!c->net[which].loaded) return fail(c, NERF_ERR_STATE, "network not loaded");
    DeviceGuard dg(c->device);
    MlpArgs a{};
    a.mode = MLP_MODE_POINTS;
    if (!valid_dtype(dtype)) return fail(c, NERF_ERR_INVALID, "mlp_dtype must be NERF_MLP_F32, NERF_MLP_BF16, NERF_MLP_BF16X3 or NERF_MLP_F16X2");
    if (dtype == NERF_MLP_F16X2 && !c->net[which].wstream_x2) return fail(c, NERF_ERR_STATE, "NERF_MLP_F16X2 is unavailable for this network: a weight exceeds the f16 range");
    a.wstream = stream_of(c->net[which], dtype); a.small_params = c->net[which].small;
    a.n_points = (int)n; a.pts_soa = d_pts; a.dirs_aos = d_dirs; a.sigma_out = d_sigma; a.rgb_out = d_rgb;
    a.nonfinite = split_dtype(dtype) ? c->d_nonfinite : nullptr;
    HIP_TRY(c, launch_mlp(c, dtype, a, true, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_forward_batch(nerf_ctx *c, int which, const float *pts, const float *dirs, size_t n, float *rgb, float *sigma) try {
    return nerf_forward_batch_ex(c, which, NERF_MLP_F32, pts, dirs, n, rgb, sigma);
} NERF_CATCH(c)

int nerf_forward_batch_ex(nerf_ctx *c, int which, int dtype, const float *pts, const float *dirs, size_t n, float *rgb, float *sigma) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (n == 0) return NERF_OK;
    if (!pts || !dirs || !rgb || !sigma) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    DeviceGuard dg(c->device);
    int rc;
    if ((rc = ensure_bytes(c, &c->d_scratch, &c->scratch_bytes, n * 10 * sizeof(float)))) return rc;
    float *d_pts = (float *)c->d_scratch, *d_dirs = d_pts + 3 * n, *d_rgb = d_dirs + 3 * n, *d_sig = d_rgb + 3 * n;
    HIP_TRY(c, hipMemcpyAsync(d_pts, pts, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_dirs, dirs, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const bool watch = split_dtype(dtype) && c->d_nonfinite;
    if (watch) HIP_TRY(c, hipMemsetAsync(c->d_nonfinite, 0, sizeof(unsigned int), c->stream));
    if ((rc = forward_device(c, which, dtype, d_pts, d_dirs, n, d_rgb, d_sig, c->stream))) return rc;
    HIP_TRY(c, hipMemcpyAsync(rgb, d_rgb, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sigma, d_sig, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    unsigned int bad = 0;
    if (watch) HIP_TRY(c, hipMemcpyAsync(&bad, c->d_nonfinite, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) {
        char msg[320];
        snprintf(msg, sizeof msg, "%u of %zu points: an operand left the range of the split arithmetic (NERF_MLP_F16X2: |activation| <= 65504) or a "
                 "density was not finite; the outputs are not usable -- use NERF_MLP_BF16X3 or NERF_MLP_F32", bad, n);
        return fail(c, NERF_ERR_STATE, msg);
    }
    return NERF_OK;
} NERF_CATCH(c)

int nerf_render_image_device(nerf_ctx *c, const nerf_camera *cam, const nerf_render_opts *opts, float *d_rgb_out,
                             void *stream, nerf_stats *stats) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    DeviceGuard dg(c->device);
    return render_device(c, cam, opts, d_rgb_out, (hipStream_t)stream, stats);
} NERF_CATCH(c)

int nerf_render_image(nerf_ctx *c, const nerf_camera *cam, const nerf_render_opts *opts, float *rgb_out, nerf_stats *stats) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (!rgb_out) return fail(c, NERF_ERR_INVALID, "output pointer is NULL");
    int rc;
    if ((rc = check_camera(c, cam))) return rc;
    if (!opts) return fail(c, NERF_ERR_INVALID, "opts is NULL");
    DeviceGuard dg(c->device);
    const bool crop = opts->crop_w > 0 || opts->crop_h > 0;
    const long long w = crop ? opts->crop_w : cam->nx;
    long long h = crop ? opts->crop_h : cam->ny;
    if (w <= 0 || h <= 0) return fail(c, NERF_ERR_INVALID, "crop window outside the frame");
    if (opts->band_count > 1) {
        if (opts->band_index < 0 || opts->band_index >= opts->band_count || opts->band_stripe_rows < 0) return fail(c, NERF_ERR_INVALID, "band_index / band_count / band_stripe_rows out of range");
        h = band_rows((int)h, opts->band_index, opts->band_count, opts->band_stripe_rows);
        if (h <= 0) return fail(c, NERF_ERR_INVALID, "this band has no rows (more bands than rows)");
    }
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    if ((rc = ensure_bytes(c, (void **)&c->d_out, &c->out_floats, bytes))) return rc;
    nerf_stats local; // a synchronous render always reads its counters: a frame computed outside a split arithmetic's range is an error
    if ((rc = render_device(c, cam, opts, c->d_out, c->stream, stats ? stats : &local))) return rc;
    HIP_TRY(c, hipMemcpyAsync(rgb_out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NERF_OK;
} NERF_CATCH(c)

int nerf_band_rows(int window_rows, int band_index, int band_count, int band_stripe_rows) {
    if (window_rows < 0 || band_count < 0 || band_stripe_rows < 0 || (band_count > 1 && (band_index < 0 || band_index >= band_count)))
        return fail(nullptr, NERF_ERR_INVALID, "nerf_band_rows: bad argument");
    return band_rows(window_rows, band_index, band_count, band_stripe_rows);
}

int nerf_kernel_time_query(nerf_ctx *c, double *ms, uint64_t *points, uint32_t *n_launches, int reset) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    DeviceGuard dg(c->device);
    double tot = 0.0; uint64_t pts = 0; uint32_t nl = 0;
    for (const auto &p : c->dominant) {
        HIP_TRY(c, hipEventSynchronize(p.b));
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, p.a, p.b));
        tot += t; pts += p.points; ++nl;
    }
    if (ms) *ms = tot;
    if (points) *points = pts;
    if (n_launches) *n_launches = nl;
    if (reset) {
        // last_render may still list these pairs (kind 1); it never recycles them, and they are not read again
        for (auto &p : c->last_render) if (p.kind == 1) p.kind = 3;
        c->last_render.erase(std::remove_if(c->last_render.begin(), c->last_render.end(), [](const EvPair &p) { return p.kind == 3; }), c->last_render.end());
        recycle_dominant(c, 0);
    }
    return NERF_OK;
} NERF_CATCH(c)

int nerf_debug_shader_clock_mhz(nerf_ctx *c, double *mhz) try {
    if (!c || !mhz) return fail(c, NERF_ERR_INVALID, "NULL argument");
    if (!c->d_clock || !c->clock_valid) return fail(c, NERF_ERR_STATE, "set NERF_DEBUG_CLOCK=1 before nerf_create and render a frame first");
    DeviceGuard dg(c->device);
    std::vector<unsigned long long> h((size_t)c->n_cus * 2);
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(h.data(), c->d_clock, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> f;
    for (int i = 0; i < c->n_cus; ++i) if (h[2 * i + 1] > 0) f.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    if (f.empty()) return fail(c, NERF_ERR_STATE, "no clock samples");
    std::sort(f.begin(), f.end());
    *mhz = f[f.size() / 2];
    return NERF_OK;
} NERF_CATCH(c)

// ---- stage entry points ------------------------------------------------------------------------------------
static int stage_rect(nerf_ctx *c, const nerf_camera *cam, int x0, int y0, int w, int h, RayGenArgs &g) {
    int rc;
    if ((rc = check_camera(c, cam))) return rc;
    if (w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > cam->nx || y0 + h > cam->ny) return fail(c, NERF_ERR_INVALID, "rectangle outside the frame");
    g = make_raygen(*cam, 1);
    g.n_rays = w * h; g.rx0 = x0; g.ry0 = y0; g.rw = w;
    return NERF_OK;
}

int nerf_stage_ray_dirs(nerf_ctx *c, const nerf_camera *cam, int x0, int y0, int w, int h, int normalize, float *out) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    DeviceGuard dg(c->device);
    RayGenArgs g; int rc;
    if ((rc = stage_rect(c, cam, x0, y0, w, h, g))) return rc;
    g.normalize = normalize ? 1 : 0;
    const size_t bytes = (size_t)g.n_rays * 3 * sizeof(float);
    if ((rc = ensure_bytes(c, &c->d_scratch, &c->scratch_bytes, bytes))) return rc;
    HIP_TRY(c, launch_ray_dirs(g, (float *)c->d_scratch, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NERF_OK;
} NERF_CATCH(c)

int nerf_stage_stratified(nerf_ctx *c, const nerf_camera *cam, int x0, int y0, int w, int h, int count, uint64_t seed, float *out) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    if (count <= 0) return NERF_OK; // src/lib.rs:235-237
    DeviceGuard dg(c->device);
    RayGenArgs g; int rc;
    if ((rc = stage_rect(c, cam, x0, y0, w, h, g))) return rc;
    const size_t bytes = (size_t)g.n_rays * count * sizeof(float);
    if ((rc = ensure_bytes(c, &c->d_scratch, &c->scratch_bytes, bytes))) return rc;
    HIP_TRY(c, launch_stratified(g, count, cam->near_, cam->far_, seed, (float *)c->d_scratch, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NERF_OK;
} NERF_CATCH(c)

static int stage_resample(nerf_ctx *c, size_t n_rays, int nc, int nf, float far_, uint64_t seed, const uint32_t *pixel_index,
                          const float *t_coarse, const float *sigma_coarse, const float *u, float *w_out, float *cdf_out,
                          float *t_new_out, float *t_fine_out, float tau, uint8_t *flags_out);

int nerf_stage_resample(nerf_ctx *c, size_t n_rays, int nc, int nf, float far_, uint64_t seed, const uint32_t *pixel_index,
                        const float *t_coarse, const float *sigma_coarse, const float *u, float *w_out, float *cdf_out,
                        float *t_new_out, float *t_fine_out) try {
    return stage_resample(c, n_rays, nc, nf, far_, seed, pixel_index, t_coarse, sigma_coarse, u, w_out, cdf_out, t_new_out, t_fine_out, 0.0f, nullptr);
} NERF_CATCH(c)

int nerf_stage_hybrid_flags(nerf_ctx *c, size_t n_rays, int nc, int nf, float far_, uint64_t seed, const uint32_t *pixel_index,
                            const float *t_coarse, const float *sigma_coarse, const float *u, float tau, uint8_t *flags_out,
                            float *t_new_out) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (n_rays && !flags_out) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    if (!(tau >= 0.0f)) return fail(c, NERF_ERR_INVALID, "tau must be >= 0 (0 selects the context's threshold)");
    return stage_resample(c, n_rays, nc, nf, far_, seed, pixel_index, t_coarse, sigma_coarse, u, nullptr, nullptr, t_new_out, nullptr,
                          tau > 0.0f ? tau : c->hybrid_tau, flags_out);
} NERF_CATCH(c)

static int stage_resample(nerf_ctx *c, size_t n_rays, int nc, int nf, float far_, uint64_t seed, const uint32_t *pixel_index,
                          const float *t_coarse, const float *sigma_coarse, const float *u, float *w_out, float *cdf_out,
                          float *t_new_out, float *t_fine_out, float tau, uint8_t *flags_out) {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (n_rays == 0) return NERF_OK;
    if (!t_coarse || !sigma_coarse || (!t_fine_out && !flags_out)) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    if (nc < 3 || nf <= 0) return fail(c, NERF_ERR_INVALID, "resample needs nc >= 3 and nf > 0 (src/lib.rs:295-297)");
    if (!u && !pixel_index) return fail(c, NERF_ERR_INVALID, "either u or pixel_index must be given");
    if (resample_lds_bytes(nc, nf) > 160 * 1024 || n_rays > 0x7fffffff / (size_t)(nc + nf)) return fail(c, NERF_ERR_INVALID, "too many samples");
    DeviceGuard dg(c->device);
    const size_t R = n_rays, M = (size_t)nc + nf;
    // layout in scratch (floats): tc, sc, u, w, cdf, tnew, tfine, pix
    const size_t o_tc = 0, o_sc = o_tc + R * nc, o_u = o_sc + R * nc, o_w = o_u + R * nf, o_cdf = o_w + R * nc,
                 o_tn = o_cdf + R * (nc - 1), o_tf = o_tn + R * nf, o_px = o_tf + R * M, o_fl = o_px + R, total = o_fl + (R + 3) / 4;
    int rc;
    if ((rc = ensure_bytes(c, &c->d_scratch, &c->scratch_bytes, total * sizeof(float)))) return rc;
    float *d = (float *)c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(d + o_tc, t_coarse, R * nc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d + o_sc, sigma_coarse, R * nc * 4, hipMemcpyHostToDevice, c->stream));
    if (u) HIP_TRY(c, hipMemcpyAsync(d + o_u, u, R * nf * 4, hipMemcpyHostToDevice, c->stream));
    if (pixel_index) HIP_TRY(c, hipMemcpyAsync(d + o_px, pixel_index, R * 4, hipMemcpyHostToDevice, c->stream));
    ResampleArgs ra{};
    ra.n_rays = (int)R; ra.nc = nc; ra.nf = nf; ra.far_ = far_;
    ra.seed_lo = (uint32_t)seed; ra.seed_hi = (uint32_t)(seed >> 32);
    ra.t_coarse = d + o_tc; ra.sigma_coarse = d + o_sc; ra.t_fine = d + o_tf;
    ra.pixel_index = pixel_index ? (const uint32_t *)(d + o_px) : nullptr;
    ra.u_in = u ? d + o_u : nullptr;
    ra.w_out = d + o_w; ra.cdf_out = d + o_cdf; ra.t_new_out = d + o_tn;
    ra.g.rw = 1; ra.g.rnx = 1; // unused when pixel_index/u are given
    if (flags_out) { ra.flag_tau = tau; ra.flag_out = (unsigned char *)(d + o_fl); }
    HIP_TRY(c, launch_resample(ra, c->stream));
    if (flags_out) HIP_TRY(c, hipMemcpyAsync(flags_out, d + o_fl, R, hipMemcpyDeviceToHost, c->stream));
    if (w_out) HIP_TRY(c, hipMemcpyAsync(w_out, d + o_w, R * nc * 4, hipMemcpyDeviceToHost, c->stream));
    if (cdf_out) HIP_TRY(c, hipMemcpyAsync(cdf_out, d + o_cdf, R * (nc - 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (t_new_out) HIP_TRY(c, hipMemcpyAsync(t_new_out, d + o_tn, R * nf * 4, hipMemcpyDeviceToHost, c->stream));
    if (t_fine_out) HIP_TRY(c, hipMemcpyAsync(t_fine_out, d + o_tf, R * M * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NERF_OK;
}

int nerf_stage_integrate(nerf_ctx *c, size_t n_rays, int n, float far_, const float *rgb, const float *sigma, const float *t,
                         float *rgb_out, float *w_out) try {
    if (!c) return fail(nullptr, NERF_ERR_INVALID, "ctx is NULL");
    if (n_rays == 0) return NERF_OK;
    if (!rgb_out) return fail(c, NERF_ERR_INVALID, "NULL buffer");
    if (n == 0) { memset(rgb_out, 0, n_rays * 3 * sizeof(float)); return NERF_OK; } // src/lib.rs:178-180
    if (n < 0 || !rgb || !sigma || !t) return fail(c, NERF_ERR_INVALID, "bad argument");
    if (composite_lds_bytes(n) > 160 * 1024 || n_rays > 0x7fffffff / (size_t)n) return fail(c, NERF_ERR_INVALID, "too many samples");
    DeviceGuard dg(c->device);
    const size_t R = n_rays, N = (size_t)n;
    const size_t o_t = 0, o_s = o_t + R * N, o_c = o_s + R * N, o_w = o_c + 3 * R * N, o_o = o_w + R * N, total = o_o + 3 * R;
    int rc;
    if ((rc = ensure_bytes(c, &c->d_scratch, &c->scratch_bytes, total * sizeof(float)))) return rc;
    float *d = (float *)c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(d + o_t, t, R * N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d + o_s, sigma, R * N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d + o_c, rgb, 3 * R * N * 4, hipMemcpyHostToDevice, c->stream));
    CompositeArgs ca{};
    ca.n_rays = (int)R; ca.n = n; ca.far_ = far_; ca.t = d + o_t; ca.sigma = d + o_s; ca.rgb = d + o_c; ca.out = d + o_o; ca.w_out = d + o_w;
    HIP_TRY(c, launch_composite(ca, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rgb_out, d + o_o, 3 * R * 4, hipMemcpyDeviceToHost, c->stream));
    if (w_out) HIP_TRY(c, hipMemcpyAsync(w_out, d + o_w, R * N * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NERF_OK;
} NERF_CATCH(c)

} // extern "C"
