// nerf_internal.h -- the context and the helpers shared by nerf_api.cpp (single-device entry points) and
// nerf_multi.cpp (multi-GPU fan-out).  Internal to libnerf_mi355x.so; nothing here is part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"

namespace nerfint {

struct DevNet {
    float *wstream = nullptr, *small = nullptr;
    uint16_t *wstream_bf16v2 = nullptr; // bf16 pieces in output-tile-major order (mlp_kernel_bf16v2.hip)
    uint16_t *wstream_bf16v3 = nullptr; // bf16 pieces for the 16x16x32 kernel (mlp_kernel_bf16v3.hip)
    uint16_t *wstream_x3 = nullptr;     // three bf16 parts per weight (mlp_kernel_bf16x3.hip)
    uint16_t *wstream_x2 = nullptr;     // two f16 parts per weight (mlp_kernel_f16x2.hip); NULL if a weight exceeds the f16 range
    uint16_t *wstream_f16v2 = nullptr;  // one f16 per weight in wstream_bf16v2's order (mlp_kernel_f16v2.hip: certify_zero's pre-filter); NULL likewise
    bool loaded = false;
};

// certify_zero: a sample is a certain zero iff its 16-bit (pre-filter) density pre-activation is below -margin.  Floors per network WITH THE bf16
// PRE-FILTER (networks beyond the f16 range), set by the statistic
// the audit watches -- the largest |bf16 - exact| on an audited certificate, which widens the margin above half of it: lego coarse 0.15-0.20
// (a fifth of 1.0), fine 0.99-1.24 (a third to 0.41 of 3.0; at round 3's fine margin of 2 it would trip the rule on most frames).  Round 3's
// fuzz without an audit: coarse 0.5 / fine 1.0 never differed in 10 031 frames, 0.25 / 0.5 did in 1 % of them.  What the floors cost in work
// (C3 frame, exact evaluations: tools/sweep_certify_margins.py): coarse 0.5 -> 8.2 %, 1.0 -> 12.2 %, 1.5 -> 18.6 % of the samples; fine 2 -> 18.8 %,
// 3 -> 20.7 %, 4 -> 23.3 %.
constexpr float kCertMarginCoarse = 1.0f, kCertMarginFine = 3.0f;
// The same floors when the pre-filter runs in f16 (mlp_kernel_f16v2.hip; the default wherever the network's weights fit the f16 range): its
// pre-activations are 8 x closer to the exact ones (numpy emulation on lego rays: largest |f16 - f32| on true zeros 0.048 coarse / 0.113 fine
// against bf16's 0.20 / 0.87), so the margins can be tighter by about that factor with the same distance between floor and largest audited error
// (audited errors on lego: 0.03-0.055 / 0.12-0.17; floor sweep: profiles/r04_certify_f16_prefilter_margin_sweep.log -- fine 0.3 trips the half-margin rule).
constexpr float kCertMarginCoarseF16 = 0.25f, kCertMarginFineF16 = 0.5f;

struct EvPair {
    hipEvent_t a, b;
    int kind; // 0 coarse mlp, 1 fine mlp (dominant), 2 other
    uint64_t points;
};

} // namespace nerfint

struct nerf_ctx {
    int device = 0;
    int n_cus = 0;
    std::string arch;
    std::string err;
    hipStream_t stream = nullptr; // used by the host-pointer entry points
    nerfint::DevNet net[2];
    // pass workspace
    size_t ws_rays = 0, ws_nc = 0, ws_m = 0;
    float *d_dirs = nullptr, *d_tc = nullptr, *d_sc = nullptr, *d_rgbc = nullptr, *d_tf = nullptr, *d_sf = nullptr,
          *d_rgbf = nullptr;
    float *d_rayfb = nullptr; size_t rayfb_floats = 0; // SSAA ray framebuffer
    float *d_out = nullptr; size_t out_floats = 0;       // host-pointer render output staging
    // scratch for forward_batch / stage calls
    void *d_scratch = nullptr; size_t scratch_bytes = 0;
    // skip_dead: device queue/counters {u32 ray counter, u32 live count, u64 chunk count} per MLP launch of a render, the
    // compacted trunk outputs of the live samples and their sample indices
    unsigned int *d_seq = nullptr; size_t seq_slots = 0;
    float *d_h8 = nullptr; size_t h8_bytes = 0;
    unsigned int *d_slot_point = nullptr; size_t slot_point_bytes = 0;
    unsigned int *d_flag_list = nullptr; size_t flag_list_bytes = 0; // hybrid sampling: rays whose coarse pass is redone in f32
    // zero certification (nerf_render_opts.certify_zero; nerf_api.cpp cert_pass, sampling_kernels.hip k_cert_*)
    unsigned int *d_point_list = nullptr; size_t point_list_bytes = 0; // samples the exact kernel evaluates (one list, reused by every launch)
    unsigned int *d_cert = nullptr; size_t cert_bytes = 0;             // counters, 8 per (pass, network)
    int *d_jstar = nullptr; size_t jstar_bytes = 0;                    // per ray of a pass: first sample behind the predicted cut
    unsigned int *d_cert_aux = nullptr; size_t cert_aux_bytes = 0;     // {sample, 16-bit (pre-filter) pre-activation} of the audited certificates of a launch
    float cert_margin[2] = {nerfint::kCertMarginCoarse, nerfint::kCertMarginFine}; // widened by render_device when an audit fails; reset at load
    float cert_margin_floor[2] = {nerfint::kCertMarginCoarse, nerfint::kCertMarginFine};
    float cert_margin_floor_f16[2] = {nerfint::kCertMarginCoarseF16, nerfint::kCertMarginFineF16};
    bool cert_prefilter_f16[2] = {false, false}; // per network: the pre-filter runs in f16 (set at load when the weights fit; cleared for good when an activation left the f16 range)
    bool cert_allow_f16 = true;
    float cert_depth_limit = 9.6f;        // predicted cut: bf16 optical depth > 9.6 (the exact cut is at T < 1e-4 = depth 9.21; nothing but work depends on it:
                                          // lego frame 0 rays fall back at 9.5, 25 at 9.35, 1132 of 640 000 at 9.25 -- tools/sweep_certify.py)
    unsigned cert_audit_mask = 127;       // one certified sample in 128 is audited ...
    unsigned cert_audit_mask_near = 15;   // ... and one in 16 of those certified by less than twice the margin: the same number of audits as a flat 1 in 64
                                          // on the lego frame, four times as many where a certificate is at risk
    bool cert_zero_tiles = true;          // probable zeros + audited certificates in the list's back part, evaluated with skip_empty
    float cert_zero_frac = 0.1f;          // "probably zero": pre-filter pre-activation below -margin x this (C3 frame: 1/2 -> 334.6 ms, 1/3 -> 333.1, 0.1 -> 330.5, 0.02 -> 331.0: tools/sweep_certify_zero_frac.py)
    bool cert_seq_prefilter = true;       // 16-bit pre-filter ray-sequential with its own predicted cut (false: the fused 16-bit kernel over all samples)
    double cert_list_frac = 0.5;          // list capacity as a fraction of a pass's samples: what earlier frames needed + 25 %
    float hybrid_tau = 1e-5f;                                        // a draw predicted to move by more than this (in t) flags its ray
    size_t max_export_bytes = (size_t)16 << 30; // budget of d_h8: bounds the rays per pass of skip_dead in a SPLIT arithmetic (NERF_MAX_EXPORT_BYTES);
                                                // 16 GiB = 8 passes per 800x800 frame, 1.4 % slower than one pass of 128 GiB
    unsigned long long *d_skip = nullptr;  // device counter of skipped 128-point tiles (skip_empty)
    unsigned int *d_nonfinite = nullptr;   // device counter of non-finite density pre-activations (split arithmetics)
    unsigned long long *d_clock = nullptr; // diagnostic: per-workgroup {cycles, 100 MHz ticks} of the last fine-MLP launch
    bool clock_valid = false;
    size_t max_rays_per_pass = (size_t)1 << 20;
    std::vector<hipEvent_t> ev_pool;
    std::vector<nerfint::EvPair> last_render; // events of the last render
    std::vector<nerfint::EvPair> dominant;    // accumulated dominant-kernel events (nerf_kernel_time_query)
};

namespace nerfint {

// records the message on the context (if any) and for nerf_last_error(NULL) of this thread; returns `code`
int fail(nerf_ctx *c, int code, const std::string &msg);

#define HIP_TRY(c, expr)                                                                                    \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail((c), NERF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)

struct DeviceGuard { // a context is bound to one device; entry points may be called with another one current
    int prev = -1;
    bool ok;
    explicit DeviceGuard(int dev) {
        ok = hipGetDevice(&prev) == hipSuccess;
        if (ok && prev != dev) ok = hipSetDevice(dev) == hipSuccess; else if (ok) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Rows of band i of n of a window of h rows (nerf_render_opts.band_*): stripe == 0 contiguous bands, stripe > 0 stripes round-robin.
inline int band_rows(int h, int i, int n, int stripe) {
    if (n <= 1) return h;
    if (stripe <= 0) return h / n + (i < h % n ? 1 : 0);
    const int full = h / stripe, tail = h % stripe;
    return (full / n + (i < full % n ? 1 : 0)) * stripe + (tail > 0 && full % n == i ? tail : 0);
}
inline int band_first_row(int h, int i, int n) { return i * (h / n) + std::min(i, h % n); } // contiguous bands only

int ensure_bytes(nerf_ctx *c, void **p, size_t *cur, size_t need);
// render_image on the context's device, asynchronous on `st` (synchronises only when stats != NULL)
int render_device(nerf_ctx *c, const nerf_camera *cam, const nerf_render_opts *o, float *d_out, hipStream_t st,
                  nerf_stats *stats);

} // namespace nerfint
