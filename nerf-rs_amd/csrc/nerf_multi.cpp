// nerf_multi.cpp -- render_image over several GPUs of one node, behind the C ABI (no torch, no Python).
//
// The reference fans render_block out over rayon workers and scatters the blocks into one image
// (src/lib.rs:533-557); every ray is independent.  Here the fan-out is over GPUs: context i (one per device) renders band i of
// the pixel rows -- its own host thread, its own stream, weights replicated -- and the bands meet in ONE framebuffer.
// The partition (nerf_render_opts.band_*): rayon balances by work stealing; a static partition has to know where the cost is.
//   * every ray costs the same (plain renders): CONTIGUOUS bands, the first rows % n one row longer;
//   * the cost follows the scene (skip_dead, skip_empty, certify_zero: 75 % of the lego rays are background and nearly free there,
//     and the background sits in the top rows): single rows dealt out ROUND-ROBIN -- neighbouring rows cost the same, so every
//     band gets the same mix (tools/band_balance.py measures both partitions on one GPU).  The bands then arrive packed and a copy
//     kernel (k_bands_to_frame) puts the rows where they belong; still ONE gather.
// Three ways to bring the bands together (nerf_gather):
//   HOST  every band goes device -> host straight into its rows of the caller's buffer (no GPU-to-GPU traffic);
//   PEER  bands are copied GPU -> GPU over xGMI (hipMemcpyPeerAsync) into a frame on ctxs[0]'s device, one D2H from there;
//   RCCL  one ncclAllGather of the bands over xGMI leaves the whole frame on EVERY device (the north-star's "RCCL gather
//         of the final framebuffer"); librccl is dlopen'ed on first use, so single-GPU hosts do not depend on it.  Contexts that
//         share a device (test boxes) rehearse the same slot layout with device-to-device copies instead of the collective.
// The per-pixel counter RNG makes a band bit-identical to the same rows of a single-context frame.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "nerf_internal.h"
#include "sampling_kernels.h"

using namespace nerfint;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms; // one clique per ordered device list
    std::string err;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

bool rccl_load(std::string &err) { // g_rccl_mu held
    Rccl &R = g_rccl;
    if (R.lib) return true;
    if (!R.err.empty()) { err = R.err; return false; }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (R.lib) break;
    }
    if (!R.lib) { err = R.err = std::string("librccl.so not found: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { void *p = dlsym(R.lib, n); if (!p && R.err.empty()) R.err = std::string("librccl: missing symbol ") + n; return p; };
    R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!R.err.empty()) { err = R.err; dlclose(R.lib); R.lib = nullptr; return false; }
    return true;
}

struct Job {
    nerf_ctx *c;
    nerf_render_opts o;  // this context's band as a crop window
    float *d_band;       // where the band is rendered (device memory of c->device)
    size_t band_floats;
    size_t frame_off;    // float offset of the band inside the frame (contiguous bands)
    int rows = 0;        // rows of the band
    nerf_stats *stats;
    int rc = NERF_OK;
};

} // namespace

extern "C" {

int nerf_create_multi(const int *device_ids, int n, nerf_ctx **out) {
    if (!out || n <= 0) return fail(nullptr, NERF_ERR_INVALID, "nerf_create_multi: out is NULL or n <= 0");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        const int rc = nerf_create(device_ids ? device_ids[i] : i, &out[i]);
        if (rc) { // nerf_create left its message in nerf_last_error(NULL)
            for (int j = 0; j < i; ++j) { nerf_destroy(out[j]); out[j] = nullptr; }
            return rc;
        }
    }
    return NERF_OK;
}

void nerf_multi_release(void) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    for (auto &kv : g_rccl.comms)
        for (ncclComm_t cm : kv.second) if (cm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm);
    g_rccl.comms.clear();
}

int nerf_render_image_multi(nerf_ctx *const *ctxs, int n, const nerf_camera *cam, const nerf_render_opts *opts, int gather,
                            float *rgb_out, nerf_stats *per_ctx) try {
    if (!ctxs || n <= 0) return fail(nullptr, NERF_ERR_INVALID, "nerf_render_image_multi: no contexts");
    std::set<const nerf_ctx *> seen;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(nullptr, NERF_ERR_INVALID, "nerf_render_image_multi: a context is NULL");
        if (!seen.insert(ctxs[i]).second) return fail(ctxs[0], NERF_ERR_INVALID, "nerf_render_image_multi: the same context is listed twice (a context is single-caller)");
    }
    nerf_ctx *c0 = ctxs[0];
    if (!cam || cam->nx <= 0 || cam->ny <= 0) return fail(c0, NERF_ERR_INVALID, "width and height must be greater than zero");
    if (!opts) return fail(c0, NERF_ERR_INVALID, "opts is NULL");
    if (!rgb_out) return fail(c0, NERF_ERR_INVALID, "output pointer is NULL");
    if (gather != NERF_GATHER_HOST && gather != NERF_GATHER_PEER && gather != NERF_GATHER_RCCL)
        return fail(c0, NERF_ERR_INVALID, "gather must be NERF_GATHER_HOST, NERF_GATHER_PEER or NERF_GATHER_RCCL");
    int x0 = 0, y0 = 0, cw = cam->nx, ch = cam->ny;
    if (opts->crop_w > 0 || opts->crop_h > 0) { x0 = opts->crop_x0; y0 = opts->crop_y0; cw = opts->crop_w; ch = opts->crop_h; }
    if (cw <= 0 || ch <= 0 || x0 < 0 || y0 < 0 || x0 + cw > cam->nx || y0 + ch > cam->ny)
        return fail(c0, NERF_ERR_INVALID, "crop window outside the frame");
    const size_t row_floats = (size_t)cw * 3, frame_floats = row_floats * ch;
    // cost follows the scene => rows round-robin (stripes of one row); uniform cost => contiguous bands
    const int stripe = (n > 1 && (opts->skip_dead || opts->skip_empty || opts->certify_zero)) ? 1 : 0;
    const int max_rows = band_rows(ch, 0, n, stripe);   // band 0 is never shorter than another band
    const size_t slot_floats = row_floats * max_rows;   // RCCL / striped PEER: equal-sized slots, the ragged tail of a slot is unused

    // RCCL refuses two ranks on one device.  When contexts SHARE a device (single-GPU test boxes) the collective step of the RCCL
    // path is rehearsed as device-to-device copies into the same equal-slot buffers, so the slot layout, the stream ordering and the
    // ragged compaction below are exercised; RCCL proper runs whenever the devices are distinct.
    std::vector<ncclComm_t> comms;
    bool loopback = false;
    if (gather == NERF_GATHER_RCCL) {
        std::vector<int> devs;
        for (int i = 0; i < n; ++i) devs.push_back(ctxs[i]->device);
        loopback = std::set<int>(devs.begin(), devs.end()).size() != devs.size();
        if (!loopback) {
            std::lock_guard<std::mutex> lk(g_rccl_mu);
            std::string err;
            if (!rccl_load(err)) return fail(c0, NERF_ERR_STATE, err);
            auto it = g_rccl.comms.find(devs);
            if (it == g_rccl.comms.end()) {
                std::vector<ncclComm_t> cm(n, nullptr);
                const ncclResult_t r = g_rccl.CommInitAll(cm.data(), n, devs.data());
                if (r != ncclSuccess) return fail(c0, NERF_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
                it = g_rccl.comms.emplace(devs, cm).first;
            }
            comms = it->second;
        }
    }

    // Destination buffers.  HOST: each context's d_out holds its band.  PEER: ctxs[0]'s d_out holds the frame (its own band is
    // rendered in place), the others hold their band -- striped: ctxs[0]'s d_out = n slots + the frame.  RCCL: every d_out = n slots
    // (the in-place all-gather buffer), then the frame is assembled behind them when the bands are ragged or striped.
    const bool ragged = (ch % n) != 0 || stripe > 0;
    std::vector<Job> jobs(n);
    for (int i = 0; i < n; ++i) {
        nerf_ctx *c = ctxs[i];
        DeviceGuard dg(c->device);
        const int rows = band_rows(ch, i, n, stripe), b0 = stripe ? 0 : band_first_row(ch, i, n);
        size_t need = row_floats * (size_t)std::max(rows, 1);
        if (gather == NERF_GATHER_PEER && i == 0) need = stripe ? slot_floats * n + frame_floats : frame_floats;
        if (gather == NERF_GATHER_RCCL) need = slot_floats * n + (ragged ? frame_floats : 0);
        int rc;
        if ((rc = ensure_bytes(c, (void **)&c->d_out, &c->out_floats, need * sizeof(float)))) return rc;
        Job &J = jobs[i];
        J.c = c; J.o = *opts;
        J.o.crop_x0 = x0; J.o.crop_y0 = y0; J.o.crop_w = cw; J.o.crop_h = ch; // the caller's window; the band is selected by band_*
        J.o.band_index = i; J.o.band_count = n; J.o.band_stripe_rows = stripe;
        J.rows = rows;
        J.band_floats = row_floats * rows;
        J.frame_off = row_floats * b0;                                          // contiguous bands only
        J.d_band = c->d_out;
        if (gather == NERF_GATHER_PEER && i == 0) J.d_band = c->d_out + J.frame_off; // b0 == 0: the frame (contiguous) or slot 0 (striped)
        if (gather == NERF_GATHER_RCCL) J.d_band = c->d_out + slot_floats * i;
        J.stats = per_ctx ? &per_ctx[i] : nullptr;
        if (per_ctx) memset(&per_ctx[i], 0, sizeof(nerf_stats));
    }

    // Fan out: one host thread per context (HIP's current device is per thread).  A thread enqueues its band's kernels on
    // its context's stream, then its share of the gather, and waits for its own stream only.
    float *d_frame0 = c0->d_out;
    const int dev0 = c0->device;
    auto work_body = [&](int i) {
        Job &J = jobs[i];
        nerf_ctx *c = J.c;
        if (hipSetDevice(c->device) != hipSuccess) { J.rc = fail(c, NERF_ERR_HIP, "hipSetDevice failed"); return; }
        if (J.rows > 0) {
            J.rc = render_device(c, cam, &J.o, J.d_band, c->stream, J.stats);
            if (J.rc) return;
        }
        hipError_t e = hipSuccess;
        if (gather == NERF_GATHER_HOST && J.band_floats) {
            if (!stripe) e = hipMemcpyAsync(rgb_out + J.frame_off, J.d_band, J.band_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            else { // packed stripes -> every n-th stripe of the caller's frame: ONE strided copy (+ the frame's last, shorter stripe if it is ours)
                const size_t sb = (size_t)stripe * row_floats * sizeof(float);
                const int full = J.rows / stripe, tail = J.rows % stripe;
                if (full) e = hipMemcpy2DAsync(rgb_out + (size_t)i * stripe * row_floats, sb * n, J.d_band, sb, sb, full, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess && tail)
                    e = hipMemcpyAsync(rgb_out + ((size_t)full * n + i) * stripe * row_floats, J.d_band + (size_t)full * stripe * row_floats,
                                       (size_t)tail * row_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            }
        } else if (gather == NERF_GATHER_PEER && i != 0 && J.band_floats) {
            if (c->device != dev0) { // direct xGMI writes when the devices are peers (otherwise HIP stages the copy)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->device, dev0) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(dev0, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
            // contiguous: straight to the band's place in the frame; striped: to slot i, the copy kernel below assembles the frame
            e = hipMemcpyPeerAsync(d_frame0 + (stripe ? slot_floats * i : J.frame_off), dev0, J.d_band, c->device, J.band_floats * sizeof(float), c->stream);
        }
        if (e == hipSuccess && gather != NERF_GATHER_RCCL) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) J.rc = fail(c, NERF_ERR_HIP, std::string("band gather: ") + hipGetErrorString(e));
    };
    auto work = [&](int i) { // nothing may escape a thread (std::terminate) or the C ABI
        try { work_body(i); }
        catch (const std::exception &e) { jobs[i].rc = fail(jobs[i].c, NERF_ERR_INVALID, std::string("internal error: ") + e.what()); }
        catch (...) { jobs[i].rc = fail(jobs[i].c, NERF_ERR_INVALID, "internal error"); }
    };
    if (n == 1) {
        DeviceGuard dg(c0->device);
        work(0);
    } else {
        struct Joiner { // joins whatever was started, also when starting a later thread throws
            std::vector<std::thread> th;
            ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
        } pool;
        pool.th.reserve(n);
        for (int i = 0; i < n; ++i) pool.th.emplace_back(work, i);
    }
    for (int i = 0; i < n; ++i)
        if (jobs[i].rc) {
            if (gather == NERF_GATHER_RCCL) // the other bands' kernels are still in flight (that path synchronises after the collective)
                for (int k = 0; k < n; ++k) { DeviceGuard dg(ctxs[k]->device); (void)hipStreamSynchronize(ctxs[k]->stream); }
            if (ctxs[i] != c0) c0->err = ctxs[i]->err;
            return fail(c0, jobs[i].rc, c0->err);
        }

    if (gather == NERF_GATHER_RCCL) {
        // ONE collective: every rank contributes its slot, every device receives all slots (in place).  The renders above are
        // already enqueued on the same streams, so stream order is the only synchronisation needed.
        // From here on every context's stream carries work that reads other contexts' buffers: whatever fails, ALL streams are
        // synchronised before this function returns (first error wins).
        int rc_first = NERF_OK;
        std::string err_first;
        auto note = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && rc_first == NERF_OK) { rc_first = NERF_ERR_HIP; err_first = std::string(what) + ": " + hipGetErrorString(e); }
            return e == hipSuccess;
        };
        std::vector<hipEvent_t> rendered;
        if (loopback) {
            // the same data movement as the in-place all-gather: slot k of context k -> slot k of every other context
            rendered.assign(n, nullptr);
            for (int i = 0; i < n; ++i) {
                DeviceGuard dg(ctxs[i]->device);
                if (note(hipEventCreateWithFlags(&rendered[i], hipEventDisableTiming), "hipEventCreate")) note(hipEventRecord(rendered[i], ctxs[i]->stream), "hipEventRecord");
            }
            for (int i = 0; i < n && rc_first == NERF_OK; ++i) {
                nerf_ctx *c = ctxs[i];
                DeviceGuard dg(c->device);
                for (int k = 0; k < n && rc_first == NERF_OK; ++k) {
                    if (k == i) continue;
                    if (!note(hipStreamWaitEvent(c->stream, rendered[k], 0), "hipStreamWaitEvent")) break;
                    const float *src = ctxs[k]->d_out + slot_floats * k;
                    float *dst = c->d_out + slot_floats * k;
                    note(ctxs[k]->device == c->device
                             ? hipMemcpyAsync(dst, src, slot_floats * sizeof(float), hipMemcpyDeviceToDevice, c->stream)
                             : hipMemcpyPeerAsync(dst, c->device, src, ctxs[k]->device, slot_floats * sizeof(float), c->stream),
                         "all-gather rehearsal copy");
                }
            }
        } else {
            ncclResult_t r = g_rccl.GroupStart();
            for (int i = 0; i < n && r == ncclSuccess; ++i)
                r = g_rccl.AllGather(jobs[i].d_band, ctxs[i]->d_out, slot_floats, ncclFloat, comms[i], ctxs[i]->stream);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
            if (r != ncclSuccess) { rc_first = NERF_ERR_HIP; err_first = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); }
        }
        for (int i = 0; i < n; ++i) {
            nerf_ctx *c = ctxs[i];
            DeviceGuard dg(c->device);
            if (stripe && rc_first == NERF_OK) // slots of packed rows -> frame behind them (every device ends up with the whole frame)
                note(launch_bands_to_frame(c->d_out, c->d_out + slot_floats * n, cw, ch, n, stripe, slot_floats, c->stream), "bands -> frame");
            else if (ragged && rc_first == NERF_OK) { // slots -> contiguous frame behind them (every device ends up with the whole frame)
                float *frame = c->d_out + slot_floats * n;
                for (int k = 0; k < n; ++k)
                    if (jobs[k].band_floats)
                        note(hipMemcpyAsync(frame + jobs[k].frame_off, c->d_out + slot_floats * k, jobs[k].band_floats * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "slot compaction");
            }
        }
        d_frame0 = ragged ? c0->d_out + slot_floats * n : c0->d_out;
        if (rc_first == NERF_OK) {
            DeviceGuard dg(c0->device);
            note(hipMemcpyAsync(rgb_out, d_frame0, frame_floats * sizeof(float), hipMemcpyDeviceToHost, c0->stream), "frame D2H");
        }
        for (int i = 0; i < n; ++i) { // every stream, on every path: a context's buffers are read by the other contexts' streams
            DeviceGuard dg(ctxs[i]->device);
            note(hipStreamSynchronize(ctxs[i]->stream), "hipStreamSynchronize");
        }
        for (hipEvent_t e : rendered) if (e) (void)hipEventDestroy(e);
        return rc_first == NERF_OK ? NERF_OK : fail(c0, rc_first, err_first);
    }
    if (gather == NERF_GATHER_PEER) {
        DeviceGuard dg(c0->device);
        if (stripe) { // every band has arrived in its slot (the threads synchronised their streams): rows -> their places
            HIP_TRY(c0, launch_bands_to_frame(c0->d_out, c0->d_out + slot_floats * n, cw, ch, n, stripe, slot_floats, c0->stream));
            d_frame0 = c0->d_out + slot_floats * n;
        }
        HIP_TRY(c0, hipMemcpyAsync(rgb_out, d_frame0, frame_floats * sizeof(float), hipMemcpyDeviceToHost, c0->stream));
        HIP_TRY(c0, hipStreamSynchronize(c0->stream));
    }
    return NERF_OK;
} catch (const std::exception &e) {
    return fail(ctxs && n > 0 ? ctxs[0] : nullptr, NERF_ERR_INVALID, std::string("nerf_render_image_multi: ") + e.what());
}

} // extern "C"
