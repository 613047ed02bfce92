// sampling_kernels.h -- launch interface of the sampling / integration kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// A pass renders the rectangle [ry0, ry0+rh) x [rx0, rx0+rw) of the (rny x rnx) ray grid; ray r of the pass is
// (ry0 + r / rw, rx0 + r % rw); its RNG pixel index is row * rnx + col.
struct RayGenArgs {
    int n_rays;
    int rx0, ry0, rw, rnx, rny;
    float half;      // 0.5: pixel centre (src/lib.rs:221-222)
    int normalize;   // 1: dir.normalize() (src/lib.rs:371)
    float sx, sy;    // tan(alpha_width), tan(alpha_height)
    float r[3], u[3], f[3]; // orthonormal basis (src/lib.rs:216-218), computed on the host
    // striped bands (nerf_render_opts.band_*): stripe > 0 => ry0 counts BAND-LOCAL rows; band-local row j is ray-grid row
    // stripe_y0 + ((j / stripe) * stripe_n + stripe_i) * stripe + j % stripe.  stripe == 0: ry0 is the ray-grid row itself.
    int stripe, stripe_n, stripe_i, stripe_y0;
};

struct ResampleArgs {
    RayGenArgs g;
    int n_rays, nc, nf;
    float far_;
    uint32_t seed_lo, seed_hi;
    const float *t_coarse;     // n_rays x nc
    const float *sigma_coarse; // n_rays x nc
    float *t_fine;             // n_rays x (nc + nf), ascending
    // stage-test hooks (all optional)
    const uint32_t *pixel_index; // n_rays: overrides the rectangle-derived pixel index
    const float *u_in;           // n_rays x nf: overrides the Philox uniforms
    float *w_out;                // n_rays x nc
    float *cdf_out;              // n_rays x (nc - 1)
    float *t_new_out;            // n_rays x nf (unsorted draws)
    int sort_pow2 = 0;           // set by launch_resample: smallest power of two >= nc + nf (bitonic sort width)
    // hybrid sampling (nerf_render_opts.hybrid_sampling): flag the rays with a draw whose position is predicted to move by more
    // than flag_tau (in t) under the split arithmetics' density error ...
    float flag_tau = 0.0f;
    unsigned int *flag_count = nullptr; // device counter (zeroed by the caller)
    unsigned int *flag_list = nullptr;  // flagged ray indices, appended in arbitrary order
    unsigned char *flag_out = nullptr;  // stage-test hook: n_rays bytes, 1 = this ray would be flagged (same decision, dense)
    // ... and, in a second launch, resample only the rays of a list (n_rays = capacity of the launch; the count lives on the device)
    const unsigned int *ray_list = nullptr;
    const unsigned int *ray_list_count = nullptr;
};

struct CompositeArgs {
    int n_rays, n;
    float far_;
    const float *t;     // n_rays x n
    const float *sigma; // n_rays x n
    const float *rgb;   // n_rays x n x 3
    float *out;         // n_rays x 3
    float *w_out;       // optional n_rays x n
};

hipError_t sampling_init(void);
hipError_t launch_ray_dirs(const RayGenArgs &a, float *dirs, hipStream_t st);
hipError_t launch_stratified(const RayGenArgs &a, int count, float near_, float far_, uint64_t seed, float *t,
                             hipStream_t st);
hipError_t launch_resample(const ResampleArgs &a, hipStream_t st);
hipError_t launch_composite(const CompositeArgs &a, hipStream_t st);
hipError_t launch_box_downsample(const float *rays, float *out, int w, int h, int s, hipStream_t st);
// multi-GPU: n gathered bands (slot_floats apart, rows packed) -> the h x w x 3 frame; stripe = 0: contiguous bands
hipError_t launch_bands_to_frame(const float *slots, float *frame, int w, int h, int n, int stripe, size_t slot_floats, hipStream_t st);
size_t resample_lds_bytes(int nc, int nf);
size_t composite_lds_bytes(int n);
// ---- zero certification (nerf_render_opts.certify_zero; kernels and protocol: sampling_kernels.hip) ------------------------------------
struct CertPlanArgs {
    float *pre;          // in: the bf16 kernel's density pre-activations (n_rays x spr); out: 0, or a mark on uncertain samples behind j*
    const float *t;      // n_rays x spr, ascending
    int n_rays, spr;
    float far_;
    float margin;        // a sample is certified (density 0) iff pre < -margin
    float depth_limit;   // predicted cut: the first sample with bf16 optical depth in front of it > depth_limit
    unsigned audit_mask; // 2^k - 1: one certified sample in 2^k is audited ...
    unsigned audit_mask_near; // ... and one in 2^j of those certified by less than twice the margin (pre > -2 margin)
    unsigned audit_salt;
    unsigned *list;      // phase-1 list: sample index | (audit ? 1 << 31 : 0)
    unsigned *count;     // += entries (also those beyond capacity)
    unsigned capacity;
    // optional back part of the list (entry k at capacity - 1 - k): the listed samples that are PROBABLY zeros -- pre-activation below
    // -zero_threshold -- and the audited certificates; count_back = NULL: everything goes to the front part
    unsigned *count_back;
    float zero_threshold;
    unsigned *aux;       // {sample index, bits of its bf16 pre-activation} of every audited certificate
    unsigned *aux_count;
    unsigned aux_capacity;
    int *jstar;          // per ray: first sample behind the predicted cut (spr: none)
    int rays_per_wave = 0; // set by launch_cert_plan
};
struct CertVerifyArgs {
    const float *t;
    float *sigma;        // the exact pass's density buffer (marks are cleared)
    int n_rays, spr;
    float far_;
    const int *jstar;
    unsigned *list;      // phase-2 list (no audit entries)
    unsigned *count;
    unsigned capacity;
    unsigned *fallback_rays; // optional: += rays whose predicted cut was not confirmed
};
hipError_t launch_cert_plan(const CertPlanArgs &a, hipStream_t st);
hipError_t launch_cert_verify(const CertVerifyArgs &a, hipStream_t st);
// audit[0] += audited samples, audit[1] += violations, audit[2] = max(0x7f800000 - bits(min headroom)), audit[3] = max bits(|bf16 - exact|);
// resets the audited entries of `sigma` to 0
hipError_t launch_cert_audit(const unsigned *aux, const unsigned *aux_count, unsigned aux_capacity, float *sigma, unsigned *audit, int n_cus, hipStream_t st);
size_t cert_plan_lds_bytes(int spr, int *rays_per_wave);
