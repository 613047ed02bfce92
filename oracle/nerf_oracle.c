/*
 * nerf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See nerf_oracle.h.
 *
 * Every function cites the reference lines it restates (paths relative to the reference repo
 * elisabeth96/nerf-rs).  Arithmetic is IEEE f32 with NO fused multiply-add: build with
 * -ffp-contract=off (oracle/Makefile) so `a + b * c` rounds twice exactly like rustc's output.
 */
#include "nerf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Network container.  Kernels are [in x out] row-major (src/network.rs:135,139;
 * lego_rust/README.md:23-26).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int K, N;
    float *w; /* K*N */
    float *b; /* N */
} o_layer;

struct oracle_net {
    o_layer dense[8];
    o_layer bottleneck, viewdirs, rgb, alpha;
};

enum { ACT_RELU = 0, ACT_SIGMOID = 1, ACT_NONE = 2 }; /* src/network.rs:67-72 */

/* ---- loader: src/lib.rs:34-42 (load_tensor), :62-74 (load_shapes), :108-174 (assemble by name) ---- */
typedef struct {
    char name[64];
    int ndims;
    size_t dims[4];
    float *data;
    size_t len;
} o_param;

static float *read_f32_file(const char *path, size_t *n_out) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    size_t n = (size_t)sz / 4; /* chunks_exact(4): trailing bytes dropped (src/lib.rs:38) */
    float *d = (float *)malloc(n ? n * 4 : 4);
    if (n && fread(d, 4, n, f) != n) { fclose(f); free(d); return NULL; }
    fclose(f);
    *n_out = n;
    return d; /* host is little-endian (x86-64); from_le_bytes is the identity */
}

static int take(o_param *params, int np, const char *name, int want_dims, o_layer *L, int is_bias, char *err,
                size_t errlen) {
    for (int i = 0; i < np; ++i) {
        if (params[i].data && strcmp(params[i].name, name) == 0) {
            if (params[i].ndims != want_dims) {
                snprintf(err, errlen, "%s dims mismatch for %s", is_bias ? "bias" : "matrix", name);
                return -1;
            }
            if (is_bias) {
                if (params[i].dims[0] != params[i].len) { snprintf(err, errlen, "bias size mismatch for %s", name); return -1; }
                L->b = params[i].data;
            } else {
                if (params[i].dims[0] * params[i].dims[1] != params[i].len) { snprintf(err, errlen, "matrix size mismatch for %s", name); return -1; }
                L->K = (int)params[i].dims[0];
                L->N = (int)params[i].dims[1];
                L->w = params[i].data;
            }
            params[i].data = NULL; /* params.remove(name) */
            return 0;
        }
    }
    snprintf(err, errlen, "missing %s parameter: %s", is_bias ? "bias" : "matrix", name); /* src/lib.rs:118,127 */
    return -1;
}

oracle_net *oracle_net_load_dir(const char *dir, char *err, size_t errlen) {
    char path[1024];
    char ebuf[256];
    if (!err) { err = ebuf; errlen = sizeof ebuf; }
    err[0] = 0;
    snprintf(path, sizeof path, "%s/shapes.txt", dir);
    FILE *f = fopen(path, "r");
    if (!f) { snprintf(err, errlen, "read shapes: %s", path); return NULL; }
    o_param params[64];
    int np = 0;
    char line[512];
    while (fgets(line, sizeof line, f) && np < 64) {
        char *tok = strtok(line, " \t\r\n");
        if (!tok) continue;
        o_param *p = &params[np];
        memset(p, 0, sizeof *p);
        snprintf(p->name, sizeof p->name, "%s", tok);
        while ((tok = strtok(NULL, " \t\r\n")) && p->ndims < 4) p->dims[p->ndims++] = (size_t)strtoull(tok, NULL, 10);
        snprintf(path, sizeof path, "%s/%s.bin", dir, p->name);
        p->data = read_f32_file(path, &p->len);
        if (!p->data) {
            snprintf(err, errlen, "read tensor: %s", path);
            for (int i = 0; i < np; ++i) free(params[i].data);
            fclose(f);
            return NULL;
        }
        ++np;
    }
    fclose(f);
    oracle_net *net = (oracle_net *)calloc(1, sizeof *net);
    int bad = 0;
    char nm[64];
    for (int i = 0; i < 8 && !bad; ++i) { /* dense_specs, src/lib.rs:133-152 */
        snprintf(nm, sizeof nm, "dense%d_kernel", i);
        bad |= take(params, np, nm, 2, &net->dense[i], 0, err, errlen);
        if (bad) break;
        snprintf(nm, sizeof nm, "dense%d_bias", i);
        bad |= take(params, np, nm, 1, &net->dense[i], 1, err, errlen);
    }
    if (!bad) bad |= take(params, np, "bottleneck_kernel", 2, &net->bottleneck, 0, err, errlen);
    if (!bad) bad |= take(params, np, "bottleneck_bias", 1, &net->bottleneck, 1, err, errlen);
    if (!bad) bad |= take(params, np, "viewdirs_kernel", 2, &net->viewdirs, 0, err, errlen);
    if (!bad) bad |= take(params, np, "viewdirs_bias", 1, &net->viewdirs, 1, err, errlen);
    if (!bad) bad |= take(params, np, "rgb_kernel", 2, &net->rgb, 0, err, errlen);
    if (!bad) bad |= take(params, np, "rgb_bias", 1, &net->rgb, 1, err, errlen);
    if (!bad) bad |= take(params, np, "alpha_kernel", 2, &net->alpha, 0, err, errlen);
    if (!bad) bad |= take(params, np, "alpha_bias", 1, &net->alpha, 1, err, errlen);
    for (int i = 0; i < np; ++i) free(params[i].data); /* leftovers only trip a debug_assert (src/lib.rs:171) */
    if (bad) { oracle_net_free(net); return NULL; }
    return net;
}

static void free_layer(o_layer *L) { free(L->w); free(L->b); }

void oracle_net_free(oracle_net *net) {
    if (!net) return;
    for (int i = 0; i < 8; ++i) free_layer(&net->dense[i]);
    free_layer(&net->bottleneck); free_layer(&net->viewdirs); free_layer(&net->rgb); free_layer(&net->alpha);
    free(net);
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG (replaces rand::thread_rng at src/lib.rs:243,332,375,407).
 * Philox-4x32-10 (Salmon et al. 2011), key = seed, counter = (pixel_index, stream, k/4, 0),
 * word k%4.  f32 uniform has 23 random mantissa bits like rand 0.8.5's gen_range(0.0..1.0).
 * ---------------------------------------------------------------------------------------- */
void oracle_philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float oracle_uniform(uint64_t seed, uint32_t pixel_index, uint32_t stream, uint32_t k) {
    uint32_t o[4];
    oracle_philox4x32(seed, pixel_index, stream, k >> 2, 0u, o);
    return (float)(o[k & 3] >> 9) * (1.0f / 8388608.0f);
}

/* ------------------------------------------------------------------------------------------
 * MLP
 * ---------------------------------------------------------------------------------------- */
static inline float act_apply(float v, int act) { /* src/network.rs:161-169 */
    switch (act) {
        case ACT_RELU: return v > 0.0f ? v : 0.0f; /* f32::max(v, 0.0); NaN -> 0.0 like Rust's max */
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

/* forward_fallback, exact loop nest (src/network.rs:124-147): h is [K x B], out is [N x B]. */
static void layer_forward_naive(const o_layer *L, const float *h, size_t B, float *out, int act) {
    const int K = L->K, N = L->N;
    for (int n = 0; n < N; ++n) /* fill_with_bias :149-159 */
        for (size_t b = 0; b < B; ++b) out[(size_t)n * B + b] = L->b[n];
    for (int k = 0; k < K; ++k) {
        const float *wrow = L->w + (size_t)k * N;
        for (size_t b = 0; b < B; ++b) {
            const float hv = h[(size_t)k * B + b];
            for (int n = 0; n < N; ++n) {
                float cur = out[(size_t)n * B + b] + wrow[n] * hv; /* :139 mul then add */
                out[(size_t)n * B + b] = cur;
            }
        }
    }
    for (size_t i = 0; i < (size_t)N * B; ++i) out[i] = act_apply(out[i], act);
}

/* Same arithmetic per output element (acc = bias; for k ascending: acc = acc + w*h), cache-friendly
 * nest.  Bit-identical to layer_forward_naive because each (n,b) accumulates independently.
 * h has leading dimension ldh, out has leading dimension ldo, B <= block width. */
static void layer_forward_blocked(const o_layer *L, const float *h, size_t ldh, size_t B, float *out, size_t ldo,
                                  int act) {
    const int K = L->K, N = L->N;
    for (int n = 0; n < N; ++n) {
        float *o = out + (size_t)n * ldo;
        const float bias = L->b[n];
        for (size_t b = 0; b < B; ++b) o[b] = bias;
        for (int k = 0; k < K; ++k) {
            const float a = L->w[(size_t)k * N + n];
            const float *hr = h + (size_t)k * ldh;
            for (size_t b = 0; b < B; ++b) o[b] = o[b] + a * hr[b];
        }
        for (size_t b = 0; b < B; ++b) o[b] = act_apply(o[b], act);
    }
}

/* positional_encoding_batch / _dirs (src/network.rs:263-330): rows [x,y,z, sin(f x,y,z), cos(f x,y,z)]... f*=2 */
static void encode_cols(const float *x, const float *y, const float *z, size_t stride_in, size_t n, int octaves,
                        float *enc, size_t ld) {
    for (size_t c = 0; c < n; ++c) {
        const float v[3] = {x[c * stride_in], y[c * stride_in], z[c * stride_in]};
        enc[0 * ld + c] = v[0]; enc[1 * ld + c] = v[1]; enc[2 * ld + c] = v[2];
        float f = 1.0f;
        int row = 3;
        for (int o = 0; o < octaves; ++o) {
            for (int a = 0; a < 3; ++a) enc[(size_t)(row++) * ld + c] = sinf(f * v[a]);
            for (int a = 0; a < 3; ++a) enc[(size_t)(row++) * ld + c] = cosf(f * v[a]);
            f *= 2.0f;
        }
    }
}

void oracle_positional_encoding(const float *pts_soa, size_t n, int octaves, float *enc) {
    encode_cols(pts_soa, pts_soa + n, pts_soa + 2 * n, 1, n, octaves, enc, n);
}

#define OB 256 /* column block of the blocked path */

/* Network::forward_batch (src/network.rs:197-237). */
void oracle_forward_batch(const oracle_net *net, const float *pts, const float *dirs, size_t n, float *rgb,
                          float *sigma, int naive_order) {
    if (n == 0) return; /* :199-201 */
    if (naive_order) {
        float *h0 = (float *)malloc(sizeof(float) * 63 * n);
        float *a = (float *)malloc(sizeof(float) * 319 * n);
        float *b = (float *)malloc(sizeof(float) * 319 * n);
        float *sg = (float *)malloc(sizeof(float) * n);
        float *c3 = (float *)malloc(sizeof(float) * 3 * n);
        encode_cols(pts, pts + n, pts + 2 * n, 1, n, 10, h0, n);          /* :204 */
        layer_forward_naive(&net->dense[0], h0, n, a, ACT_RELU);           /* :206-208 */
        for (int i = 1; i < 5; ++i) { layer_forward_naive(&net->dense[i], a, n, b, ACT_RELU); float *t = a; a = b; b = t; }
        memcpy(b, h0, sizeof(float) * 63 * n);                              /* concat_rows(h_0, h4) :210 */
        memcpy(b + 63 * n, a, sizeof(float) * 256 * n);
        layer_forward_naive(&net->dense[5], b, n, a, ACT_RELU);
        for (int i = 6; i < 8; ++i) { layer_forward_naive(&net->dense[i], a, n, b, ACT_RELU); float *t = a; a = b; b = t; }
        layer_forward_naive(&net->alpha, a, n, sg, ACT_RELU);               /* :216 */
        layer_forward_naive(&net->bottleneck, a, n, b, ACT_NONE);           /* :218 */
        encode_cols(dirs, dirs + 1, dirs + 2, 3, n, 4, b + 256 * n, n);     /* :219-220 rows 256..282 */
        layer_forward_naive(&net->viewdirs, b, n, a, ACT_RELU);             /* :222 */
        layer_forward_naive(&net->rgb, a, n, c3, ACT_SIGMOID);              /* :223 */
        for (size_t c = 0; c < n; ++c) { rgb[3 * c] = c3[c]; rgb[3 * c + 1] = c3[n + c]; rgb[3 * c + 2] = c3[2 * n + c]; sigma[c] = sg[c]; }
        free(h0); free(a); free(b); free(sg); free(c3);
        return;
    }
    float *buf = (float *)malloc(sizeof(float) * (size_t)OB * (319 + 283 + 256 + 3 + 1));
    float *cat = buf;                 /* 319 x OB: rows 0..62 enc, 63..318 h4 */
    float *q = cat + 319 * OB;        /* 283 x OB: rows 0..255 bottleneck, 256..282 dir enc */
    float *h = q + 283 * OB;          /* 256 x OB */
    float *c3 = h + 256 * OB;         /* 3 x OB */
    float *sg = c3 + 3 * OB;          /* 1 x OB */
    for (size_t c0 = 0; c0 < n; c0 += OB) {
        const size_t B = (n - c0 < OB) ? n - c0 : OB;
        encode_cols(pts + c0, pts + n + c0, pts + 2 * n + c0, 1, B, 10, cat, OB);
        float *x = h, *y = cat + 63 * OB; /* ping-pong so that h4 lands in cat rows 63.. */
        layer_forward_blocked(&net->dense[0], cat, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[1], x, OB, B, y, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[2], y, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[3], x, OB, B, y, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[4], y, OB, B, x, OB, ACT_RELU);
        memcpy(y, x, sizeof(float) * 256 * OB); /* h4 -> cat rows 63..318 */
        layer_forward_blocked(&net->dense[5], cat, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[6], x, OB, B, q, OB, ACT_RELU);
        layer_forward_blocked(&net->dense[7], q, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->alpha, x, OB, B, sg, OB, ACT_RELU);
        layer_forward_blocked(&net->bottleneck, x, OB, B, q, OB, ACT_NONE);
        encode_cols(dirs + 3 * c0, dirs + 3 * c0 + 1, dirs + 3 * c0 + 2, 3, B, 4, q + 256 * OB, OB);
        layer_forward_blocked(&net->viewdirs, q, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->rgb, x, OB, B, c3, OB, ACT_SIGMOID);
        for (size_t c = 0; c < B; ++c) {
            rgb[3 * (c0 + c)] = c3[c]; rgb[3 * (c0 + c) + 1] = c3[OB + c]; rgb[3 * (c0 + c) + 2] = c3[2 * OB + c];
            sigma[c0 + c] = sg[c];
        }
    }
    free(buf);
}

/* ---- bf16-operand emulation (NOT in the reference): checks the arithmetic of the bf16 MFMA kernel (BASELINE config
 * C5).  Weights and every matrix-layer input (encodings, ReLU'd hidden activations, bottleneck output, direction
 * encodings) are rounded to bf16 (round-to-nearest-even); products accumulate in f32 from the bias, k ascending; the
 * alpha and rgb heads read the UNROUNDED f32 activations with f32 weights, exactly as the kernel's VALU heads do. */
static float bf16_round(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&v, &u, 4);
    return v;
}

static void round_rows(float *buf, int rows, size_t ld, size_t B) {
    for (int r = 0; r < rows; ++r)
        for (size_t b = 0; b < B; ++b) buf[(size_t)r * ld + b] = bf16_round(buf[(size_t)r * ld + b]);
}

static void layer_forward_bf16w(const o_layer *L, const float *h, size_t ldh, size_t B, float *out, size_t ldo, int act) {
    const int K = L->K, N = L->N;
    for (int n = 0; n < N; ++n) {
        float *o = out + (size_t)n * ldo;
        const float bias = L->b[n];
        for (size_t b = 0; b < B; ++b) o[b] = bias;
        for (int k = 0; k < K; ++k) {
            const float a = bf16_round(L->w[(size_t)k * N + n]);
            const float *hr = h + (size_t)k * ldh;
            for (size_t b = 0; b < B; ++b) o[b] = o[b] + a * hr[b]; /* bf16 x bf16 is exact in f32 */
        }
        for (size_t b = 0; b < B; ++b) o[b] = act_apply(o[b], act);
    }
}

void oracle_forward_batch_bf16(const oracle_net *net, const float *pts, const float *dirs, size_t n, float *rgb, float *sigma) {
    if (n == 0) return;
    float *buf = (float *)malloc(sizeof(float) * (size_t)OB * (319 + 283 + 256 + 3 + 1));
    float *cat = buf, *q = cat + 319 * OB, *h = q + 283 * OB, *c3 = h + 256 * OB, *sg = c3 + 3 * OB;
    for (size_t c0 = 0; c0 < n; c0 += OB) {
        const size_t B = (n - c0 < OB) ? n - c0 : OB;
        encode_cols(pts + c0, pts + n + c0, pts + 2 * n + c0, 1, B, 10, cat, OB);
        round_rows(cat, 63, OB, B);
        float *x = h, *y = cat + 63 * OB;
        layer_forward_bf16w(&net->dense[0], cat, OB, B, x, OB, ACT_RELU); round_rows(x, 256, OB, B);
        layer_forward_bf16w(&net->dense[1], x, OB, B, y, OB, ACT_RELU); round_rows(y, 256, OB, B);
        layer_forward_bf16w(&net->dense[2], y, OB, B, x, OB, ACT_RELU); round_rows(x, 256, OB, B);
        layer_forward_bf16w(&net->dense[3], x, OB, B, y, OB, ACT_RELU); round_rows(y, 256, OB, B);
        layer_forward_bf16w(&net->dense[4], y, OB, B, x, OB, ACT_RELU); round_rows(x, 256, OB, B);
        memcpy(y, x, sizeof(float) * 256 * OB);
        layer_forward_bf16w(&net->dense[5], cat, OB, B, x, OB, ACT_RELU); round_rows(x, 256, OB, B);
        layer_forward_bf16w(&net->dense[6], x, OB, B, q, OB, ACT_RELU); round_rows(q, 256, OB, B);
        layer_forward_bf16w(&net->dense[7], q, OB, B, x, OB, ACT_RELU);          /* h8 stays f32 for the alpha head */
        layer_forward_blocked(&net->alpha, x, OB, B, sg, OB, ACT_RELU);
        round_rows(x, 256, OB, B);
        layer_forward_bf16w(&net->bottleneck, x, OB, B, q, OB, ACT_NONE); round_rows(q, 256, OB, B);
        encode_cols(dirs + 3 * c0, dirs + 3 * c0 + 1, dirs + 3 * c0 + 2, 3, B, 4, q + 256 * OB, OB);
        round_rows(q + 256 * OB, 27, OB, B);
        layer_forward_bf16w(&net->viewdirs, q, OB, B, x, OB, ACT_RELU);
        layer_forward_blocked(&net->rgb, x, OB, B, c3, OB, ACT_SIGMOID);          /* f32 head on f32 activations */
        for (size_t c = 0; c < B; ++c) {
            rgb[3 * (c0 + c)] = c3[c]; rgb[3 * (c0 + c) + 1] = c3[OB + c]; rgb[3 * (c0 + c) + 2] = c3[2 * OB + c];
            sigma[c0 + c] = sg[c];
        }
    }
    free(buf);
}

/* ------------------------------------------------------------------------------------------
 * Vec3 helpers (src/vec3.rs:15-34)
 * ---------------------------------------------------------------------------------------- */
static inline void v_cross(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

void oracle_normalize(const float v[3], float o[3]) {
    const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    o[0] = v[0] / len; o[1] = v[1] / len; o[2] = v[2] / len;
}

/* camera_from_samples (src/lib.rs:614-645) */
void oracle_camera_from_values(float near_, float far_, const float origin[3], const float forward[3],
                               const float up[3], const float hwf[3], int width, int height, oracle_camera *c) {
    c->nx = width; c->ny = height;
    c->near_ = near_; c->far_ = far_;
    memcpy(c->pos, origin, sizeof c->pos);
    oracle_normalize(forward, c->dir);
    oracle_normalize(up, c->up);
    const float hw = hwf[1], hh = hwf[0], focal = hwf[2];
    c->alpha_width = atanf((0.5f * hw) / focal);
    c->alpha_height = atanf((0.5f * hh) / focal);
}

/* Camera::get_ray_dir (src/lib.rs:213-231); half = 0.5 in the reference. */
static void ray_dir_impl(const oracle_camera *c, int i, int j, float half, float out[3]) {
    float f[3], r[3], u[3], t[3];
    oracle_normalize(c->dir, f);
    v_cross(f, c->up, t); oracle_normalize(t, r);
    v_cross(r, f, t); oracle_normalize(t, u);
    const float x = (((float)j + half) / (float)c->nx) * 2.0f - 1.0f;
    const float y = 1.0f - (((float)i + half) / (float)c->ny) * 2.0f;
    const float sx = tanf(c->alpha_width), sy = tanf(c->alpha_height);
    const float xs = x * sx, ys = y * sy;
    for (int a = 0; a < 3; ++a) out[a] = (r[a] * xs + u[a] * ys) + f[a];
}

void oracle_get_ray_dir(const oracle_camera *c, int i, int j, float out[3]) { ray_dir_impl(c, i, j, 0.5f, out); }
void oracle_get_ray_dir_nohalf(const oracle_camera *c, int i, int j, float out[3]) { ray_dir_impl(c, i, j, 0.0f, out); }

/* ------------------------------------------------------------------------------------------
 * Sampling / integration
 * ---------------------------------------------------------------------------------------- */
/* stratified_samples (src/lib.rs:233-248) with jitter k from stream 0 */
void oracle_stratified_samples(uint64_t seed, uint32_t pixel_index, float near_, float far_, int count, float *t) {
    if (count == 0) return;
    const float interval = (far_ - near_) / (float)count;
    for (int i = 0; i < count; ++i) {
        const float lower = near_ + (float)i * interval;
        const float upper = lower + interval;
        const float jitter = oracle_uniform(seed, pixel_index, 0u, (uint32_t)i);
        t[i] = lower + (upper - lower) * jitter;
    }
}

/* compute_weights (src/lib.rs:250-283) */
void oracle_compute_weights(const float *sigmas, const float *t, int n, float far_, float *w) {
    float transmittance = 1.0f;
    for (int i = 0; i < n; ++i) {
        float delta = (i + 1 < n) ? t[i + 1] - t[i] : far_ - t[i];
        if (delta < 0.0f) delta = 0.0f;
        const float alpha = 1.0f - expf(-sigmas[i] * delta);
        w[i] = transmittance * alpha;
        transmittance *= 1.0f - alpha;
        if (transmittance < 1e-4f) {
            for (int k = i + 1; k < n; ++k) w[k] = 0.0f;
            break;
        }
    }
}

/* sample_importance (src/lib.rs:289-351) with explicit uniforms u[count]. Returns #samples written. */
int oracle_sample_importance_u(const float *u, const float *samples, const float *weights, int n, int count,
                               float *out, float *cdf_out) {
    if (count == 0 || n < 3) return 0;            /* :295-297 */
    const int m = n - 2;                          /* pdf_weights = weights[1..n-1] */
    float *bins = (float *)malloc(sizeof(float) * (size_t)(n - 1));
    float *adj = (float *)malloc(sizeof(float) * (size_t)m);
    float *cdf = (float *)malloc(sizeof(float) * (size_t)(m + 1));
    for (int i = 0; i + 1 < n; ++i) bins[i] = 0.5f * (samples[i] + samples[i + 1]); /* midpoints :285-287 */
    float sum = 0.0f;
    for (int i = 0; i < m; ++i) {
        const float w = weights[i + 1];
        adj[i] = (w > 0.0f ? w : 0.0f) + 1e-5f; /* w.max(0.0) + 1e-5 */
        sum += adj[i];                            /* iter().sum(): sequential fold */
    }
    if (sum <= 0.0f) { free(bins); free(adj); free(cdf); return 0; }
    for (int i = 0; i < m; ++i) adj[i] /= sum;
    cdf[0] = 0.0f;
    float cumulative = 0.0f;
    for (int i = 0; i < m; ++i) { cumulative += adj[i]; cdf[i + 1] = cumulative; }
    cdf[m] = 1.0f;                                /* :326-328 */
    if (cdf_out) memcpy(cdf_out, cdf, sizeof(float) * (size_t)(m + 1));
    for (int s = 0; s < count; ++s) {
        const float uu = u[s];
        int idx = m - 1;
        for (int j = 0; j < m; ++j)
            if (uu >= cdf[j] && uu < cdf[j + 1]) { idx = j; break; }
        const float cdf_lower = cdf[idx], cdf_upper = cdf[idx + 1];
        float denom = cdf_upper - cdf_lower;
        if (!(denom > 1e-6f)) denom = 1e-6f;      /* .max(1e-6) */
        const float bin_lower = bins[idx], bin_upper = bins[idx + 1];
        const float tt = (uu - cdf_lower) / denom;
        out[s] = bin_lower + (bin_upper - bin_lower) * tt;
    }
    free(bins); free(adj); free(cdf);
    return count;
}

int oracle_sample_importance(uint64_t seed, uint32_t pixel_index, const float *samples, const float *weights, int n,
                             int count, float *out) {
    if (count <= 0) return 0;
    float *u = (float *)malloc(sizeof(float) * (size_t)count);
    for (int s = 0; s < count; ++s) u[s] = oracle_uniform(seed, pixel_index, 1u, (uint32_t)s);
    const int r = oracle_sample_importance_u(u, samples, weights, n, count, out, NULL);
    free(u);
    return r;
}

/* merged.sort_by(partial_cmp) (src/lib.rs:419): stable ascending; only the values are observable. */
static void merge_sort(float *v, float *tmp, int n) {
    if (n < 2) return;
    const int h = n / 2;
    merge_sort(v, tmp, h);
    merge_sort(v + h, tmp, n - h);
    int a = 0, b = h, o = 0;
    while (a < h && b < n) tmp[o++] = (v[b] < v[a]) ? v[b++] : v[a++];
    while (a < h) tmp[o++] = v[a++];
    while (b < n) tmp[o++] = v[b++];
    memcpy(v, tmp, sizeof(float) * (size_t)n);
}

void oracle_sort_ascending(float *v, int n) {
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    merge_sort(v, tmp, n);
    free(tmp);
}

/* integrate_ray (src/lib.rs:176-195) */
void oracle_integrate_ray(const float *colors, const float *sigmas, const float *t, int n, float far_, float out[3]) {
    if (n == 0) { out[0] = out[1] = out[2] = 0.0f; return; }
    float *w = (float *)malloc(sizeof(float) * (size_t)n);
    oracle_compute_weights(sigmas, t, n, far_, w);
    float r = 0.0f, g = 0.0f, b = 0.0f, acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        r += colors[3 * i] * w[i]; g += colors[3 * i + 1] * w[i]; b += colors[3 * i + 2] * w[i];
        acc += w[i];
    }
    const float bg = 1.0f * (1.0f - acc); /* Vec3(1,1,1) * (1 - acc) */
    out[0] = r + bg; out[1] = g + bg; out[2] = b + bg;
    free(w);
}

/* ------------------------------------------------------------------------------------------
 * render_block / render_image (src/lib.rs:353-565).  Extensions (crop, coarse_only, ssaa) are
 * documented in nerf_oracle.h; with them off this is the reference's 8x8-block renderer.
 * Ray grid = (ny*s) x (nx*s); RNG pixel_index = I*RX + J in that grid.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const oracle_net *coarse, *fine;
    oracle_camera rcam; /* camera on the ray grid */
    int nc, nf, coarse_only, naive;
    uint64_t seed;
} rctx;

static void render_rays(const rctx *R, const int *ri, const int *rj, int nr, float *out_rgb, oracle_ray_dump *dump) {
    const int nc = R->nc, nf = R->nf, nm = nc + nf;
    const float near_ = R->rcam.near_, far_ = R->rcam.far_;
    const float *o = R->rcam.pos;
    float *dirs = (float *)malloc(sizeof(float) * 3 * (size_t)nr);
    float *tc = (float *)malloc(sizeof(float) * (size_t)nr * nc);
    float *pts = (float *)malloc(sizeof(float) * 3 * (size_t)nr * nm);
    float *vd = (float *)malloc(sizeof(float) * 3 * (size_t)nr * nm);
    float *sg = (float *)malloc(sizeof(float) * (size_t)nr * nm);
    float *col = (float *)malloc(sizeof(float) * 3 * (size_t)nr * nm);
    float *tm = (float *)malloc(sizeof(float) * (size_t)nr * nm);
    int *cnt = (int *)malloc(sizeof(int) * (size_t)nr);
    float *w = (float *)malloc(sizeof(float) * (size_t)nm);
    for (int r = 0; r < nr; ++r) { /* :367-386 */
        float d[3];
        oracle_get_ray_dir(&R->rcam, ri[r], rj[r], d);
        oracle_normalize(d, dirs + 3 * r);
        const uint32_t pix = (uint32_t)(ri[r] * R->rcam.nx + rj[r]);
        oracle_stratified_samples(R->seed, pix, near_, far_, nc, tc + (size_t)r * nc);
    }
    const size_t NC = (size_t)nr * nc;
    for (int r = 0; r < nr; ++r) /* :392-402 */
        for (int k = 0; k < nc; ++k) {
            const size_t c = (size_t)r * nc + k;
            const float ti = tc[c];
            for (int a = 0; a < 3; ++a) { pts[a * NC + c] = o[a] + dirs[3 * r + a] * ti; vd[3 * c + a] = dirs[3 * r + a]; }
        }
    oracle_forward_batch(R->coarse, pts, vd, NC, col, sg, R->naive); /* :404 */
    if (dump) {
        memcpy(dump->dir_hat, dirs, sizeof(float) * 3);
        memcpy(dump->t_coarse, tc, sizeof(float) * nc);
        memcpy(dump->sigma_coarse, sg, sizeof(float) * nc);
    }
    if (R->coarse_only) {
        for (int r = 0; r < nr; ++r)
            oracle_integrate_ray(col + 3 * (size_t)r * nc, sg + (size_t)r * nc, tc + (size_t)r * nc, nc, far_, out_rgb + 3 * r);
        goto done;
    }
    {
        size_t total = 0;
        for (int r = 0; r < nr; ++r) { /* :409-421 */
            const float *ts = tc + (size_t)r * nc;
            oracle_compute_weights(sg + (size_t)r * nc, ts, nc, far_, w);
            float *m = tm + total;
            memcpy(m, ts, sizeof(float) * nc);
            const uint32_t pix = (uint32_t)(ri[r] * R->rcam.nx + rj[r]);
            int extra;
            if (dump) {
                for (int s = 0; s < nf; ++s) dump->u_fine[s] = oracle_uniform(R->seed, pix, 1u, (uint32_t)s);
                extra = oracle_sample_importance_u(dump->u_fine, ts, w, nc, nf, m + nc, dump->cdf);
                memcpy(dump->w_coarse, w, sizeof(float) * nc);
                memcpy(dump->t_new, m + nc, sizeof(float) * (size_t)extra);
                dump->n_new = extra;
            } else {
                extra = oracle_sample_importance(R->seed, pix, ts, w, nc, nf, m + nc);
            }
            cnt[r] = nc + extra;
            oracle_sort_ascending(m, cnt[r]);
            total += (size_t)cnt[r];
        }
        size_t c = 0;
        for (int r = 0; r < nr; ++r) /* :432-443 */
            for (int k = 0; k < cnt[r]; ++k, ++c) {
                const float ti = tm[c];
                for (int a = 0; a < 3; ++a) { pts[a * total + c] = o[a] + dirs[3 * r + a] * ti; vd[3 * c + a] = dirs[3 * r + a]; }
            }
        oracle_forward_batch(R->fine, pts, vd, total, col, sg, R->naive); /* :445 */
        c = 0;
        for (int r = 0; r < nr; ++r) { /* :447-459 */
            oracle_integrate_ray(col + 3 * c, sg + c, tm + c, cnt[r], far_, out_rgb + 3 * r);
            c += (size_t)cnt[r];
        }
        if (dump) {
            memcpy(dump->t_merged, tm, sizeof(float) * (size_t)cnt[0]);
            memcpy(dump->sigma_fine, sg, sizeof(float) * (size_t)cnt[0]);
            memcpy(dump->rgb_fine, col, sizeof(float) * 3 * (size_t)cnt[0]);
            oracle_compute_weights(sg, tm, cnt[0], far_, dump->w_fine);
        }
    }
done:
    if (dump) memcpy(dump->rgb, out_rgb, sizeof(float) * 3);
    free(dirs); free(tc); free(pts); free(vd); free(sg); free(col); free(tm); free(cnt); free(w);
}

static int make_rctx(rctx *R, const oracle_net *coarse, const oracle_net *fine, const oracle_camera *cam,
                     const oracle_opts *opts, int *s_out) {
    const int s = opts->ssaa > 1 ? opts->ssaa : 1;
    if (opts->n_coarse <= 0) return -1; /* assert!(coarse_samples_per_ray > 0) :483-486 */
    R->coarse = coarse; R->fine = fine;
    R->rcam = *cam;
    R->rcam.nx = cam->nx * s; R->rcam.ny = cam->ny * s;
    R->nc = opts->n_coarse; R->nf = opts->coarse_only ? 0 : opts->n_fine;
    R->coarse_only = opts->coarse_only; R->naive = opts->naive_order; R->seed = opts->seed;
    *s_out = s;
    return 0;
}

int oracle_render_image(const oracle_net *coarse, const oracle_net *fine, const oracle_camera *cam,
                        const oracle_opts *opts, float *rgb_out) {
    rctx R;
    int s;
    if (make_rctx(&R, coarse, fine, cam, opts, &s)) return -1;
    int x0 = 0, y0 = 0, cw = cam->nx, ch = cam->ny;
    if (opts->crop_w > 0 && opts->crop_h > 0) { x0 = opts->crop_x0; y0 = opts->crop_y0; cw = opts->crop_w; ch = opts->crop_h; }
    if (x0 < 0 || y0 < 0 || x0 + cw > cam->nx || y0 + ch > cam->ny) return -2;
    const int RW = cw * s, RH = ch * s, RX0 = x0 * s, RY0 = y0 * s;
    const int bs = 8; /* block_size :491 (partial edge blocks allowed here; the reference asserts %8==0) */
    const int nbx = (RW + bs - 1) / bs, nby = (RH + bs - 1) / bs;
    float *rays = (s > 1) ? (float *)malloc(sizeof(float) * 3 * (size_t)RW * RH) : rgb_out;
#ifdef _OPENMP
    if (opts->n_threads > 0) omp_set_num_threads(opts->n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int blk = 0; blk < nbx * nby; ++blk) { /* block_coords row-major :503-510, par_iter :533-550 */
        const int by = (blk / nbx) * bs, bx = (blk % nbx) * bs;
        int ri[64], rj[64], nr = 0;
        float out[64 * 3];
        for (int i = by; i < by + bs && i < RH; ++i)
            for (int j = bx; j < bx + bs && j < RW; ++j) { ri[nr] = RY0 + i; rj[nr] = RX0 + j; ++nr; }
        render_rays(&R, ri, rj, nr, out, NULL);
        for (int r = 0; r < nr; ++r) /* scatter :552-557 */
            memcpy(rays + 3 * ((size_t)(ri[r] - RY0) * RW + (rj[r] - RX0)), out + 3 * r, sizeof(float) * 3);
    }
    if (s > 1) {
        const float inv = 1.0f / (float)(s * s);
        for (int i = 0; i < ch; ++i)
            for (int j = 0; j < cw; ++j)
                for (int a = 0; a < 3; ++a) {
                    float acc = 0.0f;
                    for (int di = 0; di < s; ++di)
                        for (int dj = 0; dj < s; ++dj) acc += rays[3 * ((size_t)(i * s + di) * RW + (j * s + dj)) + a];
                    rgb_out[3 * ((size_t)i * cw + j) + a] = acc * inv;
                }
        free(rays);
    }
    return 0;
}

int oracle_render_ray_debug(const oracle_net *coarse, const oracle_net *fine, const oracle_camera *cam,
                            const oracle_opts *opts, int i, int j, oracle_ray_dump *dump) {
    rctx R;
    int s;
    if (make_rctx(&R, coarse, fine, cam, opts, &s)) return -1;
    float out[3];
    render_rays(&R, &i, &j, 1, out, dump);
    return 0;
}

/* save_ppm (src/lib.rs:567-580) */
void oracle_quantize_rgb8(const float *rgb, size_t n_pixels, uint8_t *out) {
    for (size_t i = 0; i < 3 * n_pixels; ++i) {
        float v = rgb[i];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); /* clamp(0,1); NaN stays NaN -> `as u8` saturates to 0 */
        const float q = v * 255.0f + 0.5f;
        out[i] = (q != q) ? 0 : (uint8_t)q;
    }
}

int oracle_save_ppm(const char *path, int width, int height, const float *rgb) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    uint8_t *buf = (uint8_t *)malloc((size_t)width * height * 3);
    oracle_quantize_rgb8(rgb, (size_t)width * height, buf);
    fwrite(buf, 1, (size_t)width * height * 3, f);
    free(buf);
    fclose(f);
    return 0;
}
