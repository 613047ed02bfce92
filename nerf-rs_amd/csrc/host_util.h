// host_util.h -- host-side pieces around the hot path: tensor-directory loader, weight packer, camera,
// minimal JSON reader, PPM writer.  Internal to libnerf_mi355x.so.
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/nerf_mi355x.h"

namespace nerfhost {

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
};

// The 12 layers of one network, by the reference's tensor names (src/lib.rs:133-169).
struct HostNet {
    struct L {
        int K = 0, N = 0;
        std::vector<float> w, b;
    };
    L dense[8], bottleneck, viewdirs, rgb, alpha;
};

// load_shapes + load_tensor (src/lib.rs:34-42, 62-74).  Returns NERF_OK or an error code + message.
int read_tensor_dir(const std::string &dir, std::map<std::string, Tensor> &out, std::string &err);
// take_matrix/take_bias by name (src/lib.rs:115-169) + architecture check for the fused kernel.
int assemble_net(std::map<std::string, Tensor> &params, HostNet &net, std::string &err);

// Packed device images (mlp_layout.h).
void pack_network(const HostNet &net, std::vector<float> &wstream, std::vector<float> &small);
// bf16 weight stream of mlp_kernel_bf16.hip (round-to-nearest-even); the small-parameter block is shared with fp32.
void pack_network_bf16(const HostNet &net, std::vector<uint16_t> &wstream);
uint16_t f32_to_bf16_rne(float v);
// the first bf16 design's piece order as f32 values, no padding (mlp_layout.h kPiecesV1 x 512): common source of all bf16 streams
void pack_network_v1order_f32(const HostNet &net, std::vector<float> &v1f);
void bf16_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &wstream);
// f32 by three-way bf16 split (mlp_kernel_bf16x3.hip): w = w1 + w2 + w3, three pieces per v1 piece
void split_bf16x3(float v, uint16_t out[3]);
void x3_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &wstream);
// f32 by two-way f16 split (mlp_kernel_f16x2.hip): w = w1 + w2, two pieces per v1 piece; false if a weight exceeds the f16 range
void split_f16x2(float v, uint16_t out[2]);
bool x2_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &wstream);
bool f16_stream_from_v1order(const std::vector<float> &v1f, std::vector<uint16_t> &wstream); // one f16 per weight, bf16_stream_from_v1order's order

// camera_from_samples (src/lib.rs:614-645)
void camera_from_values(float near_, float far_, const float origin[3], const float forward[3], const float up[3],
                        const float hwf[3], int width, int height, nerf_camera *out);
int camera_from_json(const std::string &path, int width, int height, nerf_camera *out, std::string &err);
// Orthonormal basis + slopes of Camera::get_ray_dir (src/lib.rs:216-218, 225-226)
void camera_basis(const nerf_camera &cam, float r[3], float u[3], float f[3], float *sx, float *sy);

void quantize_rgb8(const float *rgb, size_t n_pixels, uint8_t *out);

// nerf_host_api.cpp: error message of context-free calls (nerf_last_error(NULL)), per thread; the packed-blob reader
int fail_noctx(int code, const std::string &msg);
const char *last_error_noctx();
int read_blob_file(const std::string &path, std::vector<float> &wstream, std::vector<float> &small, std::string &err);
int save_ppm(const std::string &path, int width, int height, const float *rgb, std::string &err);

// certify_zero's audit policy (nerf_api.cpp render_device; exposed host-only as nerf_debug_certify_policy so that it is tested without a
// GPU).  Given what the audit of one network found in one frame, decide whether the frame stands and, if not, the widened margin:
//   a violation (an audited certificate with a positive exact density)              -> max(4 m, 4 err)
//   least headroom below m / 2 (an audited sample closer to a positive density)     -> max(2 m, 4 err)
//   largest |bf16 - exact| on an audited certificate above m / 2                    -> max(1.25 m, 3 err)
// err = max(largest error, m - least headroom).  Returns 0 (stands), 1 / 2 / 3 (which rule widened; *new_margin is set).
int certify_policy(float margin, uint64_t audited, uint64_t violations, float headroom, float max_error, float *new_margin);

} // namespace nerfhost
