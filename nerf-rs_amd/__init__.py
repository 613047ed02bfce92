"""nerf-rs_amd: MI355X-native drop-in for the hot path of elisabeth96/nerf-rs.

Python host mirror of the reference's interface for this path (same names, argument meaning and error behaviour),
sitting on top of the C ABI of libnerf_mi355x.so (include/nerf_mi355x.h).  PyTorch is optional plumbing only
(device buffers, streams, torch.distributed over RCCL for the framebuffer gather).
"""
from ._lib import NerfError, build_native, lib_path, load_library  # noqa: F401
from .api import (Camera, Network, RenderOpts, Renderer, Stats, band_row_indices, band_rows, camera_from_pose, camera_from_samples,  # noqa: F401
                  load_network_blob, load_network_from_dir, pack_network_dir, quantize_rgb8, quantize_rgba8, render_image, render_image_multi, save_ppm)
from .distributed import band_of_rank, partition_for, render_image_distributed  # noqa: F401

FLOP_PER_POINT_FULL = 1_186_816
FLOP_PER_POINT_SIGMA = 982_528


def flop_per_ray(n_coarse, n_fine, coarse_only=False):
    """Algorithmic FLOPs per ray: coarse sigma-only + fine full on the merged samples."""
    if coarse_only:
        return n_coarse * FLOP_PER_POINT_FULL
    return n_coarse * FLOP_PER_POINT_SIGMA + (n_coarse + n_fine) * FLOP_PER_POINT_FULL
