"""Multi-GPU rendering: rays shard embarrassingly (every ray is independent -- the reference already exploits this
with rayon over 8x8 blocks, src/lib.rs:533-550).  One process per GPU; rank r renders a contiguous band of rows on
its own device and ONE collective -- an all-gather of the bands over RCCL/xGMI -- assembles the framebuffer on every
rank.  Weights are replicated (4.8 MB).  No other exchange exists on this path.
"""
import numpy as np


def band_of_rank(n_rows, rank, world_size):
    """Contiguous, balanced row bands: cost per ray is uniform (the reference never skips work), so equal row counts
    balance.  Returns (first_row, n_rows_of_rank); the first (n_rows % world) ranks get one extra row."""
    base, rem = divmod(int(n_rows), int(world_size))
    y0 = rank * base + min(rank, rem)
    return y0, base + (1 if rank < rem else 0)


def render_image_distributed(coarse, fine, camera, fine_samples_per_ray=128, *, seed=0, coarse_only=False, crop=None,
                             ssaa=1, dtype="f32", skip_empty=False, skip_dead=False, group=None, band_renderer=None, device=None, return_tensor=False):
    """render_image over all ranks of `group` (torch.distributed; backend nccl == RCCL on ROCm, gloo in CPU tests).

    Every rank returns the full (h, w, 3) frame.  `band_renderer(crop) -> ndarray (rows, w, 3)` overrides the GPU
    renderer (used by the CPU gloo tests, where the band is produced by the oracle).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x0, y0, w, h = crop if crop else (0, 0, camera.nx, camera.ny)
    b0, rows = band_of_rank(h, rank, world)
    max_rows = band_of_rank(h, 0, world)[1]
    band_crop = (x0, y0 + b0, w, rows)
    if band_renderer is not None:
        dev = torch.device("cpu") if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if rows > 0:
            band[:rows] = torch.from_numpy(np.ascontiguousarray(band_renderer(band_crop), dtype=np.float32)).to(dev)
    else:
        from .api import render_image
        dev = torch.device("cuda", coarse.renderer.device) if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if rows > 0:
            stream = torch.cuda.current_stream(dev).cuda_stream
            render_image(coarse, fine, camera, fine_samples_per_ray, seed=seed, coarse_only=coarse_only, crop=band_crop,
                         ssaa=ssaa, dtype=dtype, skip_empty=skip_empty, skip_dead=skip_dead, device_out=band.data_ptr(), stream=stream)
    if band.device.type == "cuda" and dist.get_backend(group) != "nccl":
        # rehearsal only (e.g. gloo with several ranks on one GPU): the collective runs on host copies of the bands
        torch.cuda.current_stream(band.device).synchronize()
        host = band.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        gathered = torch.stack(parts).to(band.device)
    else:
        gathered = torch.empty((world, max_rows, w, 3), dtype=torch.float32, device=band.device)
        dist.all_gather_into_tensor(gathered.view(-1), band.view(-1), group=group) if band.device.type == "cuda" else \
            dist.all_gather(list(gathered.unbind(0)), band, group=group)
    if h % world == 0:
        frame = gathered.view(h, w, 3)
    else:
        frame = torch.cat([gathered[r, :band_of_rank(h, r, world)[1]] for r in range(world)], dim=0)
    return frame if return_tensor else frame.cpu().numpy()
