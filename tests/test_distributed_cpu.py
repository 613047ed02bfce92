"""N > 1 path on CPU: world_size-2 gloo run of render_image_distributed (band split + ONE all-gather).  The band
renderer is the oracle here (allowed in tests); on the GPU box the same function calls the HIP renderer."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SCENE


def test_band_of_rank_partitions_rows(native):
    for n_rows in (1, 7, 8, 100, 800, 801):
        for world in (1, 2, 3, 4, 8):
            bands = [native.band_of_rank(n_rows, r, world) for r in range(world)]
            assert bands[0][0] == 0 and sum(b[1] for b in bands) == n_rows
            for (y0, n0), (y1, _) in zip(bands, bands[1:]):
                assert y1 == y0 + n0
            assert max(b[1] for b in bands) - min(b[1] for b in bands) <= 1
    assert native.band_of_rank(800, 3, 8) == (300, 100)  # rows [100 g, 100 g + 100) on GPU g


def _worker(rank, world, port, crop, out_dir, stripe):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import nerf_rs_amd
    import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = O.load_samples(os.path.join(SCENE, "tf_reference_samples.json"))
        co, fi = O.Net(os.path.join(SCENE, "coarse")), O.Net(os.path.join(SCENE, "fine"))
        ocam = O.camera_from_samples(S, 800, 800)
        cam = nerf_rs_amd.camera_from_samples(S, 800, 800)

        def band_renderer(c):
            return O.render_image(co, fi, ocam, O.make_opts(64, 128, crop=c, seed=0, threads=2))

        marks = []
        frame = nerf_rs_amd.render_image_distributed(None, None, cam, 128, seed=0, crop=crop, band_renderer=band_renderer, timings=marks,
                                                     stripe_rows=stripe)
        np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame)
        (ms_render, ms_gather), = [m.ms() for m in marks]  # attribution marks: render vs gather (bench.py's N > 1 line)
        assert ms_render > 0 and ms_gather > 0
    finally:
        dist.destroy_process_group()


def test_partition_rule_and_row_layout(native):
    """Cost follows the scene (skip_dead / skip_empty / certify_zero) => rows round-robin; uniform cost => contiguous bands; the torch
    path's row layout is the C ABI's (nerf_render_opts.band_*)."""
    from nerf_rs_amd.distributed import band_row_indices
    assert native.partition_for() == 0 and native.partition_for(skip_dead=True) == native.partition_for(certify_zero=True) == native.partition_for(skip_empty=True) == 1
    for h, n, stripe in ((7, 2, 1), (7, 3, 2), (800, 8, 1), (801, 8, 0), (5, 8, 1)):
        for r in range(n):
            assert np.array_equal(band_row_indices(h, r, n, stripe), native.band_row_indices(h, r, n, stripe))
            assert len(band_row_indices(h, r, n, stripe)) == native.band_rows(h, r, n, stripe)
    assert band_row_indices(800, 3, 8, 1)[:3].tolist() == [3, 11, 19]


@pytest.mark.parametrize("world,stripe", [(2, 0), (3, 0), (2, 1), (3, 2), (3, 1)])
def test_gloo_band_gather_reassembles_the_frame(native, tmp_path, world, stripe):
    import torch.multiprocessing as mp
    g = np.load(os.path.join(GOLDEN, "crop_c3_800_64_128.npz"))
    x0, y0, w, h = (int(v) for v in g["crop"])
    crop = (x0 + 8, y0 + 16, 16, 7)  # 7 rows over 2 (4+3) or 3 (3+2+2) ranks: ragged bands; striped: rows 0,2,4,6 / 1,3,5 etc., packed, then put in place
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, crop, str(tmp_path), stripe), nprocs=world, join=True)
    expect = g["image"][16:23, 8:24]
    for r in range(world):
        frame = np.load(tmp_path / f"frame{r}.npy")
        assert frame.shape == (7, 16, 3) and np.array_equal(frame, expect)
