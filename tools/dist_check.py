#!/usr/bin/env python3
"""Rehearsal of the N > 1 render path on ONE GPU: launch with
    NERF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29544 tools/dist_check.py
Every rank renders its row band on the same device, the bands are gathered (gloo, host copies), and every rank checks the
assembled frame bit for bit against its own single-process render of the full frame (the per-pixel counter RNG makes a band
identical to the same rows of a full render).  On an 8-GPU node the same code runs with nccl (RCCL) and one GPU per rank."""
import os, sys
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_rs_amd as N

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("NERF_BENCH_BACKEND", "nccl")
dev_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
size = int(os.environ.get("NERF_CHECK_SIZE", "200"))
with N.Renderer(dev_index) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), size, size, 64)
    ok = True
    # plain renders: contiguous bands; skip_dead / certify_zero: rows dealt out round-robin (distributed.partition_for), packed bands + one index_copy
    for dtype, mode in (("f32", {}), ("bf16x3", {}), ("f32", {"skip_dead": True}), ("f32", {"certify_zero": True}), ("f16x2", {"certify_zero": True})):
        for kw in ({}, {"crop": (10, 7, size - 33, size - 21)}):       # ragged bands: (size - 21) rows over `world` ranks
            full = N.render_image(r.coarse, r.fine, cam, 128, seed=4, dtype=dtype, **kw)   # the PLAIN frame: the skipping modes must reproduce it too
            got = N.render_image_distributed(r.coarse, r.fine, cam, 128, seed=4, dtype=dtype, **mode, **kw)
            same = bool((got == full).all()) and got.shape == full.shape
            ok = ok and same
            print(f"rank {rank}/{world} {dtype} {mode or 'plain'} {kw or 'full'}: {'identical' if same else 'MISMATCH'}", flush=True)
flag = torch.tensor([1 if ok else 0])
if backend == "nccl":
    flag = flag.cuda()
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
