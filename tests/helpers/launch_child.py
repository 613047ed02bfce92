"""Stand-in rank for tests/test_bench_launch.py: what bench.py's rank code does around the GPU work -- read RANK /
WORLD_SIZE / MASTER_* from the environment, join a gloo group, run one collective, rank 0 prints ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)  # before the rendezvous: the other ranks are left waiting, the launcher must stop them
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"ranks": dist.get_world_size(), "backend": dist.get_backend(), "sum": float(t.item()),
                      "local_rank": int(os.environ["LOCAL_RANK"]), "argv": sys.argv[1:]}), flush=True)
else:
    print("noise from a non-zero rank must not reach the parent's stdout", flush=True)
dist.destroy_process_group()
