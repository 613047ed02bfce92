"""`python bench.py --gpus N` must work without an outer torch.distributed.run: bench.launch_ranks starts the N ranks,
relays rank 0's single JSON line and fails if any rank fails.  Exercised here on CPU (gloo, world size 2 and 3) with a
stand-in rank script; the GPU suite runs the real `bench.py --gpus 2` (tests/test_gpu_parity.py)."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

CHILD = os.path.join(ROOT, "tests", "helpers", "launch_child.py")
DRIVER = ("import sys; sys.path.insert(0, {root!r}); import bench; "
          "bench.launch_ranks({n}, {argv!r}, script={child!r})")


def _run(n, argv, timeout=120):
    code = DRIVER.format(root=ROOT, n=n, argv=argv, child=CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("world", [2, 3, 8])          # 8 = the driver's largest scaling run
def test_launch_ranks_relays_rank0_json(world):
    p = _run(world, ["--steps", "1"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # ONE JSON line, nothing from the other ranks
    d = json.loads(lines[0])
    assert d["ranks"] == world and d["backend"] == "gloo" and d["sum"] == world * (world + 1) / 2
    assert d["local_rank"] == 0 and d["argv"] == ["--steps", "1"]
    assert "noise from a non-zero rank" in p.stderr


def test_launch_ranks_fails_when_a_rank_fails():
    t0 = time.time()
    p = _run(2, ["--fail-rank", "1"])
    assert p.returncode != 0 and "rank 1 exited with code 7" in p.stderr
    assert time.time() - t0 < 60                           # rank 0 (stuck in the rendezvous) was stopped, not waited for


def test_bench_parent_does_not_need_torch_or_a_gpu_before_forking():
    """The parent must start its ranks before anything initialises HIP: bench.py imports torch only inside the rank code."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "import torch" not in head
    main_src = src[src.index("def main():"):]
    assert main_src.index("launch_ranks(args.gpus") < main_src.index("import torch")
