"""World-size-2 gloo run (CPU) of the data-parallel plumbing bench.py uses on N GPUs: shard math,
weight-arena broadcast, max-over-ranks timing, output gather. The compute itself needs a GPU and is
covered by the -m gpu tests; here each rank's "output" is a deterministic function of its shard so the
gathered result can be checked against the single-process answer."""
import os
import socket
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # spawned workers import this module without conftest.py

load_package()

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visioncpp_amd import dist as vdist
from visioncpp_amd import synth


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 64, 255, 1000):
        for world in (1, 2, 3, 4, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # "weight arena": rank 0 packs the real synthetic checkpoint bytes, the others receive them
        sd = synth.state_dict(synth.TINY, seed=3)
        blob = np.concatenate([v.astype(np.float16).view(np.uint8).ravel() for v in sd.values()])
        arena = torch.from_numpy(blob.copy()) if rank == 0 else torch.zeros(blob.size, dtype=torch.uint8)
        vdist.broadcast_bytes(arena, src=0)
        assert np.array_equal(arena.numpy(), blob)
        # shard 5 images over 2 ranks; fake per-image result = per-image checksum plane
        imgs = synth.images(5, 28, 28, seed=1)
        b, e = vdist.shard_range(len(imgs), rank, world)
        local = torch.from_numpy(imgs[b:e].astype(np.float32).mean(axis=-1))
        got = vdist.gather_outputs(local, dst=0)
        slow = vdist.max_over_ranks(1.0 + rank, "cpu")
        assert slow == float(world)
        if rank == 0:
            np.save(os.path.join(out_dir, "gathered.npy"), got.numpy())
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    want = synth.images(5, 28, 28, seed=1).astype(np.float32).mean(axis=-1)
    np.testing.assert_array_equal(got, want)


def test_bench_self_launch_starts_n_ranks_and_returns_child_code():
    """`python bench.py --gpus 2` without RANK in the environment must start two fresh rank processes through
    torch.distributed.run as a child (never an exec) and hand back its exit code. On this CPU box every rank refuses
    ("needs an MI355X", no CPU fallback), which is exactly what shows both ranks ran the real entry."""
    import subprocess

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box this test stays a CPU test
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-backend", "gloo",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]
    assert "launch with torch.distributed.run" not in r.stderr


@pytest.mark.gpu
def test_bench_self_launch_gloo_two_ranks_one_gpu():
    """The N > 1 launch path end to end on the one GPU of the box: two ranks over gloo, both on device 0, weight arena
    broadcast from rank 0, one JSON line with n_gpus 2 from rank 0."""
    import json
    import subprocess

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
                        "--device", "0", "--min-seconds", "0", "--no-pipeline", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and res["value"] > 0
