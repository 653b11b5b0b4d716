"""Multi-rank image partition: two processes over gloo exchange their tile payloads with one
all-gather and rebuild the image (the N > 1 data path of bench.py, without the kernels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("shape", [(64, 40), (61, 35), (1920, 1080)])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_is_a_bijection(tiling, shape, world):
    w, h = shape
    rng = np.random.RandomState(0)
    img = rng.rand(h, w, 4).astype(np.float32)
    payloads = np.stack([tiling.payload_from_image(img, r, world) for r in range(world)])
    assert payloads.shape[1] == tiling.payload_float4s(w, h, world)
    back = tiling.image_from_payloads(payloads, w, h, world)
    assert np.array_equal(back, img)
    # ownership is balanced to within one tile column per tile row
    _, tiles_y, tprl = tiling.geometry(w, h, world)
    counts = [(np.abs(p).sum(axis=1) > 0).sum() for p in payloads]
    assert max(counts) - min(counts) <= tiles_y * 64 * 1 + 64 * tiles_y


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, HERE)
    from conftest import load_package
    load_package()
    from rtiow_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img = np.random.RandomState(42).rand(h, w, 4).astype(np.float32)      # the image every rank would converge to
    mine = torch.from_numpy(tiling.payload_from_image(img, rank, world))  # what this rank rendered
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    full = tiling.image_from_payloads(torch.stack(gathered).numpy(), w, h, world)
    ok = bool(np.array_equal(full, img))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_all_gather_rebuilds_the_image_on_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29517 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, 100, 60, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
