"""The multi-rank path of bench.py end to end on ONE GPU: two ranks share the device, the rank
payloads are gathered through host memory (gloo), and rank 0 checks the gathered frame bit for
bit against a single-rank render of the same samples (bench.py --check)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("launcher,exchange", [("torchrun", "gather"), ("self", "allgather")])
def test_two_rank_bench_step_gathers_the_single_rank_frame(launcher, exchange):
    """launcher "self": `python bench.py --gpus 2` with no launcher around it -- bench.py starts
    torch.distributed.run itself, as a child, and passes its output and exit code on."""
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "2", "--backend", "gloo",
            "--exchange", exchange, "--check", "--no-cpu-baseline"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_ADDR"] = "127.0.0.1"
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["gathered_frame_equals_single_rank_render"] is True
    assert out["config"]["samples_per_step"] == 1920 * 1080 * 2 * 2
    # both exchanges are timed: the one asked for in the step, the other beside it
    other = "allgather" if exchange == "gather" else "gather"
    assert out["config"]["exchange"] == exchange and f"resolve_gather_copy_ms_with_{other}" in out["config"]


def test_bench_refuses_a_world_size_that_is_not_its_gpu_count():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stdout + res.stderr)


RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
rt = load_package()
from rtiow_amd import scenes, tiling
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w, h, spp = 320, 180, 2
scene = rt.Scene()
scenes.cover_scene(scene, w, h)
ctx = rt.Context(0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
ctx.upload_scene(scene)
n4 = tiling.payload_float4s(w, h, 1)
payload = torch.zeros(n4, 4, dtype=torch.float32, device=dev)
gathered = torch.zeros(n4, 4, dtype=torch.float32, device=dev)
frame = torch.zeros(h, w, 4, dtype=torch.float32, device=dev)
frame8 = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
ctx.set_target(w, h, 0, 1, external_ptr=payload.data_ptr())
ctx.render(0, spp, 50)
dist.all_gather_into_tensor(gathered, payload)          # RCCL, on the render stream
ctx.resolve_gathered(gathered.data_ptr(), 1, frame.data_ptr())
ctx.resolve_gathered_rgba8(gathered.data_ptr(), 1, spp, frame8.data_ptr())
torch.cuda.synchronize(dev)
want, want8 = ctx.read_accum(), ctx.read_rgba8(spp)
assert np.array_equal(frame.cpu().numpy().view(np.uint32), want.view(np.uint32))
assert np.array_equal(frame8.cpu().numpy(), want8)
assert want[..., :3].max() > 0
dist.destroy_process_group()
print("rccl-one-rank-ok")
"""


def test_rccl_all_gather_feeds_the_device_resolve():
    """The collective leg of the bench step with the real backend (RCCL), as far as one GPU allows:
    a one-rank process group, all_gather_into_tensor on the render stream, both device resolves."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    res = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "rccl-one-rank-ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
