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


def test_two_rank_bench_step_gathers_the_single_rank_frame():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--spp", "2", "--backend", "gloo", "--check", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["gathered_frame_equals_single_rank_render"] is True
    assert out["config"]["samples_per_step"] == 1920 * 1080 * 2 * 2
