"""Benchmark of the path-tracing hot path: Msamples/s on the Book-1 cover scene, 1920x1080,
max depth 50, one frame of 64 samples per pixel per GPU per step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; started without
    a launcher, bench.py starts that command itself as a child process and passes its output on)

One step = clear the accumulation buffer, one render call of `spp` samples of every owned pixel,
(N > 1: one gather of the rank payloads to rank 0 over RCCL), the device resolve into the row-major
RGBA8 frame the reference hands to glTexImage2D, and the copy of that frame (8.3 MB) to pinned
host memory on rank 0.  The host does not wait between steps: two frame buffers take turns, the copy of
frame k runs on a stream of its own beside the kernels of step k + 1, and the timed region closes with a
synchronize, when the frames of all K steps are in host memory (a blocking read after every step costs
0.8 ms of idle GPU per step).  The scene lives in device memory before timing starts.

N GPUs shard the 8x8-pixel tiles of the same 1080p frame.  The headline value keeps per-GPU work
fixed ("weak" scaling): each rank renders 64*N samples per pixel of its tiles, so the gathered
frame is a 64*N-spp image.  A second timed loop splits ONE 64-spp frame over the N ranks (strong
scaling: the latency of the frame BASELINE.json's metric names) and is reported beside it.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

W, H, SPP, MAX_DEPTH = 1920, 1080, 64, 50
PEAK_FP32_TFLOPS = 157.3  # MI355X peak FP32 vector rate (MI355X_MICROARCH.md, chip-level parameters)

# Algorithmic flop costs (add/sub/mul = 1, fma = 2, div/rcp/sqrt = 1, compare/select/int = 0),
# following the reference's arithmetic: slab test 3 x (1 rcp + 2 sub + 2 mul); sphere test to the
# discriminant 23; triangle test 51; per accepted closest hit: one root 3 + hit record 17
# (triangle: 40); scatter by kind; camera ray + pixel jitter 34; accumulate 3 per sample.
COST = {"node_tests": 15, "sphere_tests": 23, "tri_tests": 51, "hits": 20, "scatter_lambertian": 35,
        "scatter_metal": 55, "scatter_dielectric": 60, "scatter_pbr": 60, "samples": 37}


def flops_from_counters(c):
    return float(sum(c[k] * v for k, v in COST.items()))


SHADER_CLOCK_HZ = 2.4e9  # MI355X peak engine clock, the one the 157.3 TFLOP/s figure is quoted at
KERNEL_SOURCES = ("rt_kernels.hip", "rt_device.h", "Makefile")  # the Makefile holds the kernels' compiler flags


def kernel_source_hash():
    """SHA-256 over the kernel sources the committed PMC figures were measured on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def profiled(world, spp):
    """The committed PMC figures (profiles/hbm_traffic.json, written by profiles/summarize_traffic.py
    from rocprofv3 passes over this same workload) -- or {} when they do not describe this run: another
    configuration, or kernel sources that changed since the passes were made (the file records the
    hash of the sources it was measured on)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world != 1 or spp != SPP or not os.path.exists(path):
        return {}
    with open(path) as f:
        d = json.load(f)
    try:
        if d.get("kernel_source_sha256") != kernel_source_hash():
            return {}
    except OSError:
        return {}
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU per step")
    ap.add_argument("--check", action="store_true",
                    help="after timing, rank 0 renders the whole frame alone and compares it bit for bit with the gathered one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path on fewer GPUs than ranks (ranks share devices, "
                         "the gather goes through host memory); not a measurement")
    ap.add_argument("--exchange", default="gather", choices=["gather", "allgather"],
                    help="per-frame exchange of the rank payloads: gather to rank 0, which alone resolves the frame "
                         "(default), or all-gather to every rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: start one rank per GPU under torch.distributed.run as a CHILD process
        # (nothing in this process has touched the GPU yet), pass its output on and leave with its code.
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
        raise SystemExit(child.returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    rt = load_package()
    from rtiow_amd import scenes, tiling

    scene = rt.Scene()
    scenes.cover_scene(scene, W, H, wrap_bvh=True)
    ctx = rt.Context(local_rank)
    # One side stream carries everything: torch's fills and collectives are ordered against the
    # render and resolve kernels on it without host synchronisation.
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.upload_scene(scene)

    n4 = tiling.payload_float4s(W, H, world)
    payload = torch.zeros(n4, 4, dtype=torch.float32, device=dev)
    # rank payloads end to end; with the root-only exchange only rank 0 holds them
    gathered = torch.zeros(world * n4, 4, dtype=torch.float32, device=dev) if world > 1 else None
    # The frame leaves the GPU as the application would take it: resolved to RGBA8 on the render stream, copied to
    # pinned host memory on a copy stream of its own.  Two frames in flight (device and host buffers take turns), so
    # the host queues step k + 1 while step k renders and the copy of frame k runs beside the kernels of k + 1; the
    # timed region ends with a synchronize, i.e. with every frame of the K steps in host memory.
    frame8 = [torch.zeros(H, W, 4, dtype=torch.uint8, device=dev) for _ in range(2)] if rank == 0 else None
    host8 = [torch.zeros(H, W, 4, dtype=torch.uint8).pin_memory() for _ in range(2)]   # what the application would upload as a texture
    copy_stream = torch.cuda.Stream(dev)
    copied = [torch.cuda.Event(), torch.cuda.Event()]
    exchange = [args.exchange]
    ctx.set_target(W, H, rank, world, external_ptr=payload.data_ptr())
    kernel_ms = []
    steps_done = [0]

    def step(spp_step, record=False):
        b = steps_done[0] & 1
        steps_done[0] += 1
        payload.zero_()
        ctx.render(0, spp_step, MAX_DEPTH)
        if record:
            kernel_ms.append(ctx.last_render_ms())
        # The one exchange of the frame.  Only rank 0 resolves, so by default the payloads are gathered there
        # (33.2 MB x (N - 1) / N into one GPU); the all-gather moves N times as much.
        if world == 1:
            pass
        elif args.backend == "nccl":
            if exchange[0] == "allgather":
                dist.all_gather_into_tensor(gathered, payload)
            else:
                dist.gather(payload, list(gathered.view(world, n4, 4).unbind(0)) if rank == 0 else None, dst=0)
        else:
            host = torch.empty(gathered.shape, dtype=gathered.dtype)
            if exchange[0] == "allgather":
                dist.all_gather_into_tensor(host, payload.cpu())
            else:
                dist.gather(payload.cpu(), list(host.view(world, n4, 4).unbind(0)) if rank == 0 else None, dst=0)
            if rank == 0 or exchange[0] == "allgather":
                gathered.copy_(host)
        if rank == 0:
            stream.wait_event(copied[b])   # the copy of two steps ago has read frame8[b]
            ctx.resolve_gathered_rgba8((gathered if world > 1 else payload).data_ptr(), world, spp_step, frame8[b].data_ptr())
            resolved = torch.cuda.Event()
            resolved.record(stream)
            copy_stream.wait_event(resolved)
            with torch.cuda.stream(copy_stream):
                host8[b].copy_(frame8[b], non_blocking=True)
                copied[b].record(copy_stream)
        return b

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(spp_step, steps, warmup):
        for _ in range(warmup):
            step(spp_step)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(spp_step)
        fence()
        dt = time.perf_counter() - t0
        return max_over_ranks(dt) if world > 1 else dt

    spp_weak = args.spp * world  # per-GPU work fixed: 1/world of the tiles, world x the samples
    elapsed = timed(spp_weak, args.steps, args.warmup)
    # Strong scaling: ONE args.spp-sample frame split over the ranks (for one rank the same workload).
    elapsed_strong = timed(args.spp, args.steps, 1) if world > 1 else elapsed
    # the other exchange, for comparison (same steps, same work)
    elapsed_other = None
    if world > 1:
        exchange[0] = "allgather" if args.exchange == "gather" else "gather"
        elapsed_other = timed(spp_weak, args.steps, 1)
        exchange[0] = args.exchange

    # Kernel duration by HIP events on the launch stream, in separate untimed steps (reading an
    # event blocks the host, which must not sit inside the timed region).
    for _ in range(3):
        step(spp_weak, True)
    fence()
    avg_kernel_ms = float(np.mean(kernel_ms))

    samples_per_step = W * H * spp_weak  # whole job: every pixel, spp * world samples
    value = samples_per_step * args.steps / elapsed / 1e6
    ms_step = 1e3 * elapsed / args.steps
    ms_strong = 1e3 * elapsed_strong / args.steps

    out = {
        "metric": "Msamples/s (1080p, 50 bounces)",
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"Book-1 cover scene (488 spheres, BVH(world)), {W}x{H}, {args.spp} spp per GPU per step "
                        f"({spp_weak} spp frame), max depth 50, progressive accumulation; every step's "
                        f"frame is resolved to RGBA8 and copied to pinned host memory (copy of frame k beside the "
                        f"kernels of k + 1; all K frames are there when the timed region ends)",
            "samples_per_step": samples_per_step,
            "parallelism": f"tiles8x8 sharded over {world} rank(s)" + ((f", {args.exchange} (RCCL)" if args.backend == "nccl" else f", {args.exchange} (gloo rehearsal through host memory)") if world > 1 else ""),
            "render_kernels_ms": round(avg_kernel_ms, 4),
            "resolve_gather_copy_ms": round(ms_step - avg_kernel_ms, 4),
            "strong_scaling": {
                "what": f"one {args.spp}-spp {W}x{H} frame split over {world} rank(s): render + gather + resolve + copy to host",
                "ms_per_64spp_frame": round(ms_strong * 64.0 / args.spp, 4),
                "msamples_per_s": round(W * H * args.spp / (ms_strong * 1e-3) / 1e6, 2),
            },
        },
    }

    if elapsed_other is not None:
        other = "allgather" if args.exchange == "gather" else "gather"
        out["config"]["exchange"] = args.exchange
        out["config"][f"resolve_gather_copy_ms_with_{other}"] = round(1e3 * elapsed_other / args.steps - avg_kernel_ms, 4)

    if args.check and world > 1:
        fence()
        last = step(spp_weak)
        fence()
        if rank == 0:
            solo = rt.Context(local_rank)
            solo.upload_scene(scene)
            solo.set_target(W, H)
            solo.render(0, spp_weak, MAX_DEPTH)
            solo.sync()
            want = solo.read_accum()
            want8 = solo.read_rgba8(spp_weak)
            solo.close()
            got = tiling.image_from_payloads(gathered.cpu().numpy().reshape(world, n4, 4), W, H, world)
            same = bool(np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)))
            same8 = bool(np.array_equal(host8[last].numpy(), want8))
            out["gathered_frame_equals_single_rank_render"] = same and same8
            if not (same and same8):
                raise SystemExit("gathered frame differs from the single-rank render")
        fence()

    if rank == 0:
        # Algorithmic flops per launch from the kernel's own path statistics (a counting launch
        # over the same pixels and samples, untimed), priced with COST.
        counters = ctx.count(0, min(spp_weak, 8), MAX_DEPTH)
        flops_per_sample = flops_from_counters(counters) / counters["samples"]
        own_samples = W * H * spp_weak / world
        achieved = flops_per_sample * own_samples / (avg_kernel_ms * 1e-3) / 1e12
        prof = profiled(world, args.spp)
        out["roofline"] = {
            "bound": "valu_fp32",
            "achieved": round(achieved, 3),
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
            "traffic": prof.get("hbm_bytes_per_launch"),   # null when the PMC passes predate the kernel sources
            # the launch as a whole: first pass (camera rays, a tile at a time), the persistent loop over its path
            # records (about three quarters of the time), and the ordered per-pixel sums
            "kernel": "rt::primaryPass<false, 9, false> + rt::renderWavefront<true, false, false, 9, false, true, false, false> "
                      "+ rt::accumulateColors (tables in LDS, spheres-only builds)",
            "kernel_ms": round(avg_kernel_ms, 4),
            "flops_per_sample": round(flops_per_sample, 1),
            "rays_per_sample": round(counters["rays"] / counters["samples"], 3),
            "node_tests_per_ray": round(counters["node_tests"] / counters["rays"], 2),
            "sphere_tests_per_ray": round(counters["sphere_tests"] / counters["rays"], 2),
        }
        # What does bound the loop: vector-instruction issue.  Wave-instructions per launch from the
        # committed PMC pass, 2 cycles each on a SIMD, over the SIMD-cycles of the launch measured here.
        valu = prof.get("valu_insts_per_launch")
        if valu:
            prop = torch.cuda.get_device_properties(dev)
            simd_cycles = prop.multi_processor_count * 4 * avg_kernel_ms * 1e-3 * SHADER_CLOCK_HZ
            out["roofline"]["valu_issue_frac"] = round(2.0 * valu / simd_cycles, 3)
            out["roofline"]["lane_utilisation"] = round(prof.get("lane_utilisation") or 0.0, 3)
        # traffic, valu_issue_frac and lane_utilisation are not measured in this run: they come from the committed
        # rocprofv3 PMC passes over this same workload, and only while the kernel sources are the ones profiled
        out["roofline"]["pmc_source"] = "profiles/hbm_traffic.json" if prof else None
    ctx.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The CPU oracle (scalar restatement of the reference's algorithm) on this box's host cores,
        # on a bounded sample of the same workload; it is the checker elsewhere, here only timed.
        orc = load_oracle()
        # Every core this process may run on (the affinity mask, further limited by a cgroup CPU quota when
        # there is one): os.cpu_count() names the whole machine, of which a GPU box hands one job a share.
        host_cpus = os.cpu_count() or 1
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, period = f.read().split()[:2]
                if q != "max":
                    quota = max(1, int(int(q) / int(period)))
        except (OSError, ValueError):
            pass
        threads = max(1, min(usable, quota) if quota else usable)
        cpu_spp = 16
        o = orc.OracleScene()
        scenes.cover_scene(o, W, H, wrap_bvh=True)
        t0 = time.perf_counter()
        o.render(W, H, 0, cpu_spp, MAX_DEPTH, rng_mode=1, order=1, threads=threads)
        dt = time.perf_counter() - t0
        # One thread, with and without the BVH (the reference application renders the flat list),
        # on a smaller frame of the same scene: the per-core figures behind the headline one.
        sw, sh, sspp = 480, 270, 2
        t0 = time.perf_counter()
        o.render(sw, sh, 0, sspp, MAX_DEPTH, rng_mode=1, order=1, threads=1)
        dt_bvh1 = time.perf_counter() - t0
        flat = orc.OracleScene()
        scenes.cover_scene(flat, W, H, wrap_bvh=False)
        t0 = time.perf_counter()
        flat.render(sw, sh, 0, 1, MAX_DEPTH, rng_mode=1, order=1, threads=1)
        dt_flat1 = time.perf_counter() - t0
        # The reference's own CPU-runnable case: 1200x675 at 10 spp, written as image.ppm, on all the
        # threads and on two fewer (RaytracerMT starts hardware_concurrency() - 2 workers,
        # R/src/Core/Raytracer.cpp:136).
        cw, ch, cspp = 1200, 675, 10
        c1 = orc.OracleScene()
        scenes.cover_scene(c1, cw, ch, wrap_bvh=True)
        t0 = time.perf_counter()
        _, c1_rgba8, _ = c1.render(cw, ch, 0, cspp, MAX_DEPTH, rng_mode=1, order=1, threads=threads)
        dt_c1 = time.perf_counter() - t0
        ppm = os.path.join(tempfile.gettempdir(), "image.ppm")
        rt.write_ppm(ppm, c1_rgba8)
        ppm_bytes = os.path.getsize(ppm)
        t2 = max(1, threads - 2)   # hardware_concurrency() - 2, R/src/Core/Raytracer.cpp:136
        t0 = time.perf_counter()
        c1.render(cw, ch, 0, cspp, MAX_DEPTH, rng_mode=1, order=1, threads=t2)
        dt_c1_t2 = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(W * H * cpu_spp / dt / 1e6, 3),
            "unit": "Msamples/s",
            "cores": threads,
            "host_cpus": host_cpus,
            "usable_cpus": {"affinity": usable, "cgroup_quota": quota},
            "kind": "port",
            "sample": f"same scene and camera, {W}x{H} at {cpu_spp} spp ({W * H * cpu_spp / 1e6:.1f} M samples), "
                      f"scalar CPU oracle with BVH, {threads} threads, {dt:.1f} s wall",
            "one_thread_bvh": round(sw * sh * sspp / dt_bvh1 / 1e6, 4),
            "one_thread_flat_list": round(sw * sh / dt_flat1 / 1e6, 4),
            "one_thread_sample": f"{sw}x{sh}, {sspp} spp (BVH) and 1 spp (flat list, as the reference application renders)",
            "c1_1200x675x10": {"msamples_per_s": round(cw * ch * cspp / dt_c1 / 1e6, 3), "threads": threads,
                               "seconds": round(dt_c1, 3), "image_ppm_bytes": ppm_bytes},
            "c1_threads_minus_2": {"msamples_per_s": round(cw * ch * cspp / dt_c1_t2 / 1e6, 3), "threads": t2,
                                   "seconds": round(dt_c1_t2, 3)},
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
