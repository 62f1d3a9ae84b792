"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one frame (one iteration of the loop at /root/reference/src/RayTracer.cpp:234: four
samples per pixel through the whole bounce loop, accumulated) of the Sponza-class procedural atrium
(~263k triangles, textures, cube-map sky) at 1920x1080, max_recursion_depth 4, one diffuse and one
specular ray per hit.  The scene, textures and BVH are resident on the GPU before
the timed region.  With N > 1 the image tiles are interleaved over the ranks and a step is N consecutive
frames, so every rank keeps the work of one whole frame per step (weak scaling: its 1/N tile share of N frames,
which the library renders as one merged pixel batch); --scaling strong keeps one frame per step instead.
There is no collective inside a step's rendering; every step ends with the RCCL gather of the tile
accumulators to rank 0.

Prints one JSON line on rank 0.  `value` is Mrays/s over all ranks: color()-equivalent ray
queries (primary, secondary, last-depth occlusion) divided by the wall time of the K steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def cpu_baseline(sc, nodes, seconds_target=15.0):
    """Times the CPU path on this host's cores over a bounded band of rows of the same frame.
    kind "reference": the reference's own compiled objects (oracle/_ref/libref.so) driven row-parallel;
    kind "port": the CPU restatement (oracle/liboracle.so).  Never part of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding
    from util import load_into
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    port = binding.Oracle()
    load_into(port, sc, build=False)
    port.set_nodes(nodes)
    mid = sc.height // 2
    use_ref = binding.have_ref()
    obj = port
    if use_ref:
        obj = binding.Reference()
        load_into(obj, sc, build=True)

    def timed(o, rows):
        r0 = max(0, mid - rows // 2)
        acc = np.zeros((sc.height, sc.width, 4), np.float32)
        return o.render_rows(1, acc, row0=r0, row1=r0 + rows, nthreads=cores), r0

    probe_rows = cores  # one row per thread, so that the probe sees the parallel rate
    t_probe, _ = timed(obj, probe_rows)
    rows = int(min(sc.height, max(cores // 2, probe_rows * seconds_target / max(t_probe, 1e-3))))
    port.reset_counts()
    t, r0 = timed(obj, rows)
    if use_ref:
        # the reference cannot count its own color() calls; the restatement renders the same rows
        # (same scene, same sampling distributions) and its exact count is used
        timed(port, rows)
    rays = port.ray_count()
    samples = rows * sc.width * 4
    return {
        "value": round(rays / t / 1e6, 4), "unit": "Mrays/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": "rows %d..%d of frame 1 of the same 1080p workload (%d samples, %d rays%s), %.1f s on %d threads" % (
            r0, r0 + rows - 1, samples, rays,
            "; rays counted by the CPU restatement on the same rows" if use_ref else "", t, cores),
        "samples_per_s": round(samples / t, 1),
    }


def measured_traffic():
    """HBM bytes per traversal launch from the rocprofv3 PMC passes (FETCH_SIZE doubled as gfx950 needs, plus
    WRITE_SIZE), as recorded by profiles/summarize.py into raytracer_amd/measured.json; None when absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracer_amd", "measured.json")
    try:
        with open(path) as f:
            return json.load(f).get("k_trace_wide_hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--balance", choices=["static", "dynamic"], default="static")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: a step is N frames on N GPUs (fixed work per GPU); strong: a step is one frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-instrument", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import api, scenes
    from raytracer_amd.dist import TileSharder, default_store

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (launch N>1 through torch.distributed.run)" % (args.gpus, world))
    # RTX_BENCH_BACKEND=gloo rehearses the N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices
    # and the gather travels through the host); the measured configuration is one rank per GPU over RCCL.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    sc = scenes.atrium_scene(width=args.width, height=args.height, max_depth=args.depth)
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(local)
    rt.set_stream(torch.cuda.current_stream().cuda_stream)
    rt.load_scene(sc, nodes)
    sharder = TileSharder(sc.width, sc.height, rank, world)
    accum = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device=dev)
    frame_buf = torch.zeros_like(accum) if args.balance == "dynamic" else None
    rt.set_accum_device((frame_buf if frame_buf is not None else accum).data_ptr())
    if args.balance == "static":
        rt.set_tiles(*sharder.static_share())
    store = default_store() if (world > 1 and args.balance == "dynamic") else None
    batch = max(1, sharder.n_tiles // (world * 8))

    counters = {"rays": 0, "trace_ms": 0.0, "shade_ms": 0.0, "launches": 0, "node_tests": 0, "tri_tests": 0,
                "shaded": 0, "fetches": 0, "untraced": 0}

    def take_stats():
        st = rt.stats()
        counters["rays"] += st["rays"]
        counters["trace_ms"] += st["trace_ms"]
        counters["shade_ms"] += st["shade_ms"]
        counters["launches"] += st["trace_launches"]
        counters["shaded"] += st["shaded_hits"]
        counters["fetches"] += st["texture_fetches"]
        counters["untraced"] += st["rays_untraced"]

    frames_per_step = world if (args.scaling == "weak" and args.balance == "static") else 1

    def step(frame):
        if args.balance == "static":
            rt.render(frame, frames_per_step)
            take_stats()
            sharder.gather_static(accum)
        else:
            frame_buf.zero_()
            torch.cuda.current_stream().synchronize()
            if store is None:
                rt.set_tiles(0, 1, -1)
                rt.render(frame, 1)
                take_stats()
            else:
                for first, count in sharder.dynamic_batches(store, frame, batch):
                    rt.set_tiles(first, 1, count)
                    rt.render(frame, 1)
                    take_stats()
            sharder.reduce_frame(frame_buf, accum)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    frame = 1
    for _ in range(args.warmup):
        step(frame)
        frame += frames_per_step
    for k in counters:
        counters[k] = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(frame)
        frame += frames_per_step
    sync()
    dt = time.perf_counter() - t0

    tot = torch.tensor([dt, float(counters["rays"]), counters["trace_ms"], counters["shade_ms"]], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    rays_total = float(tot[1])
    trace_ms_sum = float(tot[2])
    shade_ms_sum = float(tot[3])

    # node/triangle test counts of this rank's share, from one instrumented frame outside the timed region
    # (frame `frame - 1` again: identical rays); they feed the algorithmic-bytes figure of the trace kernel.
    n_node = n_tri = n_wide = n_leaf = None
    if not args.no_instrument and args.balance == "static":
        rt.set_instrumented(True)
        rt.render(frame - frames_per_step, frames_per_step)
        st = rt.stats()
        rt.set_instrumented(False)
        n_node = st["node_tests"] / max(st["rays"], 1)
        n_tri = st["tri_tests"] / max(st["rays"], 1)
        n_wide = st["wide_visits"] / max(st["rays"], 1)
        n_leaf = st["leaf_fetches"] / max(st["rays"], 1)

    if rank == 0:
        samples = sc.width * sc.height * 4 * args.steps * frames_per_step
        out = {
            "metric": "Mrays/s (incl. secondary) at 1080p", "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if frames_per_step == world else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "atrium (Sponza-class, %d triangles, %d textures + cube-map sky) %dx%d, 4 spp per step, "
                                   "max_recursion_depth %d, diffuse 1, specular 1" % (len(sc.tris), len(sc.textures), sc.width, sc.height, sc.max_depth),
                       "frames_per_step": frames_per_step,
                       "backend": "rccl" if backend == "nccl" else backend,
                       "sharding": "tile-interleave x%d, %s" % (world, "gather to rank 0 per step" if args.balance == "static" else "dynamic batches + reduce per step")},
            "samples_per_s": round(samples / dt, 1),
            "rays_per_step": round(rays_total / args.steps),
            # rays of the last level that are worth zero hit or miss (counted above: the reference traces them);
            # this rank's share
            "rays_untraced_per_step": round(counters["untraced"] / args.steps),
            "gpu_kernel_ms_per_step": {"trace": round(trace_ms_sum / args.steps / world, 3), "shade_gather_accumulate": round(shade_ms_sum / args.steps / world, 3)},
        }
        if n_node is not None:
            # algorithmic bytes the traversal kernel must move per ray: one 64-B record per node of the 4-wide
            # tree it visits, one 64-B record per triangle whose quantised box it enters, 32 B ray in, 16 B hit out
            # (a ray that is not traversed has only its direction read: 16 B less)
            bytes_per_ray = 64.0 * n_wide + 64.0 * n_leaf + 48.0 - 16.0 * counters["untraced"] * world / max(rays_total, 1.0)
            achieved = bytes_per_ray * rays_total / (trace_ms_sum * 1e-3) / 1e9 if trace_ms_sum > 0 else 0.0  # per GPU: both sums run over the ranks
            out["roofline"] = {"bound": "hbm", "kernel": "k_trace_wide", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(),
                               "node_visits_per_ray": round(n_wide, 2), "leaf_fetches_per_ray": round(n_leaf, 3),
                               "box_tests_per_ray": round(n_node, 2), "triangle_tests_per_ray": round(n_tri, 3),
                               "bytes_per_ray": round(bytes_per_ray, 1),
                               "avg_launch_ms": round(trace_ms_sum / max(counters["launches"] * world, 1), 4),
                               "traffic_GBps": (round(measured_traffic() / (trace_ms_sum / max(counters["launches"] * world, 1) * 1e-3) / 1e9, 1)
                                                if measured_traffic() and trace_ms_sum > 0 else None),
                               "note": "the records are re-read from L2/Infinity Cache (tree + triangles = 25 MB), so algorithmic "
                                       "bytes/s can approach the HBM figure without HBM traffic; the kernel's measured limiter is "
                                       "vector-ALU issue (SQ_ACTIVE_INST_VALU ~ 98 % of SIMD cycles), see profiles/"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, nodes)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
