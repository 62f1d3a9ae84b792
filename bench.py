"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A frame is one iteration of the loop at /root/reference/src/RayTracer.cpp:234 (four samples per pixel through the
whole bounce loop, accumulated) of the Sponza-class procedural atrium (~263k triangles, textures, cube-map sky),
max_recursion_depth 4, one diffuse and one specular ray per hit: 1920x1080 on one GPU, 3840x2160 (the multi-GPU
configuration) on several.  The scene, textures and BVH are resident on the GPU before the timed region.
On one GPU a step is --frames-per-step consecutive frames (default 4: a quarter of the 16 frames = 64 samples per
pixel of the 1080p configuration) handed to the library in one call, which renders them in one merged pixel batch
-- every frame's contribution is still added to the accumulator in frame order, so each frame's image can be had
(rtx_render_frames_tonemapped, what the command line uses); the time of single-frame calls is measured beside it
and reported as "single_frame_calls".  With N > 1 the image tiles are interleaved over the ranks and a step
is N consecutive frames, so every rank keeps the work of one whole frame per step (weak scaling: its 1/N tile share
of N frames, which the library renders as one merged pixel batch); --scaling strong keeps one frame per step
instead, and whichever of the two is not the headline is timed as well and reported under "other_scaling".
Before the timed loops rank 0 renders three whole frames of the same workload alone ("single_rank_same_workload"),
so that the line carries its own one-GPU anchor at the resolution it was measured at.
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the 8 XCDs
FP32_VECTOR_PEAK_TFLOPS = 157.3   # 256 CUs x 128 FP32 FMA lanes x 2 flops x 2.4 GHz


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


def offline_profile():
    """What profiles/summarize.py recorded from the rocprofv3 counter passes of this same command
    (raytracer_amd/tools/collect_profile.sh) into raytracer_amd/measured.json: lane use and busy share of the
    vector ALUs, fabric bytes per launch, the FETCH_SIZE calibration, and the commit it was taken at.  Counters
    cannot be read from inside this process, so these figures are quoted, with their source, not re-measured."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracer_amd", "measured.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--frames-per-step", type=int, default=None,
                    help="one GPU: frames per call of the library (default 4); several GPUs: fixed by --scaling")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=None, help="default: 1920 on one GPU, 3840 on several")
    ap.add_argument("--height", type=int, default=None, help="default: 1080 on one GPU, 2160 on several")
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--balance", choices=["static", "dynamic"], default="static")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: a step is N frames on N GPUs (fixed work per GPU); strong: a step is one frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame-calls", action="store_true",
                    help="leave out the loop of single-frame calls (profiling runs: every launch of the run then is a launch of the timed workload)")
    ap.add_argument("--no-instrument", action="store_true")
    args = ap.parse_args()
    args.size_given = args.width is not None or args.height is not None
    args.width = args.width or 1920
    args.height = args.height or 1080

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

    if world > 1 and not args.size_given:
        args.width, args.height = 3840, 2160   # the multi-GPU configuration is quoted at 4K
    sc = scenes.atrium_scene(width=args.width, height=args.height, max_depth=args.depth)
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(local)
    rt.set_stream(torch.cuda.current_stream().cuda_stream)
    rt.load_scene(sc, nodes)
    sharder = TileSharder(sc.width, sc.height, rank, world)
    accum = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device=dev)

    # With several GPUs: what one rank alone needs for one frame of this same workload, measured by rank 0 before the
    # timed loops while the others wait, so that the scaling efficiency can be read off this one line.
    single_rank = None
    if world > 1:
        if rank == 0:
            rt.set_accum_device(accum.data_ptr())
            rt.set_tiles(0, 1, -1)
            for f in (1, 2):
                rt.render(f, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rays1 = 0
            for f in (3, 4, 5):
                rt.render(f, 1)
                rays1 += rt.stats()["rays"]
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            single_rank = {"ms_per_frame": round(dt1 / 3 * 1e3, 3), "value": round(rays1 / dt1 / 1e6, 2), "unit": "Mrays/s",
                           "note": "rank 0 alone, whole %dx%d frames, before the timed loops" % (sc.width, sc.height)}
            accum.zero_()
        dist.barrier()

    # dynamic balancing: rank 0 renders into the accumulator, the others into a frame buffer that is empty between frames
    frame_buf = torch.zeros_like(accum) if (args.balance == "dynamic" and rank != 0) else None
    rt.set_accum_device((frame_buf if frame_buf is not None else accum).data_ptr())
    rt.clear_tile_log()
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
    if world == 1 and args.balance == "static":
        frames_per_step = max(1, args.frames_per_step or 4)

    def step(frame, nframes):
        if args.balance == "static":
            rt.render(frame, nframes)
            take_stats()
            sharder.gather_static(accum)
        else:
            if store is None:
                rt.set_tiles(0, 1, -1)
                rt.render(frame, 1)
                take_stats()
            else:
                for first, count in sharder.dynamic_batches(store, frame, batch):
                    rt.set_tiles(first, 1, count)
                    rt.render(frame, 1)
                    take_stats()
            tiles = rt.tile_log()
            rt.clear_tile_log()
            sharder.exchange_dynamic(tiles, frame_buf, accum)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    state = {"frame": 1}

    def timed_loop(nframes, warmup, steps):
        """`warmup` untimed steps, then `steps` timed ones between barriers; the time is the slowest rank's, the
        counters are summed over the ranks."""
        for _ in range(warmup):
            step(state["frame"], nframes)
            state["frame"] += nframes
        for k in counters:
            counters[k] = 0
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(state["frame"], nframes)
            state["frame"] += nframes
        sync()
        dt = time.perf_counter() - t0
        tot = torch.tensor([dt, float(counters["rays"]), counters["trace_ms"], counters["shade_ms"], float(counters["untraced"]),
                            float(counters["launches"])], dtype=torch.float64, device=dev)
        if world > 1:
            tmax = tot.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            tot[0] = tmax[0]
        return {"dt": float(tot[0]), "rays": float(tot[1]), "trace_ms": float(tot[2]), "shade_ms": float(tot[3]),
                "untraced": float(tot[4]), "launches": float(tot[5])}

    main_run = timed_loop(frames_per_step, args.warmup, args.steps)
    dt, rays_total, trace_ms_sum, shade_ms_sum = main_run["dt"], main_run["rays"], main_run["trace_ms"], main_run["shade_ms"]
    # With several GPUs the other way of scaling is timed as well: one frame per step, every rank its 1/N tile
    # share of it (what the per-frame out_NNN.png contract of the reference costs).
    other_run = None
    single_calls = None
    if world == 1 and frames_per_step > 1 and not args.no_single_frame_calls:
        # one frame per call, as a caller that needs the accumulator between frames would render
        single_calls = timed_loop(1, 2, max(4, min(args.steps, 8)))
    if world > 1 and args.balance == "static":
        other_frames = 1 if frames_per_step == world else world
        other_run = timed_loop(other_frames, min(args.warmup, 2), args.steps)
        other_run["frames_per_step"] = other_frames

    # node/triangle test counts of this rank's share, from one instrumented frame outside the timed region
    # (the last frames again: identical rays); they feed the algorithmic-bytes figure of the trace kernel.
    n_node = n_tri = n_wide = n_leaf = None
    ref_counts = None
    if not args.no_instrument and args.balance == "static":
        rt.set_instrumented(True)
        rt.render(state["frame"] - frames_per_step, frames_per_step)
        st = rt.stats()
        n_node = st["node_tests"] / max(st["rays"], 1)
        n_tri = st["tri_tests"] / max(st["rays"], 1)
        n_wide = st["wide_visits"] / max(st["rays"], 1)
        n_leaf = st["leaf_fetches"] / max(st["rays"], 1)
        if rank == 0 and world == 1:
            # What the reference's own traversal does per ray, for context: BVH::hit visits every node whose box the
            # ray meets and hands every such leaf to Triangle::hit, without ordering or culling by the best hit
            # (/root/reference/src/BVH.cpp:162-192).  Counted by the exhaustive mode of the binary-tree kernel
            # (that same candidate set) on the primary rays, beside the production traversal on the same rays.
            rt.primary_hits(1, want_tab=False)
            ex = rt.stats()
            rt.primary_hits(0, want_tab=False)
            cu = rt.stats()
            ref_counts = {"rays": "primary (%d)" % ex["rays"],
                          "box_tests_per_ray": round(ex["node_tests"] / max(ex["rays"], 1), 2),
                          "triangle_tests_per_ray": round(ex["tri_tests"] / max(ex["rays"], 1), 2),
                          "this_path_box_tests_per_ray": round(cu["node_tests"] / max(cu["rays"], 1), 2),
                          "this_path_triangle_tests_per_ray": round(cu["tri_tests"] / max(cu["rays"], 1), 2)}
        rt.set_instrumented(False)

    if rank == 0:
        samples = sc.width * sc.height * 4 * args.steps * frames_per_step
        out = {
            "metric": "Mrays/s (incl. secondary) at %s" % ("1080p" if (sc.width, sc.height) == (1920, 1080) else "%dx%d" % (sc.width, sc.height)),
            "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if (frames_per_step == world or world == 1) else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "atrium (Sponza-class, %d triangles, %d textures + cube-map sky) %dx%d, 4 spp per frame, "
                                   "max_recursion_depth %d, diffuse 1, specular 1" % (len(sc.tris), len(sc.textures), sc.width, sc.height, sc.max_depth),
                       "frames_per_step": frames_per_step,
                       "backend": "rccl" if backend == "nccl" else backend,
                       "sharding": "tile-interleave x%d, %s" % (world, "gather to rank 0 per step" if args.balance == "static" else "dynamic batches from a shared counter + exchange of the rendered tiles per step")},
            "samples_per_s": round(samples / dt, 1),
            "rays_per_step": round(rays_total / args.steps),
            # rays of the last level that are worth zero hit or miss are counted in `value` (they are color() queries the
            # reference traces) but are answered without traversal; the rate over the traversed rays alone stands beside it
            "rays_untraced_per_step": round(main_run["untraced"] / args.steps),
            "traversed_Mrays_per_s": round((rays_total - main_run["untraced"]) / dt / 1e6, 2),
            "gpu_kernel_ms_per_step": {"trace": round(trace_ms_sum / args.steps / world, 3), "shade_gather_accumulate": round(shade_ms_sum / args.steps / world, 3)},
        }
        if world == 1:
            out["ms_per_frame"] = round(dt / args.steps / frames_per_step * 1e3, 3)
        if single_calls is not None:
            steps1 = max(4, min(args.steps, 8))
            out["single_frame_calls"] = {"ms_per_frame": round(single_calls["dt"] / steps1 * 1e3, 3),
                                         "value": round(single_calls["rays"] / single_calls["dt"] / 1e6, 2), "unit": "Mrays/s"}
        if single_rank is not None:
            out["single_rank_same_workload"] = single_rank
        if other_run is not None:
            o = other_run
            out["other_scaling"] = {"scaling": "strong" if o["frames_per_step"] == 1 else "weak", "frames_per_step": o["frames_per_step"],
                                    "value": round(o["rays"] / o["dt"] / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(o["dt"] / args.steps * 1e3, 3),
                                    "ms_per_frame": round(o["dt"] / args.steps / o["frames_per_step"] * 1e3, 3)}
        if n_node is not None:
            # The traversal kernel's work per ray, from the kernels' own counters (one instrumented frame outside the
            # timed region, identical rays): bytes = one 64-B record per node of the 4-wide tree it visits, one 64-B
            # record per triangle whose quantised box it enters, 32 B ray in, 16 B hit out (a ray that is not traversed
            # has only its direction read: 16 B less); flops = 22 per child-box test (6 subtract-multiplies, the interval
            # folds and compares) + 70 per Triangle::hit (two 4-wide dots, the elimination with its divisions).
            # Launch duration: HIP events around every level's traversal on the scene's stream
            # (rtx_stats.trace_ms / trace_launches), live, in the timed region.
            launches = max(main_run["launches"], 1.0)
            avg_launch_ms = trace_ms_sum / launches
            bytes_per_ray = 64.0 * n_wide + 64.0 * n_leaf + 48.0 - 16.0 * main_run["untraced"] / max(rays_total, 1.0)
            flops_per_ray = 22.0 * n_node + 70.0 * n_tri
            algo_gbs = bytes_per_ray * rays_total / (trace_ms_sum * 1e-3) / 1e9 if trace_ms_sum > 0 else 0.0  # per GPU: both sums run over the ranks
            algo_tflops = flops_per_ray * rays_total / (trace_ms_sum * 1e-3) / 1e12 if trace_ms_sum > 0 else 0.0
            prof = offline_profile()
            pk = prof.get("k_trace_wide", {})
            rays_per_launch = rays_total / launches
            traffic = None
            # the per-launch figures of the profile belong to the launches of the default single-GPU workload
            same_launches = world == 1 and (sc.width, sc.height, sc.max_depth) == (1920, 1080, 4) and frames_per_step == prof.get("frames_per_step", 1)
            if not same_launches:
                pk = {k: v for k, v in pk.items() if k in ("lane_util", "issue_model_frac", "valu_class_share")}
            if "fetch_size_bytes_per_launch" in pk:
                # FETCH_SIZE is 64 B x fabric read requests; a request for a random 64-byte record carries 64 B (measured
                # factor 0.98), a streamed line 128 B (factor 2.0): the 32 B per ray of the queue read are added once more
                traffic = round(pk["fetch_size_bytes_per_launch"] + pk["write_size_bytes_per_launch"] + 16.0 * rays_per_launch)
            issued = pk["valu_insts_per_launch"] * 64.0 / rays_per_launch if "valu_insts_per_launch" in pk else None
            roof = {"kernel": "k_trace_wide",
                    # Work-normalised: the algorithmic flops of the launch over its duration against the FP32 vector peak
                    # (256 CUs x 128 FMA lanes x 2 x 2.4 GHz).  The kernel has no matrix work, and its 25 MB of nodes and
                    # triangles are re-read from L2 / Infinity Cache, so neither the MFMA nor the HBM ceiling is the one
                    # it runs into: the vector ALUs' issue rate is (issue_model below).
                    "bound": "valu", "achieved": round(algo_tflops, 2), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(algo_tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
                    "flops_per_ray": round(flops_per_ray, 1), "bytes_per_ray": round(bytes_per_ray, 1),
                    "algorithmic_bytes": {"GBps": round(algo_gbs, 1), "frac_of_hbm_8000": round(algo_gbs / HBM_PEAK_GBS, 4),
                                          "frac_of_l2_34500": round(algo_gbs / L2_PEAK_GBS, 4)},
                    # what the vector ALUs issue against what the algorithm needs: lane-slots (64 x wave instructions) per ray
                    "lane_slots_issued_per_ray": round(issued, 1) if issued else None,
                    "lane_slots_issued_over_algorithmic": round(issued / flops_per_ray, 2) if issued else None,
                    "lane_util": pk.get("lane_util"),
                    # sum over instruction classes of (instructions / measured issue rate of the class) over the launch time:
                    # 1.0 = the launch lasts exactly as long as the vector ALUs need to issue its instructions
                    "issue_model": {"frac_of_launch": pk.get("issue_model_frac"), "model_ms_per_launch": pk.get("issue_model_ms_per_launch"),
                                    "model_ms_bounds": pk.get("issue_model_ms_bounds"), "class_share": pk.get("valu_class_share")},
                    "counters_from": "%s @ %s (rocprofv3 passes of this command, raytracer_amd/tools/collect_profile.sh)" % (prof.get("source"), prof.get("commit")) if prof else None,
                    "traffic": traffic,
                    "traffic_GBps": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic else None,
                    "avg_launch_ms": round(avg_launch_ms, 4),
                    "node_visits_per_ray": round(n_wide, 2), "leaf_fetches_per_ray": round(n_leaf, 3),
                    "box_tests_per_ray": round(n_node, 2), "triangle_tests_per_ray": round(n_tri, 3)}
            if ref_counts:
                roof["reference_traversal_for_context"] = ref_counts
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, nodes)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
