"""Differential fuzzing of the HIP path (through the C ABI) against the CPU restatement: random small scenes of four
kinds, cameras (jittered, on round coordinates, looking along an axis, rolled), render parameters, frame counts and batches
of awkward rays (axis-parallel, from round coordinates, from vertices towards vertices, along box planes, tiny and huge
directions), under random settings of the traversal's scheduling knobs (which kernel takes small queues, stack depth, the
allowance before a whole wave takes a ray over, when waves leave the queue, refill and leaf-phase thresholds, chunk size),
none of which may change a result.  Primary-hit ids and t/alpha/beta in the culled and the exhaustive mode, accumulators
over one to three frames and the hits of the ray batches are compared bit for bit.

    python tests/fuzz_parity.py <first seed> <count> [seconds]      # needs an MI355X
    python tests/fuzz_parity.py --cpu <first seed> <count> [seconds]   # the restatement against the reference's own compiled
                                                                     # sources (oracle/_ref) on the same scenes: no GPU
    python tests/fuzz_parity.py --big <first seed> <count> [seconds]   # frames of up to 1024x640 over up to 400 k triangles:
                                                                     # production against binary-tree traversal, CPU on blocks
    python tests/fuzz_parity.py --batching <first seed> <count> [seconds]   # one image however the call is cut up: frames per
                                                                     # call, arena limits, batch sizes, tile shares, tile runs

tests/test_gpu_fuzz.py runs a fixed range of seeds with the GPU suite; campaigns over seeds 100..4099, 5000..10999, 20000..32503, 40000..46621 and
60000..72567 (an hour on one MI355X) found no difference, nor did seeds 100..2099 and 3000..7670 with --cpu, 0..399 and 1000..1473 with --big or 0..599 and
1000..1799 with --batching."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import binding  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402
from util import bits, load_into  # noqa: E402

KNOBS = ("RTX_SMALL_QUEUE", "RTX_WIDE_STACK", "RTX_LONG_VISITS", "RTX_RETIRE_BACKLOG", "RTX_TRACE_REFILL", "RTX_TRACE_TRI_BATCH",
         "RTX_TRACE_CHUNK")


def differ(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    neq = bits(a) != bits(b)
    neq &= ~(np.isnan(a) & np.isnan(b))
    return int(neq.sum())


def scene_of(seed):
    """The scene, camera and render parameters of `seed`; also the generator, for what the caller draws further."""
    rng = np.random.default_rng(1000 + seed)
    kind = int(rng.integers(0, 4))
    w, h = int(rng.integers(3, 90)), int(rng.integers(3, 70))
    depth = int(rng.integers(1, 6))
    nd, ns = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    if kind <= 1:
        sc = scenes.mixed_scene(seed=seed, width=w, height=h, max_depth=depth, n_diff=nd, n_spec=ns, with_sky=bool(rng.integers(0, 2)),
                                ior=float(rng.uniform(1.05, 2.2)), with_tex=bool(rng.integers(0, 2)), detail=float(rng.uniform(0.3, 1.6)))
    elif kind == 2:
        sc = scenes.atrium_scene(width=w, height=h, max_depth=min(depth, 4), n_diff=min(nd, 2), n_spec=min(ns, 2),
                                 detail=float(rng.uniform(0.08, 0.3)), tex_size=32, seed=seed)
    else:
        sc = scenes.heightfield_scene(n_tris=int(rng.integers(200, 6000)), width=w, height=h, seed=seed)
        sc.max_depth, sc.n_diff, sc.n_spec = depth, min(nd, 2), min(ns, 2)
    # camera: eye and target jittered, sometimes on round coordinates, sometimes looking along an axis, sometimes rolled
    eye = np.array(sc.eye, np.float32) + rng.normal(0, 3, 3).astype(np.float32)
    at = np.array(sc.at, np.float32) + rng.normal(0, 3, 3).astype(np.float32)
    view = int(rng.integers(0, 5))
    if view == 0:
        eye, at = np.round(eye), np.round(at)
    elif view == 1:
        at = eye.copy()
        at[int(rng.integers(0, 3))] += np.float32(rng.choice([-20, 20]))
    sc.eye, sc.at = tuple(float(x) for x in eye), tuple(float(x) for x in at)
    sc.focal = float(rng.uniform(15, 120))
    sc.rotate = float(rng.choice([0.0, 0.0, rng.uniform(-180, 180)]))
    sc.seed = int(rng.integers(0, 2**31))
    if rng.integers(0, 3) == 0:
        sc.bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    return sc, rng, kind


def awkward_rays(sc, rng, nr=6000):
    """Rays from inside the scene's bounds with the awkward ones mixed in."""
    V = sc.tris["v"].reshape(-1, 4)[:, :3].astype(np.float32)
    o = rng.uniform(V.min(0) - 1, V.max(0) + 1, (nr, 3)).astype(np.float32)
    d = rng.normal(0, 1, (nr, 3)).astype(np.float32)
    d[:300, int(rng.integers(0, 3))] = 0
    a, b = rng.choice(3, 2, replace=False)
    d[300:500, a] = 0
    d[300:500, b] = 0
    o[500:1500] = np.round(o[500:1500])
    d[1500:1600] *= np.float32(1e-6)
    d[1600:1700] *= np.float32(1e6)
    k = rng.integers(0, len(V), 600)
    k2 = rng.integers(0, len(V), 600)
    o[1700:2300] = V[k]                      # from vertices
    d[1700:2300] = V[k2] - V[k]              # towards other vertices: edge and corner hits
    o[2300:2600, 0] = V[rng.integers(0, len(V), 300), 0]   # on x planes of boxes, running along them
    d[2300:2600, 0] = 0
    return o, d


def one_scene(seed):
    """Builds the scene of `seed`, renders and traces it on both sides; returns (description, list of differences)."""
    sc, rng, kind = scene_of(seed)
    w, h, depth = sc.width, sc.height, sc.max_depth
    # scheduling knobs of the traversal, read when the scene object is created
    env = {}
    if rng.integers(0, 4) != 0:
        env["RTX_SMALL_QUEUE"] = "0"           # small queues through the one-ray-per-lane kernel, not the wave-per-ray one
    env["RTX_WIDE_STACK"] = str(rng.choice([16, 24]))
    allowance = int(rng.choice([-1, 0, 3, 12]))
    if allowance >= 0:
        env["RTX_LONG_VISITS"] = str(allowance)
    env["RTX_RETIRE_BACKLOG"] = str(rng.choice([0, 128, 1000]))
    env["RTX_TRACE_REFILL"] = str(rng.choice([1, 8, 16, 40, 64]))
    env["RTX_TRACE_TRI_BATCH"] = str(rng.choice([1, 12, 28, 50, 64]))
    env["RTX_TRACE_CHUNK"] = str(rng.choice([64, 256]))
    saved = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        rt = api.RayTracer(0)
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    O = load_into(binding.Oracle(), sc)
    rt.load_scene(sc)
    msg = []
    ids_o, tab_o = O.primary_hits()
    for m in (0, 1):
        ids_g, tab_g = rt.primary_hits(m)
        if not np.array_equal(ids_o, ids_g):
            msg.append("primary ids, mode %d: %d differ" % (m, int((ids_o != ids_g).sum())))
        else:
            hit = ids_o >= 0
            n = differ(tab_o[hit], tab_g[hit])
            if n:
                msg.append("primary t/alpha/beta, mode %d: %d values differ" % (m, n))
    frames = int(rng.integers(1, 4))
    want = np.zeros((sc.height, sc.width, 4), np.float32)
    O.reset_counts()
    for f in range(1, frames + 1):
        O.render_rows(f, want, nthreads=16)
    rt.reset()
    rays = 0
    if rng.integers(0, 2):
        rt.render(1, frames)
        rays = rt.stats()["rays"]
    else:
        for f in range(1, frames + 1):
            rt.render(f, 1)
            rays += rt.stats()["rays"]
    n = differ(want[..., :3], rt.accum()[..., :3])
    if n:
        msg.append("accumulator: %d values differ" % n)
    if rays != O.ray_count():
        msg.append("rays: %d against %d" % (rays, O.ray_count()))
    st = rt.stats()
    o, d = awkward_rays(sc, rng)
    ids_ro, tab_ro = O.closest_hits(o, d)
    for m in (0, 1):
        ids_rg, tab_rg = rt.trace_rays(o, d, m)
        if not np.array_equal(ids_ro, ids_rg):
            msg.append("ray batch, mode %d: %d ids differ" % (m, int((ids_ro != ids_rg).sum())))
        else:
            hit = ids_ro >= 0
            n = differ(tab_ro[hit], tab_rg[hit])
            if n:
                msg.append("ray batch t/alpha/beta, mode %d: %d values differ" % (m, n))
    what = "seed %d kind %d %dx%d depth %d diffuse %d specular %d, %d triangles, %d frame(s), %d rays by whole waves, %d set aside, %s" % (
        seed, kind, w, h, depth, sc.n_diff, sc.n_spec, len(sc.tris), frames, st["rays_wave_traced"], st["rays_set_aside"],
        " ".join("%s=%s" % (k[4:], v) for k, v in env.items()))
    return what, msg


def one_scene_cpu(seed):
    """The CPU restatement against the reference's own compiled sources (oracle/_ref, where it has been built) on the scene
    of `seed`: camera, tree, rays, closest hits of the primary rays and of the awkward batch, candidate lists, radiance of the
    recursive color() with the reference's generators."""
    from scene_io import nodes_fields
    sc, rng, kind = scene_of(seed)
    sc.width, sc.height = min(sc.width, 24), min(sc.height, 16)   # the reference traces every ray on a thread of its own here
    O, R = load_into(binding.Oracle(), sc), load_into(binding.Reference(), sc)
    msg = []
    if differ(O.get_camera(), R.get_camera()):
        msg.append("camera")
    if not np.array_equal(nodes_fields(O.get_nodes()), nodes_fields(R.get_nodes())):
        msg.append("tree")
    rays_o = np.array([O.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)
    rays_r = np.array([R.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)
    if differ(rays_o, rays_r):
        msg.append("camera rays")
    ids_o, tab_o = O.primary_hits()
    ids_r, tab_r = R.closest_hits(rays_r[:, 0, :3], rays_r[:, 1, :3])
    if not np.array_equal(ids_o, ids_r) or differ(tab_o, tab_r):
        msg.append("primary hits")
    o, d = awkward_rays(sc, rng, 3000)
    ids_o, tab_o = O.closest_hits(o, d)
    ids_r, tab_r = R.closest_hits(o, d)
    if not np.array_equal(ids_o, ids_r) or differ(tab_o, tab_r):
        msg.append("ray batch: %d ids differ" % int((ids_o != ids_r).sum()))
    for i in range(0, len(o), 211):
        if not np.array_equal(O.bvh_hit(o[i], d[i]), R.bvh_hit(o[i], d[i])):
            msg.append("candidate list of ray %d" % i)
            break
    col_o, tir = O.color_mt(0, rays_r[:, 0, :3], rays_r[:, 1, :3], fresh_each=True)
    col_r = R.color(0, rays_r[:, 0, :3], rays_r[:, 1, :3], fresh_each=True)
    ok = tir == 0
    n = differ(col_o[ok][:, :3], col_r[ok][:, :3])
    if n:
        msg.append("radiance: %d values differ" % n)
    what = "seed %d kind %d %dx%d depth %d diffuse %d specular %d, %d triangles, %d samples past total internal reflection left out" % (
        seed, kind, sc.width, sc.height, sc.max_depth, sc.n_diff, sc.n_spec, len(sc.tris), int((~ok).sum()))
    return what, msg


def one_scene_big(seed):
    """Larger frames (the one-ray-per-lane kernel with many chunks per wave, the queues of every level in the millions): an
    atrium or a displaced mesh of up to a few hundred thousand triangles at up to 1024x640.  The production traversal against
    the binary-tree kernel over the reference's own tree (RTX_TRACE_BINARY=1: every box with the exact predicate), whole
    accumulators and ray counts, and against the CPU restatement on pixel blocks scattered over the frame."""
    from util import scattered_blocks
    rng = np.random.default_rng(77000 + seed)
    w, h = int(rng.integers(300, 1025)), int(rng.integers(200, 641))
    depth = int(rng.integers(2, 5))
    if rng.integers(0, 2):
        sc = scenes.atrium_scene(width=w, height=h, max_depth=depth, detail=float(rng.uniform(0.2, 1.0)), tex_size=64, seed=seed)
    else:
        sc = scenes.heightfield_scene(n_tris=int(rng.integers(20000, 400000)), width=w, height=h, seed=seed)
        sc.max_depth, sc.n_diff, sc.n_spec = depth, 1, 1
    eye = np.array(sc.eye, np.float32) + rng.normal(0, 2, 3).astype(np.float32)
    at = np.array(sc.at, np.float32) + rng.normal(0, 2, 3).astype(np.float32)
    sc.eye, sc.at = tuple(float(x) for x in eye), tuple(float(x) for x in at)
    sc.focal = float(rng.uniform(20, 90))
    sc.seed = int(rng.integers(0, 2**31))
    nodes = api.build_bvh(sc.tris)
    frames = int(rng.integers(1, 3))
    env = {"RTX_RETIRE_BACKLOG": str(rng.choice([0, 128, 400])), "RTX_WIDE_STACK": str(rng.choice([16, 24])),
           "RTX_TRACE_CHUNK": str(rng.choice([64, 256]))}
    got = {}
    for binary in (False, True):
        saved = {k: os.environ.get(k) for k in KNOBS + ("RTX_TRACE_BINARY",)}
        os.environ.update(env)
        if binary:
            os.environ["RTX_TRACE_BINARY"] = "1"
        try:
            rt = api.RayTracer(0)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        rt.load_scene(sc, nodes)
        rt.reset()
        rt.render(1, frames)
        st = rt.stats()
        got[binary] = (rt.accum(), st["rays"], st["rays_per_level"])
        del rt
    msg = []
    n = differ(got[False][0], got[True][0])
    if n:
        msg.append("accumulator, production against binary-tree traversal: %d values differ" % n)
    if got[False][1:] != got[True][1:]:
        msg.append("ray counts: %s against %s" % (got[False][2], got[True][2]))
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    xy = scattered_blocks(sc.width, sc.height, 24, seed=seed)
    acc = np.zeros(got[False][0].shape, np.float32)
    for f in range(1, frames + 1):
        O.render_pixels(f, xy, acc, nthreads=16)
    n = differ(acc[xy[:, 1], xy[:, 0], :3], got[False][0][xy[:, 1], xy[:, 0], :3])
    if n:
        msg.append("accumulator against the CPU on %d pixels: %d values differ" % (len(xy), n))
    what = "seed %d %s %dx%d depth %d, %d triangles, %d frame(s), %d rays, %s" % (
        seed, sc.name, w, h, depth, len(sc.tris), frames, got[False][1], " ".join("%s=%s" % (k[4:], v) for k, v in env.items()))
    return what, msg


def one_scene_batching(seed):
    """How a call is cut up must not show in the image: frames one call each with the default work arena are the baseline
    (its first frame checked against the CPU restatement on scattered blocks); then all frames in one call (merged pixel
    batches), a work arena between tight and ample (many batches, queue capacities that have to grow), queues sized for the
    worst case, batches limited to a random number of pixel blocks, the image composed from n interleaved tile shares or
    from runs of consecutive tiles rendered one after the other, the 8-bit images of the frames of a merged call."""
    from util import scattered_blocks
    rng = np.random.default_rng(31000 + seed)
    w, h = int(rng.integers(70, 420)), int(rng.integers(60, 300))
    depth = int(rng.integers(1, 5))
    nd, ns = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    if rng.integers(0, 2):
        sc = scenes.atrium_scene(width=w, height=h, max_depth=depth, n_diff=nd, n_spec=ns, detail=float(rng.uniform(0.1, 0.4)), tex_size=32, seed=seed)
    else:
        sc = scenes.mixed_scene(seed=seed, width=w, height=h, max_depth=depth, n_diff=nd, n_spec=ns, with_sky=bool(rng.integers(0, 2)),
                                detail=float(rng.uniform(0.5, 1.5)))
    sc.seed = int(rng.integers(0, 2**31))
    nodes = api.build_bvh(sc.tris)
    frames = int(rng.integers(1, 5))
    first = int(rng.integers(1, 50))
    msg = []

    def tracer(env=None):
        saved = {k: os.environ.get(k) for k in ("RTX_MERGE_BLOCKS", "RTX_WORST_CASE_QUEUES")}
        os.environ.update(env or {})
        try:
            rt = api.RayTracer(0)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        rt.load_scene(sc, nodes)
        rt.reset()
        return rt

    rt = tracer()
    images = []
    for f in range(first, first + frames):
        rt.render(f, 1)
        images.append(api.tonemap(rt.accum(), f - first + 1) if first == 1 else None)
    base = rt.accum()
    rays = None
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    xy = scattered_blocks(sc.width, sc.height, 12, seed=seed)
    acc = np.zeros(base.shape, np.float32)
    for f in range(first, first + frames):
        O.render_pixels(f, xy, acc, nthreads=16)
    n = differ(acc[xy[:, 1], xy[:, 0], :3], base[xy[:, 1], xy[:, 0], :3])
    if n:
        msg.append("baseline against the CPU on %d pixels: %d values differ" % (len(xy), n))

    def check(what, a):
        n = differ(base, a)
        if n:
            msg.append("%s: %d values differ" % (what, n))

    rt.reset()
    rt.render(first, frames)
    check("one call over %d frames" % frames, rt.accum())
    # a tight work arena: a few thousand bytes per sample of one 4x4-pixel block upwards
    per_block = 16 * 4 * 3000
    arena = int(per_block * rng.uniform(1.0, 40.0))
    rt.set_arena_bytes(arena)
    rt.reset()
    try:
        rt.render(first, frames)
        st = rt.stats()
        check("work arena of %d bytes (%d batches, %d retries)" % (arena, st["chunks"], st["queue_retries"]), rt.accum())
    except api.RtxError as e:
        if "arena" not in str(e):
            msg.append("work arena of %d bytes: %s" % (arena, e))
    rt.set_arena_bytes(0)
    del rt
    rt = tracer({"RTX_WORST_CASE_QUEUES": "1", "RTX_MERGE_BLOCKS": str(int(rng.integers(1, 3000)))})
    rt.render(first, frames)
    check("worst-case queues, limited batches", rt.accum())
    if first == 1:
        rt.reset()
        got = rt.render_images(1, frames)
        for k in range(frames):
            if not np.array_equal(got[k].reshape(-1, 3), images[k]):
                msg.append("8-bit image of frame %d of a merged call" % (k + 1))
    # tile shares
    n_sh = int(rng.choice([2, 3, 5]))
    merged = np.zeros_like(base)
    tiles_x, tiles_y = (sc.width + 63) // 64, (sc.height + 63) // 64
    for r in range(n_sh):
        rt.set_tiles(r, n_sh, -1)
        rt.reset()
        rt.render(first, frames)
        part = rt.accum()
        for t in range(r, tiles_x * tiles_y, n_sh):
            y0, x0 = (t // tiles_x) * 64, (t % tiles_x) * 64
            merged[y0:y0 + 64, x0:x0 + 64] = part[y0:y0 + 64, x0:x0 + 64]
            part[y0:y0 + 64, x0:x0 + 64] = 0
        if part.any():
            msg.append("share %d of %d wrote outside its tiles" % (r, n_sh))
    check("%d interleaved tile shares" % n_sh, merged)
    # runs of consecutive tiles into one accumulator, frame by frame (what dynamic balancing does)
    rt.reset()
    for f in range(first, first + frames):
        t = 0
        while t < tiles_x * tiles_y:
            cnt = int(rng.integers(1, 6))
            rt.set_tiles(t, 1, cnt)
            rt.render(f, 1)
            t += cnt
    rt.set_tiles(0, 1, -1)
    check("runs of consecutive tiles", rt.accum())
    what = "seed %d %s %dx%d depth %d diffuse %d specular %d, %d triangles, frames %d..%d" % (
        seed, sc.name, w, h, depth, nd, ns, len(sc.tris), first, first + frames - 1)
    return what, msg


def run(first, count, seconds=1e9, verbose=False, scene=None):
    """Scenes of seeds first .. first+count-1 (fewer when `seconds` run out); returns the descriptions of those that differ."""
    t0 = time.time()
    failed = []
    for seed in range(first, first + count):
        if time.time() - t0 > seconds:
            break
        what, msg = (scene or one_scene)(seed)
        if verbose:
            print(what + ": " + ("ok" if not msg else "; ".join(msg)), flush=True)
        if msg:
            failed.append(what + ": " + "; ".join(msg))
    return failed


if __name__ == "__main__":
    cpu = "--cpu" in sys.argv   # the restatement against the compiled reference instead (no GPU)
    big = "--big" in sys.argv   # larger frames: production against binary-tree traversal, and the CPU on scattered blocks
    batching = "--batching" in sys.argv   # how a call is cut into batches, shares and frames must not show in the image
    args = [a for a in sys.argv[1:] if a not in ("--cpu", "--big", "--batching")]
    bad = run(int(args[0]), int(args[1]), float(args[2]) if len(args) > 2 else 1e9, verbose=True,
              scene=one_scene_cpu if cpu else one_scene_big if big else one_scene_batching if batching else None)
    print("done: %d scenes differ" % len(bad), flush=True)
    sys.exit(1 if bad else 0)
