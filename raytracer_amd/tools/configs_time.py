"""Times the five benchmark configurations on one GPU through the C ABI (wall time of the calls, best of a few):

    python raytracer_amd/tools/configs_time.py [c1 c2 c3 c4 c5]

c1  small scene (every material class, textures, sky), 512x512, depth 2, two specular + two diffuse rays per hit
c2  bunny-class closed mesh (69.5 k triangles), 1920x1080, primary hits only (rtx_primary_hits into page-locked arrays)
c3  atrium (263.6 k triangles), 1920x1080, depth 4
c4  atrium, 3840x2160, depth 4 (one GPU renders the whole frame)
c5  displaced mesh of 10.1 M triangles, 3840x2160, depth 8
For c1, c3, c4, c5: one frame per call, and four frames per call (merged pixel batches)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402


def frames(rt, label, n_tris):
    rt.reset()
    rt.render(1, 1)
    rt.render(2, 4)
    out = []
    for nf in (1, 4):
        best, rays = 1e9, 0
        f = 6
        for _ in range(4):
            t0 = time.perf_counter()
            rt.render(f, nf)
            dt = time.perf_counter() - t0
            f += nf
            if dt < best:
                best, rays = dt, rt.stats()["rays"]
        out.append((nf, best / nf * 1e3, rays / nf, rays / best / 1e6))
    w, h = rt.width, rt.height
    for nf, ms, rays, mrays in out:
        print("%s  %d triangles  %dx%d  %d frame(s) per call: %.3f ms per frame, %.2f M rays per frame, %.0f Mrays/s, %.0f Msamples/s" % (
            label, n_tris, w, h, nf, ms, rays / 1e6, mrays, w * h * 4 / ms / 1e3), flush=True)


def main():
    which = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]
    if "c1" in which:
        sc = scenes.mixed_scene(seed=3, width=512, height=512, max_depth=2, n_diff=2, n_spec=2)
        rt = api.RayTracer(0)
        rt.load_scene(sc)
        frames(rt, "c1", len(sc.tris))
        del rt
    if "c2" in which:
        sc = scenes.bunny_class_scene()
        rt = api.RayTracer(0)
        rt.load_scene(sc)
        rt.primary_hits(0)
        best_all, best_ids, kern = 1e9, 1e9, 0.0
        for _ in range(4):
            t0 = time.perf_counter()
            rt.primary_hits(0, want_tab=True, pinned=True)
            best_all = min(best_all, time.perf_counter() - t0)
            kern = rt.stats()["trace_ms"]
            t0 = time.perf_counter()
            rt.primary_hits(0, want_tab=False, pinned=True)
            best_ids = min(best_ids, time.perf_counter() - t0)
        n = sc.width * sc.height * 4
        print("c2  %d triangles  %dx%d  primary hits: traversal kernels %.3f ms (%.0f Mrays/s); the call %.2f ms with the ids, %.2f ms with t/alpha/beta as well" % (
            len(sc.tris), sc.width, sc.height, kern, n / kern / 1e3, best_ids * 1e3, best_all * 1e3), flush=True)
        del rt
    for key, (w, h) in (("c3", (1920, 1080)), ("c4", (3840, 2160))):
        if key in which:
            sc = scenes.atrium_scene(width=w, height=h, max_depth=4)
            rt = api.RayTracer(0)
            rt.load_scene(sc)
            frames(rt, key, len(sc.tris))
            del rt
    if "c5" in which:
        sc = scenes.heightfield_scene(n_tris=10100000, width=3840, height=2160)
        sc.max_depth, sc.n_diff, sc.n_spec = 8, 1, 1
        rt = api.RayTracer(0)
        rt.load_scene(sc)
        frames(rt, "c5", len(sc.tris))
        del rt


if __name__ == "__main__":
    main()
