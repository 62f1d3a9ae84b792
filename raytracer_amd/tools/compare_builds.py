"""Same-box comparison of two builds of the device library over the benchmark configurations: the traversal kernel's time
depends on the code the compiler makes of it in ways that show on one configuration and not on another (a change worth
0.3 % on the 1080p atrium has cost the 10 M-triangle mesh 7 %), and two boxes differ by more than most changes are worth,
so a change is judged by alternating runs of both builds on one box, on every configuration.

    python raytracer_amd/tools/compare_builds.py <other librtx_hip.so> [rounds] [configs ...]     # needs an MI355X

Runs tools/configs_time.py (default: c1 c3 c5; wall time of the calls, best of four per run) `rounds` times (default 2) for the
build in the tree and for the other one in turn, adds the 1/8 tile share of the 1080p atrium, and prints ms per frame side
by side.  The other build is selected with RTX_HIP_SO; build it from another checkout with the flags of raytracer_amd/build.py."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHARE = r"""
import sys, time
sys.path.insert(0, %r)
from raytracer_amd import api, scenes
sc = scenes.atrium_scene(width=1920, height=1080, max_depth=4)
rt = api.RayTracer(0)
rt.load_scene(sc)
rt.set_tiles(0, 8, -1)
rt.reset()
rt.render(1, 1); rt.render(2, 1)
best = 1e9
for f in range(3, 11):
    t0 = time.perf_counter(); rt.render(f, 1); best = min(best, time.perf_counter() - t0)
print("c3  1/8 tile share  1 frame(s) per call: %%.3f ms per frame" %% (best * 1e3))
""" % ROOT


def one_run(other, configs):
    env = dict(os.environ)
    env.pop("RTX_HIP_SO", None)
    if other:
        env["RTX_HIP_SO"] = other
    out = subprocess.run([sys.executable, os.path.join(ROOT, "raytracer_amd", "tools", "configs_time.py")] + configs, env=env,
                         capture_output=True, text=True, timeout=900).stdout
    if "c3" in configs:
        out += subprocess.run([sys.executable, "-c", SHARE], env=env, capture_output=True, text=True, timeout=300).stdout
    res = {}
    for line in out.splitlines():
        m = re.match(r"(c\d)\s+(.*?)(\d+) frame\(s\) per call: ([0-9.]+) ms per frame", line)
        if m:
            key = "%s %sx%s" % (m.group(1), "1/8 share " if "share" in m.group(2) else "", m.group(3))
            res[key] = float(m.group(4))
        m = re.match(r"(c2).*traversal kernels ([0-9.]+) ms", line)
        if m:
            res["c2 kernels"] = float(m.group(2))
    return res


def main():
    other = os.path.abspath(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    configs = sys.argv[3:] or ["c1", "c3", "c5"]
    rows = {}
    for r in range(rounds):
        for name, so in (("tree", None), ("other", other)):
            for key, ms in one_run(so, configs).items():
                rows.setdefault(key, {"tree": [], "other": []})[name].append(ms)
    print("%-22s %-28s %-28s %s" % ("ms per frame", "build in the tree", os.path.basename(other), "other / tree (best of each)"))
    for key in sorted(rows):
        t, o = rows[key]["tree"], rows[key]["other"]
        if t and o:
            print("%-22s %-28s %-28s %.3f" % (key, " ".join("%.3f" % x for x in t), " ".join("%.3f" % x for x in o), min(o) / min(t)), flush=True)


if __name__ == "__main__":
    main()
