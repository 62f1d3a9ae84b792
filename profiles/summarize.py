"""Condenses rocprofv3 CSV output (kernel-trace stats and --pmc passes) into the text summaries kept in
this directory, and records the HBM bytes per launch of the traversal kernel (FETCH_SIZE doubled, as gfx950
needs for wide loads, plus WRITE_SIZE; both in KiB) in raytracer_amd/measured.json, where bench.py picks it up.
Usage: python profiles/summarize.py <dir with kt/ fetch/ write/ sq/ cache/> <out.txt>"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main(src, dst):
    out = []
    for f in glob.glob(src + "/kt/*/*kernel_stats.csv"):
        out.append("== rocprofv3 --kernel-trace --stats  (per kernel: calls, total ms, average us, share)")
        for r in csv.DictReader(open(f)):
            out.append("%-44s calls %5s  total %10.3f ms  avg %10.1f us  %6.2f %%" % (
                short(r["Name"])[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    for sub, title in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ counters"), ("cache", "cache counters")):
        files = glob.glob(src + "/" + sub + "/*/*counter_collection.csv")
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
        out.append("")
        out.append("== rocprofv3 --pmc pass: %s  (sums over all dispatches of the kernel; n = dispatches)" % title)
        for k in sorted(agg):
            for c in sorted(agg[k]):
                out.append("%-44s %-30s %16.6g  n=%d" % (k[:44], c, agg[k][c], calls[(k, c)]))
    traffic = {}
    for sub, scale in (("fetch", 2.0), ("write", 1.0)):
        files = glob.glob(src + "/" + sub + "/*/*counter_collection.csv")
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if short(r["Kernel_Name"]).startswith("k_trace_wide") and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                t = traffic.setdefault(sub, [0.0, 0])
                t[0] += float(r["Counter_Value"]) * 1024.0 * scale
                t[1] += 1
    if "fetch" in traffic and "write" in traffic and traffic["fetch"][1] == traffic["write"][1]:
        per_launch = (traffic["fetch"][0] + traffic["write"][0]) / traffic["fetch"][1]
        out.append("")
        out.append("== k_trace_wide HBM traffic per launch: (2 x FETCH_SIZE + WRITE_SIZE) / %d launches = %.4g bytes" % (traffic["fetch"][1], per_launch))
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "measured.json")
        json.dump({"k_trace_wide_hbm_bytes_per_launch": round(per_launch), "launches": traffic["fetch"][1],
                   "source": os.path.basename(dst)}, open(path, "w"), indent=1)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
