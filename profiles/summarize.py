"""Condenses rocprofv3 CSV output (kernel-trace stats and --pmc passes) into the text summaries kept in
this directory.  Usage: python profiles/summarize.py <dir with kt/ fetch/ write/ sq/ cache/> <out.txt>"""
import collections
import csv
import glob
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
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
