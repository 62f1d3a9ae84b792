"""Condenses the rocprofv3 CSV output of raytracer_amd/tools/collect_profile.sh into the text summaries kept in this
directory and into raytracer_amd/measured.json, the offline figures bench.py quotes beside its live ones.

    python profiles/summarize.py <dir with kt/ sq/ sq2/ grbm/ fetch/ write/ cache/ calib_*/> <out.txt>

What is derived for the traversal kernel k_trace_wide (all template variants together):
  lane_util   = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU): active lanes per vector-ALU instruction / 64
  issue model = sum over instruction classes (sq3 pass: FP32 add / mul / fma, conversions, transcendentals counted by
                the hardware, the rest split into fast and slow opcodes by the static mix of the walk step's code)
                of instructions / issue rate of the class (tools/valu_rate at the kernel's seven waves per SIMD), over 1,024 SIMDs,
                against the launch time of the kernel trace: 1.0 = the launch lasts as long as issuing takes
  valu_busy_uncalibrated = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * SIMDs): kept for comparison with earlier
                summaries; it is 4 x instructions / SIMD-cycles from two different runs, not a busy share
  fabric bytes: FETCH_SIZE and WRITE_SIZE are KiB.  FETCH_SIZE is 64 B x read requests of the L2 to the fabric
                whatever a request carries; the calibration runs (tools/fetch_calib) give the factor per access
                pattern: ~1.0 for random 64-byte records, 2.0 for random 128-byte records and for streams.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

SIMDS = 256 * 4


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def read_pass(src, sub):
    files = glob.glob(src + "/" + sub + "/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    rows = []
    if files:
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
            rows.append((k, r["Counter_Name"], float(r["Counter_Value"])))
    return agg, calls, rows


def family(agg, calls, prefix, counter):
    """Sum of a counter over every kernel whose name starts with prefix, and the number of dispatches.  The
    instrumented variants (last template argument true; bench.py runs them once, outside the timed region, for its
    test counts) are left out."""
    tot, n = 0.0, 0
    for k in agg:
        if k.startswith(prefix) and not k.endswith("true>") and counter in agg[k]:
            tot += agg[k][counter]
            n += calls[(k, counter)]
    return tot, n


def git_state():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    try:
        h = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "raytracer_amd/csrc", "bench.py"], text=True).strip()
        return h + ("+uncommitted" if dirty else "")
    except Exception:
        return "unknown"


def main(src, dst):
    out = []
    measured = {"source": "profiles/" + os.path.basename(dst), "commit": git_state()}
    kt_ms = {}
    # the workload of the profiled command: the JSON line bench.py printed in the kernel-trace pass
    try:
        for line in open(src + "/kt.log"):
            if line.startswith("{") and '"frames_per_step"' in line:
                measured["frames_per_step"] = json.loads(line)["config"]["frames_per_step"]
    except (OSError, ValueError, KeyError):
        pass
    for f in glob.glob(src + "/kt/*/*kernel_stats.csv"):
        out.append("== rocprofv3 --kernel-trace --stats  (per kernel: calls, total ms, average us, share)")
        for r in csv.DictReader(open(f)):
            out.append("%-44s calls %5s  total %10.3f ms  avg %10.1f us  %6.2f %%" % (
                short(r["Name"])[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
            kt_ms[short(r["Name"])] = (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]))
    passes = {}
    for sub, title in (("sq", "SQ counters"), ("sq3", "SQ vector instructions by class"), ("sq2", "SQ instruction mix"), ("grbm", "GRBM (clock)"), ("fetch", "FETCH_SIZE"),
                       ("write", "WRITE_SIZE"), ("cache", "cache counters")):
        agg, calls, _ = read_pass(src, sub)
        passes[sub] = (agg, calls)
        if not agg:
            continue
        out.append("")
        out.append("== rocprofv3 --pmc pass: %s  (sums over all dispatches of the kernel; n = dispatches)" % title)
        for k in sorted(agg):
            for c in sorted(agg[k]):
                out.append("%-44s %-30s %16.6g  n=%d" % (k[:44], c, agg[k][c], calls[(k, c)]))

    # ---- calibration of FETCH_SIZE
    calib = {}
    for M in (25, 1024):
        _, _, rows = read_pass(src, "calib_fetch_%d" % M)
        _, _, req = read_pass(src, "calib_req_%d" % M)
        if not rows:
            continue
        threads, reads = 1 << 21, 8
        asked = {"k_random_records<4>": threads * reads * 64.0, "k_random_records<8>": threads * reads * 128.0, "k_stream": M * 1048576.0}
        last = {}
        for k, c, v in rows:      # the second round of every kernel is the warm one
            if c == "FETCH_SIZE" and k in asked:
                last[k] = v
        lastreq = {}
        for k, c, v in req:
            if c == "TCC_EA0_RDREQ_sum" and k in asked:
                lastreq[k] = v
        for k in sorted(last):
            calib[(M, k)] = (asked[k], last[k] * 1024.0, lastreq.get(k))
    if calib:
        out.append("")
        out.append("== FETCH_SIZE calibration (tools/fetch_calib; second round of each kernel): bytes the kernel asked for, FETCH_SIZE x 1024,")
        out.append("   factor = asked / reported, fabric read requests (TCC_EA0_RDREQ_sum) and bytes asked per request")
        for (M, k), (asked, rep, rq) in sorted(calib.items()):
            out.append("table %5d MiB  %-22s asked %12.4g B  FETCH_SIZE %12.4g B  factor %.3f  requests %s  asked/request %s" % (
                M, k, asked, rep, asked / rep, "%.4g" % rq if rq else "-", "%.1f B" % (asked / rq) if rq else "-"))
        for M in (25, 1024):
            p = os.path.join(src, "calib_%d.log" % M)
            if os.path.exists(p):
                out.append("   un-profiled run, %d MiB table:" % M)
                out.extend("     " + line.rstrip() for line in open(p) if line.startswith("k_"))
        f64 = calib.get((1024, "k_random_records<4>"))
        f128 = calib.get((1024, "k_random_records<8>"))
        fst = calib.get((1024, "k_stream"))
        measured["fetch_size_factor"] = {"random_64B_records": round(f64[0] / f64[1], 3) if f64 else None,
                                         "random_128B_records": round(f128[0] / f128[1], 3) if f128 else None,
                                         "stream": round(fst[0] / fst[1], 3) if fst else None}

    # ---- the traversal kernel
    sq, sqc = passes.get("sq", ({}, {}))
    sq2, sq2c = passes.get("sq2", ({}, {}))
    gr, grc = passes.get("grbm", ({}, {}))
    fe, fec = passes.get("fetch", ({}, {}))
    wr, wrc = passes.get("write", ({}, {}))
    K = "k_trace_wide"
    insts, n = family(sq, sqc, K, "SQ_INSTS_VALU")
    if n:
        thr, _ = family(sq, sqc, K, "SQ_THREAD_CYCLES_VALU")
        act, _ = family(sq, sqc, K, "SQ_ACTIVE_INST_VALU")
        wavecyc, _ = family(sq, sqc, K, "SQ_WAVE_CYCLES")
        gui, ng = family(gr, grc, K, "GRBM_GUI_ACTIVE")
        salu, _ = family(sq2, sq2c, K, "SQ_INSTS_SALU")
        vmem, _ = family(sq2, sq2c, K, "SQ_INSTS_VMEM_RD")
        lds, _ = family(sq2, sq2c, K, "SQ_INSTS_LDS")
        lane_util = thr / (64.0 * insts)
        rec = {"launches": n, "valu_insts_per_launch": round(insts / n), "lane_util": round(lane_util, 4),
               "active_lanes_per_valu_inst": round(64.0 * lane_util, 2)}
        out.append("")
        out.append("== %s (the variants of the timed region, %d dispatches): vector-ALU model" % (K, n))
        out.append("   SQ_INSTS_VALU %.4g  SQ_THREAD_CYCLES_VALU %.4g  ->  %.1f of 64 lanes active per vector instruction (lane_util %.3f)" % (
            insts, thr, 64.0 * lane_util, lane_util))
        if ng:
            cycles = gui / 8.0
            busy = 4.0 * act / (cycles * SIMDS)
            rec["valu_busy_uncalibrated"] = round(busy, 3)
            rec["waves_per_simd"] = round(4.0 * wavecyc / (cycles * SIMDS), 2)
            out.append("   4 x SQ_ACTIVE_INST_VALU %.4g  /  (GRBM_GUI_ACTIVE / 8 = %.4g cycles x %d SIMDs)  =  %.3f -- uncalibrated: the counter charges 4 cycles per instruction" % (
                4.0 * act, cycles, SIMDS, busy))
            out.append("      whatever it issues in and the two counters come from different runs, so this is 4 x instructions / SIMD-cycles, not a busy share; the issue-time model below is the")
            out.append("      calibrated statement.  %.2f waves per SIMD resident" % rec["waves_per_simd"])
        if salu:
            out.append("   per vector instruction: %.3f scalar, %.4f vector-memory reads, %.4f LDS" % (salu / insts, vmem / insts, lds / insts))
        # ---- issue-time model: the kernel's vector instructions by class (sq3 pass) over the issue rates measured by
        # tools/valu_rate at the kernel's occupancy (rates.log).  FP32 add / mul / fma and conversions are counted by the
        # hardware; the rest (integer, compares, min/max, selects, moves) issues at the fast or the slow rate depending
        # on the opcode, and is split by the share the walk step's code has of each (profiles/isa_table.py).
        sq3, sq3c = passes.get("sq3", ({}, {}))
        rates = {}
        rp = os.path.join(src, "rates.log")
        if os.path.exists(rp):
            for line in open(rp):
                # the traversal kernel's own occupancy: seven waves per SIMD (six in logs taken before it had seven)
                if line.startswith("rate ") and ("waves_per_simd 7" in line or ("waves_per_simd 6" in line and line[5:26].strip() not in rates)):
                    name = line[5:26].strip()
                    rates[name] = float(line.split("per_simd_instr_per_ns")[1])
        tot3, n3 = family(sq3, sq3c, K, "SQ_INSTS_VALU")
        if n3 and rates:
            cls = {c: family(sq3, sq3c, K, "SQ_INSTS_VALU_" + c)[0] for c in ("ADD_F32", "MUL_F32", "FMA_F32", "CVT", "TRANS_F32", "INT32", "FLOPS_FP32")}
            rest = tot3 - cls["ADD_F32"] - cls["MUL_F32"] - cls["FMA_F32"] - cls["CVT"] - cls["TRANS_F32"]
            try:
                sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                import isa_table
                _, static = isa_table.analyze()
                fast_share = static["walk_other_fast_share"]
            except Exception as e:     # no hipcc at hand: the figure of the committed table
                static, fast_share = None, 0.28
            r_fast = (rates.get("v_or_b32", 0.96) + rates.get("v_add_u32", 0.96) + rates.get("v_mov_b32", 1.0)) / 3.0
            r_slow = sum(rates.get(k, 0.565) for k in ("v_max3_f32", "v_min_f32", "v_min_u32", "v_cmp_lt_f32 sgpr", "v_cndmask_b32 sgpr", "v_lshl_or_b32")) / 6.0
            r_trans = 0.25 * r_fast * 1.16   # v_rcp_f32 and friends: a quarter of the full rate (not measured by the tool)
            simd_ns = (cls["FMA_F32"] / rates.get("v_fma_f32", 0.776) + cls["MUL_F32"] / rates.get("v_mul_f32", 0.85) + cls["ADD_F32"] / rates.get("v_add_f32", 0.92) +
                       cls["CVT"] / rates.get("v_cvt_f32_ubyte1", 0.577) + cls["TRANS_F32"] / r_trans + rest * (fast_share / r_fast + (1.0 - fast_share) / r_slow))
            model_ms = simd_ns / SIMDS / n3 * 1e-6
            lo_ms = (simd_ns - rest * (fast_share / r_fast + (1.0 - fast_share) / r_slow) + rest / r_fast) / SIMDS / n3 * 1e-6
            hi_ms = (simd_ns - rest * (fast_share / r_fast + (1.0 - fast_share) / r_slow) + rest / r_slow) / SIMDS / n3 * 1e-6
            rec["valu_class_share"] = {c: round(v / tot3, 4) for c, v in cls.items() if c != "FLOPS_FP32"}
            rec["valu_class_share"]["other"] = round((rest - cls["INT32"]) / tot3, 4)
            rec["hw_fp32_flops_per_launch"] = round(cls["FLOPS_FP32"] * 64.0 * lane_util / n3)   # the counter counts per wave instruction
            rec["issue_model_ms_per_launch"] = round(model_ms, 4)
            rec["issue_model_ms_bounds"] = [round(lo_ms, 4), round(hi_ms, 4)]
            rec["issue_model_other_fast_share"] = fast_share
            out.append("   vector instructions by class (sq3 pass, %d dispatches): " % n3 + ", ".join("%s %.1f %%" % (c, 100.0 * v / tot3) for c, v in cls.items() if c != "FLOPS_FP32") +
                       ", other %.1f %%" % (100.0 * (rest - cls["INT32"]) / tot3))
            out.append("   rates at the kernel's occupancy (tools/valu_rate, wave-instructions per ns per SIMD): fma %.3f mul %.3f add %.3f cvt %.3f, fast integer/move %.3f, slow (min/max, compare, select, 3-operand) %.3f" % (
                rates.get("v_fma_f32", 0), rates.get("v_mul_f32", 0), rates.get("v_add_f32", 0), rates.get("v_cvt_f32_ubyte1", 0), r_fast, r_slow))
            out.append("   issue-time model: sum over classes of instructions / rate, over %d SIMDs = %.3f ms per launch (%.0f %% of the rest at the fast rate, as in the walk step's code;" % (SIMDS, model_ms, 100 * fast_share))
            out.append("      all of it fast: %.3f ms, all of it slow: %.3f ms)" % (lo_ms, hi_ms))
        f, nf = family(fe, fec, K, "FETCH_SIZE")
        w, nw = family(wr, wrc, K, "WRITE_SIZE")
        if nf and nw:
            rec["fetch_size_bytes_per_launch"] = round(f * 1024.0 / nf)
            rec["write_size_bytes_per_launch"] = round(w * 1024.0 / nw)
            out.append("   fabric traffic per launch: FETCH_SIZE %.4g B (= 64 B x read requests; x1 if they are 64-byte records, x2 if whole lines), WRITE_SIZE %.4g B" % (
                f * 1024.0 / nf, w * 1024.0 / nw))
        if K + "<24, false, false>" in kt_ms or any(k.startswith(K) for k in kt_ms):
            tot = sum(v[0] for k, v in kt_ms.items() if k.startswith(K) and not k.endswith("true>"))
            cnt = sum(v[1] for k, v in kt_ms.items() if k.startswith(K) and not k.endswith("true>"))
            rec["kernel_trace_avg_launch_ms"] = round(tot / cnt, 4)
            out.append("   kernel trace: %d launches, %.3f ms in all, %.4f ms per launch" % (cnt, tot, tot / cnt))
            if "issue_model_ms_per_launch" in rec:
                rec["issue_model_frac"] = round(rec["issue_model_ms_per_launch"] / (tot / cnt), 4)
                out.append("   issue-time model / measured launch time = %.3f: the share of the launch the vector ALUs need just to issue what the kernel executes" % rec["issue_model_frac"])
        measured[K] = rec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "measured.json")
    json.dump(measured, open(path, "w"), indent=1)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[-40:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
