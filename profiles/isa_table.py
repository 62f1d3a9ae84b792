"""Static instruction budget of k_trace_wide<16, false, false> (the closest-hit traversal kernel of the benchmark): compiles
raytracer_amd/csrc/rtx_hip.hip to gfx950 assembly with the product's flags, cuts the kernel into its phases at anchors
in the instruction stream, and tabulates every phase by instruction class and -- for the walk step -- by purpose.

    python profiles/isa_table.py [out.txt]        (needs hipcc; no GPU)

Classes follow the issue rates measured by raytracer_amd/tools/valu_rate (wave-instructions per ns per SIMD at six
waves per SIMD): "fast" vector instructions (v_fma/v_fmac/v_mul/v_add/v_sub f32, v_and/v_or/v_xor, v_add/v_sub u32,
v_lshrrev/v_ashrrev, v_mov, v_cndmask in its 32-bit encoding) issue at 0.78-1.0, every other vector instruction
("slow": conversions, min/max, compares, v_cndmask with an SGPR-pair condition, three-operand integer forms, packed
forms, the division helpers) at 0.55-0.58.
Phases (anchors): refill = from the first global_load of a ray record to the node fetch; walk = from the four
global_load_dwordx4 of a node (offsets 48, 32, 16, 0 off one base) to the first ds_write_b8 (the helper table); leaf = from
there to the end of the loop.  Instructions outside these are "loop control"."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-S"]
KERNEL = "k_trace_wideILi16ELb0ELb0E"

FAST = re.compile(r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|and_b32|or_b32|xor_b32|not_b32|add_u32|sub_u32|subrev_u32|lshrrev_b32|ashrrev_i32|mov_b32|mov_b64|cndmask_b32_e32)")


def klass(op):
    if op.startswith("v_"):
        return "valu fast" if FAST.match(op) else "valu slow"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def purpose(op, state):
    """Purpose of an instruction of the walk step, from its opcode and its position in the step."""
    if op.startswith("ds_") or op in ("v_lshl_or_b32", "v_lshl_add_u32", "v_lshlrev_b32_e32"):
        return "stack (LDS traffic and its addresses)"
    if op.startswith("v_cvt_f32_ubyte") or op.startswith("v_fma_f32") or op.startswith("v_fmac_f32"):
        return "bounds (byte -> float, one FMA per plane)"
    if op in ("v_mul_f32_e32", "v_sub_f32_e32", "v_add_f32_e32") and not state["sorted"]:
        return "bounds (per-node constants: origin, steps, slack)"
    if op.startswith("v_cndmask_b32_e64") and not state["seen_cvt"]:
        return "bounds (near/far planes by ray sign)"
    if op.startswith("v_max_f32") or op.startswith("v_min_f32") or op.startswith("v_max3_f32") or op.startswith("v_min3_f32") or op.startswith("v_cmp_nlt_f32"):
        return "cull (slab intervals, t >= 0.001, best hit)"
    if op.startswith("v_addc_co_u32") or (op.startswith("v_cndmask_b32_e64") and state["seen_cvt"] and not state["keys"]):
        return "cull (count of entered children)"
    if op.startswith("v_min_u32") or op.startswith("v_max_u32") or op.startswith("v_cmp_gt_u32") or (op.startswith("v_cndmask") and state["keys"]):
        return "sort (keys, five comparators on key and code)"
    if op.startswith("global_load"):
        return "node fetch"
    if op.startswith("global_atomic") or op.startswith("global_store") or op.startswith("v_mbcnt") or op.startswith("v_readfirstlane"):
        return "set-aside list (rare path)"
    return "state machine (next item, leaf slots, loop)"


def analyze():
    """Returns (text, data): the table as text and, for profiles/summarize.py, the static counts of the walk step."""
    with tempfile.TemporaryDirectory() as d:
        s_path = os.path.join(d, "dev.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(ROOT, "raytracer_amd", "csrc", "rtx_hip.hip"), "-o", s_path])
        lines = open(s_path).read().splitlines()
    body, on, meta = [], False, []
    for ln in lines:
        if re.match(r"^_ZN.*" + KERNEL + ".*:", ln):
            on = True
        if on:
            body.append(ln)
            if "s_endpgm" in ln:
                on = False
        if KERNEL in ln and ("NumVgprs" in ln or "Occupancy" in ln):
            meta.append(ln)
    for i, ln in enumerate(lines):
        if re.match(r"^_ZN.*" + KERNEL + ".*:", ln):
            for l2 in lines[i:]:
                if re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize)", l2):
                    meta.append(l2)
                if l2.startswith("; Occupancy"):
                    break
            break
    ins = []
    for ln in body:
        m = re.match(r"^\s+([a-z_0-9]+)\b(.*)", ln)
        if m and not ln.strip().startswith(";") and not ln.strip().startswith("."):
            ins.append((m.group(1), m.group(2)))
    # anchors
    # the node fetch: four 16-byte loads off one base, offsets 48, 32, 16, 0 (a scheduling s_nop may sit among them)
    node = next(i for i, (op, a) in enumerate(ins) if op == "global_load_dwordx4" and "offset:48" in a)
    first_ray = next(i for i, (op, a) in enumerate(ins) if op == "global_load_dwordx4")
    bperm = next(i for i, (op, a) in enumerate(ins) if op == "ds_write_b8")   # the helper table of the leaf phase
    # the walk step starts a few instructions before the fetch (address arithmetic): back to the last branch
    walk0 = node
    while walk0 > 0 and not ins[walk0 - 1][0].startswith("s_cbranch") and not ins[walk0 - 1][0].startswith("s_or_b64"):
        walk0 -= 1
    leaf0 = bperm
    while leaf0 > 0 and not ins[leaf0 - 1][0].startswith("s_cbranch"):
        leaf0 -= 1
    # the leaf code (triangle phase) sits behind the walk code in the stream; find where the walk ends: first s_cbranch_vccnz after the pops
    phases = collections.OrderedDict()
    def phase_of(i):
        if i < first_ray:
            return "prologue + loop head (ballots, refill decision)"
        if i < walk0:
            return "refill (claim a chunk, load rays, per-ray constants incl. three divisions)"
        if i < leaf0:
            return "walk step (one node: fetch, 4 child boxes, sort, push, next item)"
        return "leaf phase + loop tail (helper matching, 12 cross-lane pulls, box + triangle test, merge, hit store)"
    table = collections.defaultdict(collections.Counter)
    purp = collections.Counter()
    purp_class = collections.defaultdict(collections.Counter)
    state = {"seen_cvt": False, "keys": False, "sorted": False}
    for i, (op, a) in enumerate(ins):
        ph = phase_of(i)
        table[ph][klass(op)] += 1
        if ph.startswith("walk"):
            if op.startswith("v_cvt_f32_ubyte"):
                state["seen_cvt"] = True
            if op.startswith("v_max_u32") or op.startswith("v_min_u32") or op.startswith("v_cmp_gt_u32"):
                state["keys"] = True
            if state["seen_cvt"] and op.startswith("v_cndmask_b32") and a.strip().split(",")[1].strip() == "-1":
                state["keys"] = True   # key = entered ? bits : all ones
            p = purpose(op, state)
            if klass(op).startswith("valu") or klass(op) in ("lds", "vmem"):
                purp[p] += 1
                purp_class[p][klass(op)] += 1
    out = []
    out.append("Static instruction budget of k_trace_wide<16, false, false> (closest hit, 16-entry stack), hipcc " + " ".join(FLAGS[:5]))
    out.extend("  " + m.strip() for m in dict.fromkeys(meta))
    out.append("")
    cols = ["valu fast", "valu slow", "salu", "branch", "wait/nop", "lds", "vmem", "smem", "other"]
    out.append("%-92s" % "phase" + "".join("%10s" % c for c in cols) + "%10s" % "all")
    tot = collections.Counter()
    for ph, c in table.items():
        out.append("%-92s" % ph + "".join("%10d" % c[k] for k in cols) + "%10d" % sum(c.values()))
        tot.update(c)
    out.append("%-92s" % "whole kernel" + "".join("%10d" % tot[k] for k in cols) + "%10d" % sum(tot.values()))
    out.append("")
    out.append("Walk step by purpose (vector, LDS and vector-memory instructions; every path of the step counted once):")
    out.append("%-58s%10s%10s%8s%8s%8s" % ("purpose", "valu fast", "valu slow", "lds", "vmem", "all"))
    for p, n in sorted(purp.items(), key=lambda kv: -kv[1]):
        c = purp_class[p]
        out.append("%-58s%10d%10d%8d%8d%8d" % (p, c["valu fast"], c["valu slow"], c["lds"], c["vmem"], n))
    w = table[[k for k in table if k.startswith("walk")][0]]
    out.append("")
    out.append("Issue time of one walk step at the measured rates (fast 0.85, slow 0.565 wave-instructions per ns per SIMD): "
               "%d / 0.85 + %d / 0.565 = %.0f ns of vector issue per wave and step" % (w["valu fast"], w["valu slow"], w["valu fast"] / 0.85 + w["valu slow"] / 0.565))
    text = "\n".join(out) + "\n"
    fp_fast = sum(1 for i, (op, a) in enumerate(ins) if phase_of(i).startswith("walk") and re.match(r"^v_(fma|fmac|mul|add|sub|subrev)_f32", op))
    cvt = sum(1 for i, (op, a) in enumerate(ins) if phase_of(i).startswith("walk") and op.startswith("v_cvt_"))
    data = {"walk_valu_fast": w["valu fast"], "walk_valu_slow": w["valu slow"], "walk_fp_arith": fp_fast, "walk_cvt": cvt,
            # among the walk step's vector instructions that are neither FP32 add/mul/fma nor conversions: the share that issues at the fast rate
            "walk_other_fast_share": round((w["valu fast"] - fp_fast) / max(1, w["valu fast"] + w["valu slow"] - fp_fast - cvt), 4)}
    return text, data


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    text, _ = analyze()
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
