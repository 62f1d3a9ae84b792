"""Builds the two native libraries in-tree: librtx_hip.so (hipcc, gfx950) and librtx_host.so (g++).

hipcc cross-compiles without a GPU, so this runs anywhere the ROCm toolchain is installed.
Floating-point mode matters for parity with the reference arithmetic: contraction off,
correctly rounded divide/sqrt, no fast-math.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# RTX_HIP_SO names another build of the device library (experiments, -DRTX_BOUNDS_CHECK builds); unset: the product's
HIP_SO = os.environ.get("RTX_HIP_SO") or os.path.join(PKG, "librtx_hip.so")
HOST_SO = os.path.join(PKG, "librtx_host.so")
CLI_BIN = os.path.join(PKG, "RayTracer")

HIP_SOURCES = ["rtx_hip.hip"]
HOST_SOURCES = ["rtx_host.cpp", "rtx_config.cpp", "rtx_image.cpp", "rtx_image_formats.cpp", "rtx_jpeg.cpp"]
HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-fvisibility=hidden"]
HOST_FLAGS = ["-O2", "-std=c++17", "-mavx2", "-mfma", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("rtx_wide.h", "rtx_dev_common.h", "rtx_dev_trace.h", "rtx_dev_shade.h")] + [os.path.join(PKG, "..", "include", "rtx.h")]
    if not force and _newer(HIP_SO, deps):
        return HIP_SO
    cmd = [_hipcc()] + HIP_FLAGS + srcs + ["-o", HIP_SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HIP_SO


def build_host(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, "rtx_host_internal.h"), os.path.join(CSRC, "rtx_image.h"), os.path.join(PKG, "..", "include", "rtx.h")]
    if not force and _newer(HOST_SO, deps):
        return HOST_SO
    cmd = ["g++"] + HOST_FLAGS + ["-shared"] + srcs + ["-o", HOST_SO, "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_SO


def build_cli(force=False, verbose=False):
    src = os.path.join(CSRC, "main.cpp")
    if not os.path.exists(src):
        return None
    if not force and _newer(CLI_BIN, [src, HOST_SO]):
        return CLI_BIN
    cmd = ["g++", "-O2", "-std=c++17", src, "-o", CLI_BIN, "-L" + PKG, "-lrtx_host", "-Wl,-rpath,$ORIGIN", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI_BIN


def build_tools(force=False, verbose=False):
    """tools/fetch_calib: the microbenchmark that calibrates rocprofv3's FETCH_SIZE; tools/valu_rate: the issue rates of
    the vector instructions the traversal kernel uses (see tools/collect_profile.sh)."""
    outs = []
    for name in ("fetch_calib", "valu_rate"):
        src = os.path.join(PKG, "tools", name + ".hip")
        out = os.path.join(PKG, "tools", name)
        if not os.path.exists(src):
            continue
        if force or not _newer(out, [src]):
            cmd = [_hipcc(), "-O2", "-std=c++17", "--offload-arch=gfx950", src, "-o", out]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        outs.append(out)
    return outs


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)
    build_cli(force, verbose)
    build_tools(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
