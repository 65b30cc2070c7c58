"""Builds libjxl_amd/csrc -> libjxl_hip.so (+ libjxl_threads_hip.so) with hipcc
for gfx950, in-tree.  Cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
BUILD = os.path.join(CSRC, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
LIB_SOURCES = ["context.hip", "kernels_blocks.hip", "kernels_filters.hip", "kernels_filters_fast.hip", "kernels_filters_fast_b.hip", "kernels_filters_fast_c.hip",
               "kernels_filters_fast_d.hip",
               "kernels_fused.hip", "kernels_fused_b.hip", "kernels_fused_pc.hip", "kernels_mfma.hip", "kernels_epf0.hip", "kernels_tables.hip", "entropy.cc"]
RUNNER_SOURCES = ["runner.cc"]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "jxl_hip.h"))
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "jxl_hip_entropy.h"))
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "jxl_hip_frame.h"))
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "jxl_hip_codestream.h"))
    return hdrs


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(src):
    obj = os.path.join(BUILD, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    extra = [os.path.join(CSRC, "kernels_fused.hip")] if src in ("kernels_fused_b.hip", "kernels_fused_pc.hip") else []  # they #include it
    if src.startswith("kernels_filters_fast_"):
        extra = [os.path.join(CSRC, "kernels_filters_fast.hip")]
    if _stale(obj, [path] + extra + _deps()):
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [HIPCC] + FLAGS + lang + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def _link(target, objs, extra=()):
    if _stale(target, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed for %s:\n%s" % (target, r.stderr[-4000:]))
    return target


def build(verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = list(LIB_SOURCES)
    have_runner = all(os.path.exists(os.path.join(CSRC, s)) for s in RUNNER_SOURCES)
    if have_runner:
        srcs += RUNNER_SOURCES
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = dict(zip(srcs, ex.map(_compile, srcs)))
    lib = _link(os.path.join(CSRC, "libjxl_hip.so"), [objs[s] for s in LIB_SOURCES])
    out = [lib]
    if have_runner:
        out.append(_link(os.path.join(CSRC, "libjxl_threads_hip.so"),
                         [objs[s] for s in RUNNER_SOURCES], ["-lpthread"]))
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
