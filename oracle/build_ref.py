#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libjxl_ref.so: the libjxl
REFERENCE decoder sources compiled IN PLACE from /root/reference (never copied)
with g++, against oracle/hwy_shim (a from-scratch single-lane stand-in for the
un-vendored Highway submodule) -- no cmake, no reference build system.  The
driver oracle/ref_driver.cc calls the reference's own internals
(DecodeGroupForRoundtrip + the real render pipeline, dec_group.cc:820-841,
enc_adaptive_quantization.cc:840-919) on in-memory coefficients.

Used only by tests/ and bench.py's cpu_baseline as a checker.  /root/reference
exists only in the build container: on the GPU box the prebuilt .so travels.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("JXL_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
OBJ = os.path.join(OUT, "obj")
SHIM = os.path.join(HERE, "hwy_shim")
CXX = os.environ.get("CXX", "g++")
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fno-lto", "-ffunction-sections", "-fdata-sections",
         "-ffp-contract=off", "-fvisibility=hidden", "-w",
         "-I" + SHIM, "-I" + REF, "-I" + os.path.join(REF, "lib", "include"),
         "-DJPEGXL_ENABLE_SKCMS=0", "-DJXL_DEBUG_ON_ERROR=0"]  # (defining JXL_CRASH_ON_ERROR at all turns it on)
# reference translation units that are not needed by (or not linkable into) the
# VarDCT back-end harness: public API front-end, JPEG reconstruction (the ICC codec is in: ref_real_stream.cc
# writes streams of ICC originals and runs ICCReader beside the product's jxlhip_icc_decode)
SKIP = re.compile(r"(decode\.cc|decode_to_jpeg\.cc|jpeg/|_test\.cc|_gbench\.cc|test_)")


def source_list():
    txt = open(os.path.join(REF, "lib", "jxl_lists.cmake")).read()

    def lst(name):
        m = re.search(r"set\(%s\n(.*?)\n\)" % name, txt, re.S)
        return m.group(1).split()
    files = lst("JPEGXL_INTERNAL_BASE_SOURCES") + lst("JPEGXL_INTERNAL_DEC_SOURCES")
    # the ENCODER translation units: the tests let the reference write authentic AC
    # streams for the product's host entropy decoder (f1, tests/test_entropy.py) and whole
    # VarDCT codestreams of natural-looking images (tests/test_real_streams.py)
    files += [f for f in lst("JPEGXL_INTERNAL_ENC_SOURCES") if f not in files]
    return [f for f in files if f.endswith(".cc") and not SKIP.search(f)]


def available():
    return os.path.isdir(os.path.join(REF, "lib", "jxl"))


def _compile(job):
    src, obj = job
    deps = [src] + ([os.path.join(SHIM, "hwy", h) for h in os.listdir(os.path.join(SHIM, "hwy"))])
    if src.startswith(HERE):  # the drivers also see the C ABI and the oracle's POD mirror
        inc = os.path.join(HERE, "..", "include")
        deps += [os.path.join(inc, h) for h in os.listdir(inc)] + [os.path.join(HERE, "jxl_oracle.h")]
    vec = os.sep + "obj_v8" + os.sep in obj and any(src.endswith(u) for u in V8_UNITS)
    if vec:
        deps.append(os.path.join(SHIM_V, "hwy", "highway.h"))
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
        return obj, ""
    extra = ["-I" + HERE] if src.startswith(HERE) else []
    flags = ["-I" + SHIM_V] + FLAGS if vec else FLAGS  # (the vector highway.h in front of the single-lane one)
    r = subprocess.run([CXX] + flags + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
    return (obj if r.returncode == 0 else None), r.stderr


# variant "fma": the same sources and the same single-lane Highway shim compiled -O3 -mavx2 -mfma -- MulAdd (= fmaf) is
# one instruction instead of a libm call and the compiler may vectorise the one-lane loops.  Only bench.py's
# cpu_baseline uses it (libjxl_ref_fma.so); the checker stays the portable -O2 build.  Same results bit for bit
# (IEEE arithmetic either way, -ffp-contract=off; tests/test_reference_parity.py holds the two to equality).
VARIANT_FLAGS = {"": [], "fma": ["-O3", "-mavx2", "-mfma"], "v8": ["-O3", "-mavx2", "-mfma"]}
# variant "v8" (libjxl_ref_v8.so; bench.py's cpu_baseline): the translation units of the decode hot path compiled against
# oracle/hwy_shim_v -- 256-bit vectors, 8 float lanes, HWY_TARGET = HWY_AVX2: libjxl's SIMD code paths (vector DCTs and
# transposes, 8-pixel filter steps) instead of its one-lane ones -- and every other unit taken from the "fma" build.
# Target-specific code lives in per-TU namespaces (HWY_NAMESPACE) behind plain-C++ entry points, so the two kinds of
# object link together.  NOT bit-identical to the checker (other summation orders inside the vector DCTs): held to it
# within the reference's own executor tolerance by tests/test_reference_parity.py.
SHIM_V = os.path.join(HERE, "hwy_shim_v")
V8_UNITS = ("jxl/dec_group.cc", "jxl/render_pipeline/stage_gaborish.cc", "jxl/render_pipeline/stage_epf.cc",
            "jxl/render_pipeline/stage_xyb.cc", "jxl/render_pipeline/stage_write.cc", "jxl/dec_xyb.cc",
            # MaxVectorSize(): what image rows (image.cc) and scratch buffers are padded for -- must say 32 bytes when
            # any unit stores whole 8-lane vectors at row ends
            "jxl/simd_util.cc")


def build(verbose=False, only_compile=False, variant=""):
    global FLAGS
    lib = os.path.join(OUT, "libjxl_ref%s.so" % ("_" + variant if variant else ""))
    if not available():
        if os.path.exists(lib):
            return lib  # prebuilt, travelled with the snapshot
        raise RuntimeError("reference tree not present and no prebuilt " + lib)
    obj_dir = OBJ + ("_" + variant if variant else "")
    os.makedirs(obj_dir, exist_ok=True)
    base_flags = FLAGS
    if variant:
        FLAGS = [f for f in base_flags if f != "-O2"] + VARIANT_FLAGS[variant]
    try:
        return _build(lib, obj_dir, verbose, only_compile)
    finally:
        FLAGS = base_flags


def _build(lib, OBJ, verbose, only_compile):
    jobs = []
    v8 = OBJ.endswith("obj_v8")
    shared = OBJ[:-len("obj_v8")] + "obj_fma" if v8 else OBJ  # v8: only the hot units have objects of their own
    if v8:
        os.makedirs(shared, exist_ok=True)
    for f in source_list():
        d = OBJ if (not v8 or f in V8_UNITS) else shared
        jobs.append((os.path.join(REF, "lib", f), os.path.join(d, f.replace("/", "__")[:-3] + ".o")))
    if not only_compile:
        for drv in ("ref_driver", "ref_real_stream"):
            jobs.append((os.path.join(HERE, drv + ".cc"), os.path.join(shared, drv + ".o")))
    objs, failed = [], []
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for (src, _), (obj, err) in zip(jobs, ex.map(_compile, jobs)):
            if obj is None:
                failed.append((src, err))
            else:
                objs.append(obj)
    if failed:
        msg = "\n".join("== %s\n%s" % (s, e[:3000]) for s, e in failed[:8])
        raise RuntimeError("%d reference TUs failed to compile: %s\n%s" % (
            len(failed), [os.path.basename(s) for s, _ in failed], msg))
    if only_compile:
        return objs
    cmd = [CXX, "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,--gc-sections", "-Wl,--no-undefined",
                                                         "-lpthread", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-6000:])
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    build(verbose=True, only_compile="--compile-only" in sys.argv,
          variant="fma" if "--fma" in sys.argv else ("v8" if "--v8" in sys.argv else ""))
    sys.exit(0)
