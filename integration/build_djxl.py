#!/usr/bin/env python3
"""Builds the reference's own command-line decoder, `djxl`, UNMODIFIED, twice:

  oracle/_ref/djxl_ref   tools/djxl_main.cc + lib/extras (PNM / PFM / PGX / NPY writers; the PNG / JPEG / EXR / GIF
                         codecs compile to their "not available" stubs) on oracle/_ref/libjxl_dec_ref.so (the
                         reference decoder, unpatched) and oracle/_ref/libjxl_threads_ref.so (lib/threads)
  oracle/_ref/djxl_hip   the same objects on oracle/_ref/libjxl_dec_hip.so (the reference decoder with the
                         seam of integration/build_seam.py -> libjxl_hip.so) and the product's runner
                         libjxl_amd/csrc/libjxl_threads_hip.so

Every translation unit is compiled in place from /root/reference (g++, the Highway shim of oracle/hwy_shim);
nothing of the reference is stored in the repository.  The only source added is integration/djxl_support.cc (one
encoder-API helper lib/extras needs from the full libjxl, and the default CMS when lcms2 is not installed).
tests/test_djxl.py runs both tools on the same .jxl files (GPU suite) and tools/conformance_hip.py runs a
conformance corpus through them.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))      # integration/: the binding and its build recipes
ROOT = os.path.dirname(HERE)
ORACLE = os.path.join(ROOT, "oracle")                   # build_ref.py, the Highway shim, _ref/ (outputs), _build/ (scratch)
sys.path.insert(0, HERE)
sys.path.insert(0, ORACLE)
import build_ref as B  # noqa: E402
import build_seam as S  # noqa: E402

OBJ = os.path.join(ORACLE, "_build", "djxl")
HIPLIB_DIR = os.path.join(ROOT, "libjxl_amd", "csrc")
TOOL_TUS = ["tools/djxl_main.cc", "tools/cmdline.cc", "tools/codec_config.cc", "tools/speed_stats.cc",
            "tools/tool_version.cc"]
EXTRAS_TUS = ["alpha_blend.cc", "common.cc", "exif.cc", "packed_image.cc", "time.cc", "mmap.cc",
              "dec/color_description.cc", "dec/color_hints.cc", "dec/decode.cc", "dec/jxl.cc",
              "dec/apng.cc", "dec/exr.cc", "dec/gif.cc", "dec/jpg.cc", "dec/pgx.cc", "dec/pnm.cc",
              "enc/encode.cc", "enc/apng.cc", "enc/exr.cc", "enc/jpg.cc", "enc/npy.cc", "enc/pgx.cc", "enc/pnm.cc"]
THREADS_TUS = ["resizable_parallel_runner.cc", "thread_parallel_runner.cc", "thread_parallel_runner_internal.cc"]
FLAGS = [f for f in B.FLAGS if f != "-fvisibility=hidden"] + [
    '-DJPEGXL_VERSION="0.13.0-graft"', "-DJPEGXL_ENABLE_APNG=0", "-DJPEGXL_ENABLE_EXR=0", "-DJPEGXL_ENABLE_JPEG=0",
    "-DJPEGXL_ENABLE_SJPEG=0", "-DJPEGXL_ENABLE_GIF=0"]
LCMS_INC = "/opt/conda/include"
LCMS_LIB = "/opt/conda/lib/liblcms2.so.2"


def have_lcms():
    return os.path.exists(os.path.join(LCMS_INC, "lcms2.h")) and os.path.exists(LCMS_LIB)


def _cc(job):
    src, obj, extra = job
    deps = [src, os.path.abspath(__file__)] + [os.path.join(B.SHIM, "hwy", h) for h in os.listdir(os.path.join(B.SHIM, "hwy"))]
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
        return obj, ""
    r = subprocess.run([B.CXX] + FLAGS + list(extra) + ["-c", src, "-o", obj], capture_output=True, text=True)
    return (obj if r.returncode == 0 else None), r.stderr


def available():
    return B.available()


def outputs():
    return [os.path.join(B.OUT, n) for n in ("djxl_ref", "djxl_hip", "libjxl_threads_ref.so", "djxl_ref_v8")]


def build(verbose=False):
    if not B.available():
        if all(os.path.exists(o) for o in outputs()):
            return outputs()[:2]  # prebuilt, travelled with the snapshot
        raise RuntimeError("reference tree not present and no prebuilt djxl")
    dec_ref, dec_hip = S.build()
    os.makedirs(OBJ, exist_ok=True)
    lcms = have_lcms()
    jobs = [(os.path.join(B.REF, f), os.path.join(OBJ, f.replace("/", "__")[:-3] + ".o"), ()) for f in TOOL_TUS]
    jobs += [(os.path.join(B.REF, "lib", "extras", f), os.path.join(OBJ, "extras__" + f.replace("/", "__")[:-3] + ".o"), ())
             for f in EXTRAS_TUS]
    jobs += [(os.path.join(B.REF, "lib", "threads", f), os.path.join(OBJ, "threads__" + f[:-3] + ".o"), ("-fvisibility=hidden",))
             for f in THREADS_TUS]
    jobs.append((os.path.join(HERE, "djxl_support.cc"), os.path.join(OBJ, "djxl_support.o"),
                 ("-DDJXL_SUPPORT_HAVE_LCMS=%d" % int(lcms),)))
    if lcms:  # the reference's own CMS (lib/jxl/cms/jxl_cms.cc over lcms2): djxl --color_space
        jobs.append((os.path.join(B.REF, "lib", "jxl", "cms", "jxl_cms.cc"), os.path.join(OBJ, "jxl_cms.o"), ("-I" + LCMS_INC,)))
    objs, failed = [], []
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for (src, _, _), (obj, err) in zip(jobs, ex.map(_cc, jobs)):
            (objs.append(obj) if obj else failed.append((src, err)))
    if failed:
        raise RuntimeError("djxl build: %s" % "\n".join("== %s\n%s" % (s, e[-3000:]) for s, e in failed[:4]))
    thr_objs = [o for o in objs if os.path.basename(o).startswith("threads__")]
    tool_objs = [o for o in objs if o not in thr_objs]
    thr_so = os.path.join(B.OUT, "libjxl_threads_ref.so")

    def link(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd[:6]), r.stderr[-6000:]))
    link([B.CXX, "-shared", "-fPIC", "-o", thr_so] + thr_objs + ["-lpthread"])
    libs = []
    if lcms:
        local = os.path.join(B.OUT, "liblcms2.so.2")
        if not os.path.exists(local):
            shutil.copy(LCMS_LIB, local)  # a binary of this image, beside the tools that load it ($ORIGIN)
        libs = [local]
    rpath = ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../libjxl_amd/csrc", "-Wl,-rpath," + HIPLIB_DIR]
    link([B.CXX, "-o", os.path.join(B.OUT, "djxl_ref")] + tool_objs + [dec_ref, thr_so] + libs + ["-lpthread", "-lm"] + rpath)
    link([B.CXX, "-o", os.path.join(B.OUT, "djxl_hip")] + tool_objs +
         [dec_hip, os.path.join(HIPLIB_DIR, "libjxl_threads_hip.so")] + libs + ["-lpthread", "-lm"] + rpath)
    # djxl_ref_v8: the same tool on the decoder whose hot path runs libjxl's 8-lane SIMD code (bench.py's e2e block: the
    # honest CPU partner of djxl_hip; the tests keep the one-lane djxl_ref as their checker)
    dec_ref_v8 = S.build_ref_v8()
    link([B.CXX, "-o", os.path.join(B.OUT, "djxl_ref_v8")] + tool_objs + [dec_ref_v8, thr_so] + libs + ["-lpthread", "-lm"] + rpath)
    if verbose:
        print("built", outputs())
    return outputs()[:2]


if __name__ == "__main__":
    build(verbose=True)
