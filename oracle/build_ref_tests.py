#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  The libjxl reference's OWN unit tests of the VarDCT hot path, compiled IN PLACE from
/root/reference and run against the checkers' Highway stand-ins:

  lib/jxl/dct_test.cc            1-D / 2-D (I)DCT of every size against the double-precision matrix form (1e-7 * N),
                                 the vector transposes
  lib/jxl/ac_strategy_test.cc    all 27 strategies: TransformFromPixels -> TransformToPixels round trip, DC = mean,
                                 LowestFrequenciesFromDC <-> DC, the downsampling identities, AFV
  lib/jxl/quant_weights_test.cc  DequantMatrices: library defaults, custom encodings, the DC-preserving property
  lib/jxl/opsin_inverse_test.cc  OpsinToLinear inverts ToXYB; the YCbCr pair

(googletest and Highway are un-vendored submodules of the reference and not installed here: oracle/gtest_shim is a
stand-in for the part of googletest these files use, oracle/hwy_shim / oracle/hwy_shim_v the one- and eight-lane
Highway stand-ins.)  Two binaries per test file:

  oracle/_ref/reftest_<name>_1   everything single-lane: the bit-exact checker's own build (objects of build_ref.py)
  oracle/_ref/reftest_<name>_8   the test file, dec_transforms_testonly.cc and the decode hot-path units on 8 lanes
                                 (build_ref.py variant "v8": HWY_TARGET = HWY_AVX2 code paths of the reference)

so the reference's known-answer tests check the stand-ins the checker and the CPU baseline rest on -- every
lane-crossing operation of the 8-lane header is exercised by dct_test's transposes and ac_strategy_test's transforms.
tests/test_reference_own_tests.py runs them (build container: builds; GPU box: the prebuilt binaries travel)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_ref as B  # noqa: E402

GTEST = os.path.join(HERE, "gtest_shim")
TESTS = ["dct_test", "ac_strategy_test", "quant_weights_test", "opsin_inverse_test"]
SUPPORT = ["jxl/test_memory_manager.cc", "jxl/dec_transforms_testonly.cc"]
MAIN = os.path.join(HERE, "ref_tests_main.cc")


def available():
    return B.available()


def binaries():
    return {(t, lanes): os.path.join(B.OUT, "reftest_%s_%d" % (t, lanes)) for t in TESTS for lanes in (1, 8)}


def _cc(src, obj, flags):
    deps = [src, os.path.join(GTEST, "gtest", "gtest.h"), os.path.join(GTEST, "hwy", "tests", "hwy_gtest.h"),
            os.path.join(B.SHIM, "hwy", "highway.h"), os.path.join(B.SHIM_V, "hwy", "highway.h")]
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
        return
    r = subprocess.run([B.CXX] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("%s:\n%s" % (src, r.stderr[-3000:]))


def build():
    """Builds (when the reference tree is present) and returns {(test, lanes): path}."""
    out = binaries()
    if not available():
        missing = [p for p in out.values() if not os.path.exists(p)]
        if missing:
            raise RuntimeError("reference tree not present and no prebuilt " + missing[0])
        return out
    B.build()               # oracle/_ref/obj: the single-lane objects
    B.build(variant="v8")   # obj_fma + obj_v8
    threads = os.path.join(B.OUT, "libjxl_threads_ref.so")
    if not os.path.exists(threads):
        raise RuntimeError("oracle/_ref/libjxl_threads_ref.so missing: run integration/build_djxl.py first")
    base = [f for f in B.FLAGS if f not in ("-fvisibility=hidden",)] + ["-I" + GTEST]
    for lanes in (1, 8):
        od = os.path.join(B.OUT, "obj_tests_%d" % lanes)
        os.makedirs(od, exist_ok=True)
        if lanes == 1:
            flags = base
            lib_dirs = [os.path.join(B.OUT, "obj")]
        else:
            flags = ["-I" + B.SHIM_V] + [f for f in base if f != "-O2"] + B.VARIANT_FLAGS["v8"]
            lib_dirs = [os.path.join(B.OUT, "obj_v8"), os.path.join(B.OUT, "obj_fma")]
        objs = []
        for d in lib_dirs:  # the reference's own objects (first directory wins: the 8-lane hot-path units)
            for o in sorted(os.listdir(d)):
                if o.endswith(".o") and not o.startswith(("ref_driver", "ref_real_stream")) and \
                        o not in [os.path.basename(x) for x in objs]:
                    objs.append(os.path.join(d, o))
        support = []
        for s in SUPPORT:
            o = os.path.join(od, s.replace("/", "__")[:-3] + ".o")
            _cc(os.path.join(B.REF, "lib", s), o, flags)
            support.append(o)
        om = os.path.join(od, "ref_tests_main.o")
        _cc(MAIN, om, flags)
        for t in TESTS:
            ot = os.path.join(od, t + ".o")
            _cc(os.path.join(B.REF, "lib", "jxl", t + ".cc"), ot, flags)
            exe = out[(t, lanes)]
            ins = [ot, om] + support + objs
            if os.path.exists(exe) and all(os.path.getmtime(exe) > os.path.getmtime(i) for i in ins):
                continue
            r = subprocess.run([B.CXX, "-o", exe] + ins + [threads, "-Wl,--gc-sections", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"],
                               capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError("link of %s failed:\n%s" % (exe, r.stderr[-4000:]))
    return out


def run(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    return r.returncode, r.stdout, r.stderr


if __name__ == "__main__":
    bins = build()
    bad = 0
    for (t, lanes), exe in sorted(bins.items()):
        rc, out, err = run(exe)
        tail = out.strip().splitlines()[-1] if out.strip() else ""
        print("%-22s %d lane(s): rc %d  %s" % (t, lanes, rc, tail))
        if rc:
            bad += 1
            print(err[-2000:])
    sys.exit(1 if bad else 0)
