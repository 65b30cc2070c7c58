"""Memory-safety fuzzing of the host entropy decoder: entropy.cc is compiled with
g++ -fsanitize=address,undefined into tests/fuzz/fuzz_entropy.cc, which damages
genuine libjxl streams (and their side info) in seeded ways and runs the whole host
path on them.  Any out-of-bounds access, undefined behaviour or leak fails the test;
decode errors are the expected outcome.  CPU only."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "fuzz", "fuzz_entropy.cc")
ENTROPY = os.path.join(ROOT, "libjxl_amd", "csrc", "entropy.cc")


def blob(b):
    b = bytes(b)
    return struct.pack("<Q", len(b)) + b


def write_case(path, rs):
    shift = list(rs.shift) + [0] * (11 - len(rs.shift))
    with open(path, "wb") as f:
        f.write(b"JXHF" + struct.pack("<5I11I", rs.xsize, rs.ysize, rs.num_groups, rs.num_passes, rs.used_acs, *shift))
        f.write(blob(rs.block_ctx_bytes.tobytes()) + blob(rs.ac_strategy.tobytes()) +
                blob(np.ascontiguousarray(rs.raw_quant, np.int32).tobytes()) + blob(rs.quant_dc.tobytes()) +
                blob(rs.ac_global()))
        for p in range(rs.num_passes):
            for g in range(rs.num_groups):
                f.write(blob(rs.ac_group(g, p)))


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    out = tmp_path_factory.mktemp("fuzz") / "fuzz_entropy"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-DJXLHIP_NO_DEVICE", SRC, ENTROPY, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("sanitizer build failed:\n" + r.stderr[-3000:])
    return str(out)


@pytest.fixture(scope="module")
def cases(oracle, tmp_path_factory):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    d = tmp_path_factory.mktemp("cases")
    paths = []
    for i, kw in enumerate([dict(xsize=264, ysize=200, seed=3, distance=1.0, speed_tier=3),
                            dict(xsize=300, ysize=264, seed=4, distance=3.0, speed_tier=5),
                            dict(xsize=264, ysize=136, seed=5, distance=1.5, speed_tier=3, progressive=1),
                            dict(xsize=200, ysize=264, seed=6, distance=1.5, speed_tier=3, progressive=2)]):
        rs = oracle.RealStream(**kw)
        p = str(d / ("case%d.bin" % i))
        write_case(p, rs)
        paths.append(p)
    return paths


def test_damaged_streams_under_asan_ubsan(fuzzer, cases):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    iters = int(os.environ.get("JXLHIP_FUZZ_ITERS", "3000"))
    r = subprocess.run([fuzzer, str(iters), "20260923"] + cases, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    ok, rejected = map(int, r.stdout.split())
    assert ok + rejected == iters
    assert rejected > iters // 4   # the damage is real ...
    assert ok > 0                  # ... and not everything is thrown away (side-info-only damage often decodes)


def test_damaged_codestreams_through_the_front_end_chain_under_asan_ubsan(oracle, tmp_path):
    """Headers, TOC, DC global, the Modular global tree and DC groups (csrc/modular.inc) on damaged
    genuine codestreams: tests/fuzz/fuzz_codestream.cc."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    out = str(tmp_path / "fuzz_codestream")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-DJXLHIP_NO_DEVICE", os.path.join(ROOT, "tests", "fuzz", "fuzz_codestream.cc"),
           ENTROPY, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    paths = []
    for i, kw in enumerate([dict(xsize=264, ysize=200, seed=3, distance=1.0, speed_tier=3),
                            dict(xsize=300, ysize=264, seed=4, distance=3.0, speed_tier=5),
                            dict(xsize=520, ysize=136, seed=5, distance=0.5, speed_tier=2, progressive=1),
                            dict(xsize=264, ysize=200, seed=6, distance=2.0, icc="profile"),  # an ICC original
                            dict(xsize=520, ysize=300, seed=7, distance=2.0, alpha_bits=8),
                            dict(xsize=2200, ysize=264, seed=8, distance=2.0, alpha_bits=8, progressive=1, speed_tier=4),  # squeezed alpha
                            dict(xsize=520, ysize=513, seed=480, speed_tier=4, alpha_bits=8, alpha_levels=5, extra=3,
                                 original="srgb16")]):  # one palette over four extra channels
        if kw.get("icc"):
            from test_icc import extra_tags, make_profile
            kw["icc"] = make_profile(False, 300, extra_tags(np.random.default_rng(1)))
        rs = oracle.RealStream(**kw)
        p = str(tmp_path / ("cs%d.jxl" % i))
        with open(p, "wb") as f:
            f.write(rs.codestream.tobytes())
        paths.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    iters = int(os.environ.get("JXLHIP_FUZZ_ITERS", "3000"))
    r = subprocess.run([out, str(iters), "20260924"] + paths, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    ok, rejected = map(int, r.stdout.split())
    assert ok + rejected == iters and rejected > iters // 4 and ok > 0
