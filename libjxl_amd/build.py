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
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # the gfx950 code objects travel zstd-compressed inside the fat binary (the HIP runtime unpacks them when the
         # library is loaded: measured on the GPU box, profiles/r06_library_size.txt): 12.1 -> ~4 MB
         "--offload-compress"]
LIB_SOURCES = ["context.hip", "kernels_blocks.hip", "kernels_filters.hip", "kernels_filters_fast.hip", "kernels_filters_fast_b.hip", "kernels_filters_fast_c.hip",
               "kernels_filters_fast_d.hip",
               "kernels_fused.hip", "kernels_fused_epf0.hip", "kernels_mfma.hip", "kernels_epf0.hip", "kernels_tables.hip", "entropy.cc"]
RUNNER_SOURCES = ["runner.cc"]
# Per-file flags.  kernels_blocks.hip: the SLP vectoriser pairs the butterflies of the in-register IDCTs into packed
# fp32 operations (v_pk_fma / v_pk_add / v_pk_mul on aligned register PAIRS, stitched together with v_mov): the pairs
# pushed k_transform_r<short> over its 168 registers (148 spilled VGPRs whose traffic reached HBM: 8K frames of DCT32X32
# read 1.44x / wrote 1.21x their bytes) -- without it the kernel needs 165 and spills nothing, k_transform_r16 drops from
# 113 to 93 (five waves per SIMD), and a packed instruction holds the SIMD ~1.6x as long as a plain one anyway
# (tools/probes/valu_issue.hip).  Explicit vector types (filters_march.h) are not affected.
EXTRA_FLAGS = {"kernels_blocks.hip": ["-fno-slp-vectorize"]}
if os.environ.get("JXLHIP_BUILD_NO_SLP_ALL"):  # experiment builds
    EXTRA_FLAGS = {k: ["-fno-slp-vectorize"] for k in LIB_SOURCES if k.endswith(".hip")}


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
    extra = [os.path.join(CSRC, "kernels_fused.hip")] if src == "kernels_fused_epf0.hip" else []  # it #includes it
    if src.startswith("kernels_filters_fast_"):
        extra = [os.path.join(CSRC, "kernels_filters_fast.hip")]
    if _stale(obj, [path] + extra + _deps() + [os.path.abspath(__file__)]):
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + lang + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def _link(target, objs, extra=()):
    if _stale(target, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", target] + objs + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed for %s:\n%s" % (target, r.stderr[-4000:]))
    return target


def _uint(b, i):
    t = b[i]
    if t <= 0x7F:
        return t
    if t == 0xCC:
        return b[i + 1]
    if t == 0xCD:
        return int.from_bytes(b[i + 1:i + 3], "big")
    if t == 0xCE:
        return int.from_bytes(b[i + 1:i + 5], "big")
    raise ValueError(hex(t))


LLVM_BIN = os.environ.get("JXLHIP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _code_objects(so):
    """The gfx950 code objects inside a built library, as bytes: the .hip_fatbin section is a sequence of offload bundles,
    one per translation unit, zstd-compressed since round 6 (--offload-compress: magic CCOB, 64-bit total size at offset
    8) or plain (__CLANG_OFFLOAD_BUNDLE__); clang-offload-bundler unpacks either."""
    import struct
    import tempfile
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            raise RuntimeError("llvm-objcopy could not read .hip_fatbin of %s: %s" % (so, r.stderr[-400:]))
        b = open(fat, "rb").read()
        blobs, i = [], 0
        while i < len(b):
            if b[i:i + 4] == b"CCOB":
                size = struct.unpack_from("<Q", b, i + 8)[0]
                blobs.append(b[i:i + size])
                i += size
            elif b[i:i + 24] == b"__CLANG_OFFLOAD_BUNDLE__":
                j = b.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
                k = b.find(b"CCOB", i + 24)
                ends = [x for x in (j, k) if x > 0]
                end = min(ends) if ends else len(b)
                blobs.append(b[i:end])
                i = end
            else:
                i += 1  # padding between bundles
        for n, blob in enumerate(blobs):
            src, dst = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
            open(src, "wb").write(blob)
            r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + src, "--output=" + dst],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(dst):
                raise RuntimeError("clang-offload-bundler could not unpack bundle %d of %s: %s" % (n, so, r.stderr[-400:]))
            out.append(open(dst, "rb").read())
    return out


def kernel_resources(so):
    """{mangled kernel name: {scratch, vgprs, spills}} from the code-object metadata inside a built library (the AMDGPU
    msgpack notes: .private_segment_fixed_size, .symbol, .vgpr_count, .vgpr_spill_count; the keys of a kernel record are
    sorted)."""
    import re
    out = {}
    for b in _code_objects(so):
        for m in re.finditer(rb"\xbb\.private_segment_fixed_size", b):
            scratch = _uint(b, m.end())
            s = b.find(b"\xa7.symbol", m.end(), m.end() + 400)
            if s < 0:
                continue
            t = b[s + 8]
            if t == 0xD9:
                n, at = b[s + 9], s + 10
            elif t == 0xDA:
                n, at = int.from_bytes(b[s + 9:s + 11], "big"), s + 11
            else:
                n, at = t & 0x1F, s + 9
            name = b[at:at + n].decode()
            v = b.find(b"\xab.vgpr_count", at, at + 600)
            sp = b.find(b"\xb1.vgpr_spill_count", at, at + 700)
            out[name] = dict(scratch=scratch, vgprs=_uint(b, v + 12), spills=_uint(b, sp + 18))
    return out


def check_no_scratch(so, pattern="k_fused_pc"):
    """A build whose producer / consumer fused kernels touch scratch must not ship: k_fused_pc's producing wave
    prefetches through inline-asm loads whose only wait is the barrier's vmcnt(0) (kernels_fused.hip) -- the compiler
    believes those registers hold their values from the asm statement on, so a spill or a scratch copy of one of them
    in between stores a value that has not arrived (a 128-VGPR ablation build did exactly that and faulted).  A ROCm
    point release that changes the register allocation must fail HERE, not corrupt memory on the GPU."""
    res = kernel_resources(so)
    if not any(pattern in k for k in res):  # (a check that finds no kernel to look at has checked nothing)
        raise RuntimeError("no %s kernel found in the metadata of %s" % (pattern, os.path.basename(so)))
    bad = {k: v for k, v in res.items() if pattern in k and (v["scratch"] or v["spills"])}
    if bad:
        os.remove(so)
        raise RuntimeError("refusing to ship %s: %d %s kernels use scratch (inline-asm prefetch registers may be "
                           "spilled before their loads land): %s" % (os.path.basename(so), len(bad), pattern,
                                                                      sorted(bad.items())[:2]))


def build(verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = list(LIB_SOURCES)
    have_runner = all(os.path.exists(os.path.join(CSRC, s)) for s in RUNNER_SOURCES)
    if have_runner:
        srcs += RUNNER_SOURCES
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = dict(zip(srcs, ex.map(_compile, srcs)))
    lib = _link(os.path.join(CSRC, "libjxl_hip.so"), [objs[s] for s in LIB_SOURCES])
    check_no_scratch(lib)
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
