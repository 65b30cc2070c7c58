#!/usr/bin/env python3
"""The drop-in boundary of INTEGRATION.md compiled for real (the maintainer's patch + the recipe that builds libjxl with it).

Two shared libraries with the reference's PUBLIC decoder API (lib/include/jxl/decode.h: JxlDecoderCreate,
JxlDecoderProcessInput, JxlDecoderSetImageOutBuffer ...; lib/jxl/decode.cc compiled in place) over the same
reference translation units oracle/build_ref.py builds (objects reused from oracle/_ref/obj):

  oracle/_ref/libjxl_dec_ref.so   the reference decoder, unmodified
  oracle/_ref/libjxl_dec_hip.so   the same with FrameDecoder::ProcessSections handing the AC groups of eligible
                                  frames to the HIP back-end: a patched COPY of lib/jxl/dec_frame.cc (written to
                                  oracle/_build/seam/, git-ignored, never committed) + integration/hip_seam.cc,
                                  linked against libjxl_amd/csrc/libjxl_hip.so

The patch is six insertions (declarations, the seam's clock, the DC-group hook of round 5 -- JxlHipAfterDcGlobal /
JxlHipDcGroup --, AC global's bit position, JxlHipTryAcGroups), applied by anchor (PATCH below) -- the reference file is read where it
lies under /root/reference; nothing of it is stored in this repository.  tests/test_seam.py drives both libraries
through JxlDecoderProcessInput on genuine codestreams with the JxlParallelRunner of libjxl_threads_hip.so and
holds their pixels to 2e-5 of each other.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))      # integration/: the binding and its build recipes
ROOT = os.path.dirname(HERE)
ORACLE = os.path.join(ROOT, "oracle")                   # build_ref.py, the Highway shim, _ref/ (outputs), _build/ (scratch)
sys.path.insert(0, HERE)
sys.path.insert(0, ORACLE)
import build_ref as B  # noqa: E402

SEAM = os.path.join(ORACLE, "_build", "seam")
HIPLIB_DIR = os.path.join(ROOT, "libjxl_amd", "csrc")
EXTRA_TUS = ["jxl/decode.cc", "jxl/decode_to_jpeg.cc"]  # what build_ref.py leaves out
FLAGS = B.FLAGS + ["-DJPEGXL_ENABLE_BOXES=0", "-DJPEGXL_ENABLE_TRANSCODE_JPEG=0"]

# (anchor line of lib/jxl/dec_frame.cc, text inserted BEFORE its FIRST occurrence); every anchor must exist
DECL = ("namespace jxl {\n"
        "class FrameDecoder;\n"
        "Status JxlHipTryAcGroups(FrameDecoder* fd, const FrameDecoder::SectionInfo* sections, size_t num,\n"
        "                         const std::vector<std::vector<size_t>>& ac_group_sec,\n"
        "                         const std::vector<size_t>& desired_num_ac_passes, size_t ac_global_sec,\n"
        "                         size_t ac_global_bit, FrameDecoder::SectionStatus* section_status, bool* done);\n"
        "void JxlHipNoteSectionsBegin();  // a timestamp for the seam's own clock (JXLHIP_SEAM_VERBOSE)\n"
        "void JxlHipAfterDcGlobal(FrameDecoder* fd, const BitReader* br);\n"
        "Status JxlHipDcGroup(FrameDecoder* fd, size_t dc_group, BitReader* br, bool* handled);\n"
        "}  // namespace jxl\n")
PATCH = [
    # the declaration: after the file's own includes (FrameDecoder is complete there)
    ("namespace jxl {", DECL),
    ("  std::fill(section_status, section_status + num, SectionStatus::kSkipped);",
     "  JxlHipNoteSectionsBegin();  // jxlhip seam: clock only\n"),
    # the DC groups through the product's host front-end, written into the reference's own state
    ("      section_status[dc_global_sec] = SectionStatus::kDone;",
     "      JxlHipAfterDcGlobal(this, sections[dc_global_sec].br);  // jxlhip seam\n"),
    ("        JXL_RETURN_IF_ERROR(ProcessDCGroup(i, sections[dc_group_sec[i]].br));",
     "        bool jxlhip_dc = false;  // jxlhip seam: the product's DC-group decoder when it takes the section\n"
     "        JXL_RETURN_IF_ERROR(JxlHipDcGroup(this, i, sections[dc_group_sec[i]].br, &jxlhip_dc));\n"
     "        if (!jxlhip_dc)\n"),
    ("  if (finalized_dc_ && ac_global_sec != num && !decoded_ac_global_) {",
     "  // jxlhip seam: where AC global starts in its reader (a one-section frame shares the reader)\n"
     "  const size_t jxlhip_ac_global_bit = ac_global_sec != num ? sections[ac_global_sec].br->TotalBitsConsumed() : 0;\n"),
    ("    // Mark all the AC groups that we received as not complete yet.",
     "    {  // jxlhip seam (integration/hip_seam.cc): the whole frame's AC groups on the HIP back-end when eligible\n"
     "      bool jxlhip_done = false;\n"
     "      JXL_RETURN_IF_ERROR(JxlHipTryAcGroups(this, sections, num, ac_group_sec, desired_num_ac_passes,\n"
     "                                            ac_global_sec, jxlhip_ac_global_bit, section_status, &jxlhip_done));\n"
     "      if (jxlhip_done) {\n"
     "        MarkSections(sections, num, section_status);\n"
     "        return true;\n"
     "      }\n"
     "    }\n"),
]


def patched_dec_frame():
    src = open(os.path.join(B.REF, "lib", "jxl", "dec_frame.cc")).read()
    out, hits = [], [0] * len(PATCH)
    for line in src.split("\n"):
        for i, (anchor, text) in enumerate(PATCH):
            if line == anchor and hits[i] == 0:
                out.append(text.rstrip("\n"))
                hits[i] = 1
        out.append(line)
    if hits != [1] * len(PATCH):
        raise RuntimeError("dec_frame.cc changed: patch anchors found: %s" % hits)
    os.makedirs(SEAM, exist_ok=True)
    path = os.path.join(SEAM, "dec_frame_hip.cc")
    new = "\n".join(out)
    if not os.path.exists(path) or open(path).read() != new:
        open(path, "w").write(new)
    return path


def _cc(src, obj, extra=()):
    # the seam's own sources see the C ABI headers (a stale object with an older jxlhip_frame_params is a
    # silent struct-size mismatch)
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    deps = [src, os.path.abspath(__file__)]
    if not src.startswith(B.REF):
        deps += [os.path.join(inc, h) for h in os.listdir(inc) if h.endswith(".h")]
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
        return
    r = subprocess.run([B.CXX] + FLAGS + list(extra) + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("%s:\n%s" % (src, r.stderr[-6000:]))


def available():
    return B.available()


# Both decoder libraries are compiled like bench.py's cpu_baseline build of the same sources: -O3 -mavx2 -mfma over
# the one-lane Highway shim (build_ref.py variant "fma": hardware FMA for MulAdd; bit-identical to the -O2 checker
# build, tests/test_reference_parity.py) -- djxl_ref should not be slower than it has to be when its MP/s stand next
# to djxl_hip's.
VARIANT = "fma"


def build(verbose=False):
    global FLAGS
    ref_so = os.path.join(B.OUT, "libjxl_dec_ref.so")
    hip_so = os.path.join(B.OUT, "libjxl_dec_hip.so")
    if not B.available():
        if os.path.exists(ref_so) and os.path.exists(hip_so):
            return ref_so, hip_so  # prebuilt, travelled with the snapshot
        raise RuntimeError("reference tree not present and no prebuilt seam libraries")
    objs = B.build(only_compile=True, variant=VARIANT)
    FLAGS = [f for f in B.FLAGS if f != "-O2"] + B.VARIANT_FLAGS[VARIANT] + ["-DJPEGXL_ENABLE_BOXES=0", "-DJPEGXL_ENABLE_TRANSCODE_JPEG=0"]
    os.makedirs(SEAM, exist_ok=True)
    extra_objs = []
    for f in EXTRA_TUS:
        o = os.path.join(SEAM, f.replace("/", "__")[:-3] + ".o")
        _cc(os.path.join(B.REF, "lib", f), o)
        extra_objs.append(o)
    inc = ["-I" + os.path.join(ROOT, "include")]
    patched = patched_dec_frame()
    o_patched = os.path.join(SEAM, "dec_frame_hip.o")
    _cc(patched, o_patched, inc)
    o_seam = os.path.join(SEAM, "hip_seam.o")
    _cc(os.path.join(HERE, "hip_seam.cc"), o_seam, inc)
    # the encoder's public API (JxlEncoder*) wants the JPEG-transcoding translation units: not part of a decoder library
    objs = [o for o in objs if not o.endswith("jxl__encode.o")]
    dec_frame_obj = [o for o in objs if o.endswith("jxl__dec_frame.o")]
    assert len(dec_frame_obj) == 1
    others = [o for o in objs if o != dec_frame_obj[0]]
    link = ["-Wl,--gc-sections", "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-lpthread", "-lm"]
    for so, parts, libs in ((ref_so, objs + extra_objs, []),
                            (hip_so, others + extra_objs + [o_patched, o_seam],
                             ["-L" + HIPLIB_DIR, "-l:libjxl_hip.so", "-Wl,-rpath," + HIPLIB_DIR,
                              "-Wl,-rpath,$ORIGIN/../../libjxl_amd/csrc"])):
        r = subprocess.run([B.CXX, "-shared", "-fPIC", "-o", so] + parts + link + libs, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link %s failed:\n%s" % (so, r.stderr[-6000:]))
    if verbose:
        print("built", ref_so, hip_so)
    return ref_so, hip_so


def build_ref_v8(verbose=False):
    """oracle/_ref/libjxl_dec_ref_v8.so: the unpatched reference decoder with the decode hot path (dec_group.cc and the
    transforms it includes, the Gaborish / EPF / XYB / write stages) compiled against the 8-lane Highway stand-in
    (oracle/build_ref.py variant "v8": libjxl's SIMD code paths, what bench.py's cpu_baseline times) -- for
    oracle/_ref/djxl_ref_v8, the CPU partner of djxl_hip in bench.py's e2e block.  The checker of the test suite stays the
    one-lane djxl_ref (bit-identical to the -O2 oracle build)."""
    global FLAGS
    so = os.path.join(B.OUT, "libjxl_dec_ref_v8.so")
    if not B.available():
        if os.path.exists(so):
            return so
        raise RuntimeError("reference tree not present and no prebuilt libjxl_dec_ref_v8.so")
    objs = B.build(only_compile=True, variant="v8")
    FLAGS = [f for f in B.FLAGS if f != "-O2"] + B.VARIANT_FLAGS["v8"] + ["-DJPEGXL_ENABLE_BOXES=0", "-DJPEGXL_ENABLE_TRANSCODE_JPEG=0"]
    os.makedirs(SEAM, exist_ok=True)
    extra_objs = []
    for f in EXTRA_TUS:
        o = os.path.join(SEAM, f.replace("/", "__")[:-3] + ".o")
        _cc(os.path.join(B.REF, "lib", f), o)
        extra_objs.append(o)
    objs = [o for o in objs if not o.endswith("jxl__encode.o")]
    link = ["-Wl,--gc-sections", "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-lpthread", "-lm"]
    r = subprocess.run([B.CXX, "-shared", "-fPIC", "-o", so] + objs + extra_objs + link, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("link %s failed:\n%s" % (so, r.stderr[-6000:]))
    if verbose:
        print("built", so)
    return so


if __name__ == "__main__":
    build(verbose=True)
    build_ref_v8(verbose=True)
