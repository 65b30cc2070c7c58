"""djxl, unmodified, on the HIP back-end (integration/build_djxl.py).

  oracle/_ref/djxl_ref : tools/djxl_main.cc + lib/extras on the reference decoder + lib/threads
  oracle/_ref/djxl_hip : the SAME objects on libjxl_dec_hip.so (reference JxlDecoder + the seam ->
                         libjxl_hip.so) + libjxl_threads_hip.so (the product's JxlParallelRunner)

djxl always decodes through an image-out CALLBACK (lib/extras/dec/jxl.h:64, jxl.cc:543-556) and asks for the
sample type of the file it writes: uint8 / big-endian uint16 for PPM (lib/extras/enc/pnm.cc:118-132), float for
PFM / NPY.  The GPU suite runs both tools on the same .jxl files, compares the files they write (integer samples
within 1 LSB and < 0.1 % different, float within 2e-5 of the range) and requires the seam's own log line: the frame
went through the HIP back-end.  A conformance mini-corpus (tools/conformance_hip.py: libjxl's corpus layout and
checks, tools/conformance/conformance.py:112-238) generated with djxl_ref is then run through djxl_hip.
CPU suite: both tools build and agree bit for bit without a device (the seam declines: CPU path)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
TIGHT = 2e-5


@pytest.fixture(scope="module")
def tools(oracle):
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import build_djxl
    try:
        ref, hip = build_djxl.build()
    except RuntimeError as e:
        pytest.skip(str(e))
    return ref, hip


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    return oracle


def stream(ref, original=None, orientation=None, **kw):
    """A genuine VarDCT codestream from the reference encoder; `original` = what the stream says its original was."""
    old = {k: os.environ.get(k) for k in ("JXR_ORIGINAL", "JXR_ORIENTATION")}
    if isinstance(kw.get("icc"), str):  # "rgb" | "grey": a display profile made on the spot (tests/test_icc.py)
        from test_icc import make_profile
        kw = dict(kw, icc=make_profile(kw["icc"] == "grey", 1024))
    try:
        for k, v in (("JXR_ORIGINAL", original), ("JXR_ORIENTATION", orientation)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return ref.RealStream(**kw).codestream.tobytes()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run(tool, args, verbose=False):
    env = dict(os.environ)
    if verbose:
        env["JXLHIP_SEAM_VERBOSE"] = "1"
    r = subprocess.run([tool] + args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (tool, args, r.stderr[-2000:])
    return r.stderr


def read_pnm(path):
    """P6 / PF as djxl writes them (lib/extras/enc/pnm.cc): returns [H, W, 3]."""
    b = open(path, "rb").read()
    magic, rest = b.split(b"\n", 1)
    dims, rest = rest.split(b"\n", 1)
    mx, data = rest.split(b"\n", 1)
    w, h = (int(v) for v in dims.split())
    if magic in (b"PF", b"Pf"):
        a = np.frombuffer(data, "<f4" if float(mx) < 0 else ">f4").reshape(h, w, 3 if magic == b"PF" else 1)
        return a[::-1]  # PFM rows run bottom to top
    assert magic in (b"P6", b"P5")  # P5: a grey image
    dt = np.uint8 if int(mx) < 256 else np.dtype(">u2")
    return np.frombuffer(data, dt).reshape(h, w, 3 if magic == b"P6" else 1), int(mx)


def read_pam(path):
    """P7 as djxl writes it for RGBA (lib/extras/enc/pnm.cc): returns ([H, W, 4], maxval)."""
    b = open(path, "rb").read()
    head, data = b.split(b"ENDHDR\n", 1)
    f = dict(l.split(b" ", 1) for l in head.split(b"\n")[1:] if l)
    w, h, d, mx = int(f[b"WIDTH"]), int(f[b"HEIGHT"]), int(f[b"DEPTH"]), int(f[b"MAXVAL"])
    assert head.startswith(b"P7") and d in (2, 4)  # RGB_ALPHA / GRAYSCALE_ALPHA
    return np.frombuffer(data, np.uint8 if mx < 256 else np.dtype(">u2")).reshape(h, w, d), mx


def compare(a_path, b_path, ext):
    if ext == "pam":
        (a, ma), (b, mb) = read_pam(a_path), read_pam(b_path)
        assert ma == mb and a.shape == b.shape
        # alpha is coded losslessly and leaves through the same arithmetic: identical
        assert np.array_equal(a[..., -1], b[..., -1])
        sd = a[..., :-1].astype(np.int64) - b[..., :-1].astype(np.int64)
        d = np.abs(sd)
        assert int(d.max()) <= (1 if ma < 256 else 2), int(d.max())
        assert float((d != 0).mean()) < (1e-3 if ma < 256 else 0.5)
        assert abs(float(sd.mean())) < (1e-3 if ma < 256 else 0.02), float(sd.mean())  # no systematic bias (in LSB)
        return
    if ext == "npy":
        a, b = np.load(a_path), np.load(b_path)
    elif ext == "pfm":
        a, b = read_pnm(a_path), read_pnm(b_path)
    else:
        (a, ma), (b, mb) = read_pnm(a_path), read_pnm(b_path)
        assert ma == mb and a.shape == b.shape
        sd = a.astype(np.int64) - b.astype(np.int64)
        d = np.abs(sd)
        # the float pipeline in front differs by <= 2e-5: a sample may land on the other side of a rounding step
        # (16-bit: 2e-5 x 65535 = 1.3 LSB, so up to half of the samples may differ by one) -- but not to ONE side: the
        # mean SIGNED difference stays a small fraction of an LSB (a systematic half-LSB bias would pass the two
        # bounds above and fail here)
        assert int(d.max()) <= (1 if ma < 256 else 2), int(d.max())
        assert float((d != 0).mean()) < (1e-3 if ma < 256 else 0.5)
        assert abs(float(sd.mean())) < (1e-3 if ma < 256 else 0.02), float(sd.mean())
        return
    assert a.shape == b.shape and a.dtype == b.dtype
    scale = max(1.0, float(np.abs(a).max()))
    assert float(np.abs(a.astype(np.float64) - b).max()) / scale <= TIGHT


def test_both_tools_agree_without_a_device(tools, ref, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: see the GPU suite")
    djxl_ref, djxl_hip = tools
    jxl = tmp_path / "a.jxl"
    jxl.write_bytes(stream(ref, original="srgb8", seed=5, xsize=200, ysize=120, distance=1.0))
    rgba = tmp_path / "rgba.jxl"
    rgba.write_bytes(stream(ref, original="srgb8", seed=5, xsize=200, ysize=120, distance=1.0, alpha_bits=8))
    for src, ext in ((jxl, "ppm"), (jxl, "pfm"), (jxl, "npy"), (rgba, "pam"), (rgba, "npy")):
        run(djxl_ref, [str(src), str(tmp_path / f"r.{ext}")])
        err = run(djxl_hip, [str(src), str(tmp_path / f"h.{ext}")], verbose=True)
        assert "jxlhip seam: frame" not in err and "jxlhip seam declines the frame" in err  # libjxl's own path
        assert (tmp_path / f"r.{ext}").read_bytes() == (tmp_path / f"h.{ext}").read_bytes()
    a, mx = read_pam(str(tmp_path / "r.pam"))
    assert mx == 255 and a.shape == (120, 200, 4) and len(np.unique(a[..., 3])) > 16


def test_dc_groups_through_the_product_front_end_leave_the_reference_state(tools, ref, tmp_path, monkeypatch):
    """Round 5: a VarDCT frame's DC groups are decoded by the product's host front-end (jxlhip_dc_group_decode) and written
    into libjxl's own state (integration/hip_seam.cc: JxlHipDcGroup) -- FrameDecoder::ProcessDCGroup is not called for
    them.  Without a device everything behind runs on libjxl's CPU path FROM THAT STATE: the bytes must be djxl_ref's.
    JXLHIP_SEAM_LIBJXL_DC=1 leaves the DC groups to libjxl (the control)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: see the GPU suite")
    djxl_ref, djxl_hip = tools
    for name, data in (("rgb", stream(ref, original="srgb8", seed=21, xsize=2200, ysize=300, distance=1.0)),
                       ("rgba", stream(ref, original="srgb16", seed=22, xsize=520, ysize=300, distance=1.5, alpha_bits=8)),
                       ("prog", ref.feature_stream("progressive"))):
        jxl = tmp_path / f"{name}.jxl"
        jxl.write_bytes(data)
        ext = "pam" if name == "rgba" else "ppm"
        run(djxl_ref, [str(jxl), str(tmp_path / f"r.{ext}")])
        for libjxl_dc in (False, True):
            if libjxl_dc:
                monkeypatch.setenv("JXLHIP_SEAM_LIBJXL_DC", "1")
            else:
                monkeypatch.delenv("JXLHIP_SEAM_LIBJXL_DC", raising=False)
            err = run(djxl_hip, [str(jxl), str(tmp_path / f"h.{ext}")], verbose=True)
            took = [l for l in err.splitlines() if "DC groups decoded by the product's front-end" in l]
            if libjxl_dc:
                assert not took, err[-800:]
            else:
                n = 2 if name == "rgb" else 1
                assert took and took[0].startswith(f"jxlhip seam: {n} of {n} DC groups"), err[-800:]
            assert (tmp_path / f"r.{ext}").read_bytes() == (tmp_path / f"h.{ext}").read_bytes(), (name, libjxl_dc)


def test_conformance_runner_agrees_with_the_reference_script(tools, ref, tmp_path):
    """Build container only: the corpus tools/conformance_hip.py writes is accepted by libjxl's own conformance.py,
    the corpus libjxl's generator.py writes is accepted by tools/conformance_hip.py, and both report a damaged
    expectation."""
    import conformance_hip as ch
    script = "/root/reference/tools/conformance/conformance.py"
    gen = "/root/reference/tools/conformance/generator.py"
    if not os.path.exists(script):
        pytest.skip("reference tree not present")
    djxl_ref, _ = tools
    jxl = tmp_path / "case_a.jxl"
    jxl.write_bytes(stream(ref, original="srgb8", seed=9, xsize=200, ysize=120, distance=1.0))
    env = dict(os.environ, LCMS2_LIB_PATH="/opt/conda/lib/liblcms2.so.2")
    ours, theirs = str(tmp_path / "ours"), str(tmp_path / "theirs")
    ch.generate(djxl_ref, ours, [str(jxl)], 1e-4, 2e-5)
    subprocess.check_call([sys.executable, gen, "--decoder", djxl_ref, "--output", theirs, "--peak_error", "1e-4",
                           "--rmse", "2e-5", str(jxl)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name in ("corpus.txt", "case_a/test.json", "case_a/reference.icc", "case_a/reference_image.npy"):
        assert open(os.path.join(ours, name), "rb").read() == open(os.path.join(theirs, name), "rb").read(), name
    res, _ = ch.run_corpus(djxl_ref, theirs, log=lambda s: None)
    assert res == {"case_a": True}
    assert subprocess.run([sys.executable, script, "--decoder", djxl_ref, "--corpus", ours], env=env,
                          capture_output=True).returncode == 0
    # a damaged expectation (one pixel off by 1e-3) must fail in both
    p = os.path.join(ours, "case_a", "reference_image.npy")
    a = np.load(p)
    a[0, 7, 9, 1] += 1e-3
    np.save(p, a)
    res, _ = ch.run_corpus(djxl_ref, ours, log=lambda s: None)
    assert res == {"case_a": False}
    assert subprocess.run([sys.executable, script, "--decoder", djxl_ref, "--corpus", ours], env=env,
                          capture_output=True).returncode != 0


CASES = [
    # (stream parameters, what the stream says its original was, EXIF orientation, outputs)
    (dict(seed=5, xsize=520, ysize=300, distance=1.0, speed_tier=3), "srgb8", None, ("ppm", "pfm", "npy")),
    (dict(seed=6, xsize=776, ysize=520, distance=2.0, speed_tier=3, progressive=1), "srgb16", None, ("ppm", "npy")),
    (dict(seed=7, xsize=640, ysize=264, distance=0.5, speed_tier=5), None, None, ("pfm", "npy")),
    (dict(seed=8, xsize=2200, ysize=264, distance=1.5, speed_tier=4), "srgb8", 6, ("ppm", "pfm")),   # rotate 90
    (dict(seed=9, xsize=200, ysize=120, distance=1.0, speed_tier=3), "srgb8", 3, ("ppm", "npy")),     # one section
    # RGBA: the alpha channel comes out of the frame's Modular bytes through the product's host front-end
    # (PAM is djxl's interleaved RGBA output; for NPY / PPM it asks for the alpha channel in a float / integer buffer of
    # its own, lib/extras/dec/jxl.cc:574-607, which the seam fills from the plane the host front-end decoded)
    (dict(seed=10, xsize=520, ysize=300, distance=1.0, speed_tier=3, alpha_bits=8), "srgb8", None, ("pam", "npy", "ppm")),
    (dict(seed=11, xsize=776, ysize=520, distance=2.0, speed_tier=4, alpha_bits=16), "srgb16", 5, ("pam", "npy")),
    (dict(seed=12, xsize=200, ysize=120, distance=1.0, speed_tier=3, alpha_bits=8), "srgb8", 8, ("npy", "pam")),
    # grey originals: djxl asks for 1 (or, with alpha, 2) channels; the back-end writes RGB(A) with R = G = B and the
    # seam hands out the first sample of every pixel
    (dict(seed=13, xsize=520, ysize=300, distance=1.0, speed_tier=3), "gray8", None, ("pgm", "npy", "pfm")),
    (dict(seed=14, xsize=456, ysize=280, distance=1.5, speed_tier=4, alpha_bits=8), "gray8", 6, ("pam", "npy", "pgm")),
    # alpha + three more extra channels under ONE palette: .npy carries all seven channels (extra-channel buffers)
    (dict(seed=480, xsize=520, ysize=513, speed_tier=4, alpha_bits=8, alpha_levels=5, extra=3), "srgb16", None, ("npy", "ppm")),
    # ICC originals (8-bit samples): no CMS in either build, so both write linear sRGB (dec_xyb.cc:160-164)
    (dict(seed=15, xsize=520, ysize=300, distance=1.0, speed_tier=3, icc="rgb"), "srgb8", None, ("ppm", "npy", "pfm")),
    (dict(seed=16, xsize=456, ysize=280, distance=1.5, speed_tier=4, icc="grey", alpha_bits=8), "gray8", None, ("pam", "npy")),
    # round 5: the remaining orientations (2 flip, 4 flip vertical, 7 anti-transpose; 1 is every case without one), 10- and
    # 12-bit originals (16-bit PNM with maxval 1023 / 4095), sizes that are multiples of nothing
    (dict(seed=17, xsize=517, ysize=301, distance=1.0, speed_tier=3), "srgb8", 2, ("ppm", "npy")),
    (dict(seed=18, xsize=333, ysize=267, distance=1.5, speed_tier=4), "srgb10", 4, ("ppm", "npy")),
    (dict(seed=19, xsize=601, ysize=299, distance=0.7, speed_tier=3, alpha_bits=8), "srgb12", 7, ("pam", "npy")),
    (dict(seed=20, xsize=259, ysize=131, distance=1.0, speed_tier=3), "srgb12", None, ("ppm", "pfm")),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kw,original,orientation,outputs", CASES)
def test_djxl_on_the_hip_backend_writes_what_djxl_writes(tools, ref, tmp_path, kw, original, orientation, outputs):
    djxl_ref, djxl_hip = tools
    jxl = tmp_path / "a.jxl"
    jxl.write_bytes(stream(ref, original=original, orientation=orientation, **kw))
    for ext in outputs:
        for threads in (("--num_threads", "0"), ()) if ext == outputs[0] else ((),):
            run(djxl_ref, [str(jxl), str(tmp_path / f"r.{ext}")] + list(threads))
            err = run(djxl_hip, [str(jxl), str(tmp_path / f"h.{ext}")] + list(threads), verbose=True)
            assert "jxlhip seam: frame" in err and "callback" in err, err[-1500:]
            compare(str(tmp_path / f"r.{ext}"), str(tmp_path / f"h.{ext}"), ext)


@pytest.mark.gpu
def test_djxl_on_a_genuine_4k_stream(tools, tmp_path):
    """tests/data/real_4k_d1.npz: 3840x2160 d1.0 written by the reference encoder (42 % 64x64, 32 % 32x32 ...)."""
    djxl_ref, djxl_hip = tools
    d = np.load(os.path.join(ROOT, "tests", "data", "real_4k_d1.npz"))
    jxl = tmp_path / "real4k.jxl"
    jxl.write_bytes(d["codestream"].tobytes())
    # (the stream describes a 32-bit float original: djxl itself refuses integer PNM output for it)
    for ext, extra in (("pfm", []), ("npy", []), ("npy", ["--num_threads", "3"])):
        run(djxl_ref, [str(jxl), str(tmp_path / f"r.{ext}")] + extra)
        err = run(djxl_hip, [str(jxl), str(tmp_path / f"h.{ext}")] + extra, verbose=True)
        assert "jxlhip seam: frame 3840x2160" in err, err[-1500:]
        compare(str(tmp_path / f"r.{ext}"), str(tmp_path / f"h.{ext}"), ext)


def e2e_stream_path():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_e2e_stream
    p = make_e2e_stream.DEFAULT
    return p if os.path.exists(p) else None


@pytest.mark.gpu
def test_djxl_on_a_genuine_8k_stream(tools, tmp_path):
    """BASELINE's frame size through the drop-in path: tests/data/e2e_8k_d1.jxl (7680x4320, d1.0, effort 7, written by
    the reference encoder: oracle/make_e2e_stream.py; the stream bench.py's e2e block times) through the unmodified
    djxl on the HIP back-end against the same tool on libjxl's CPU decoder: float pixels (.npy, .pfm) within 2e-5."""
    djxl_ref, djxl_hip = tools
    jxl = e2e_stream_path()
    if jxl is None:
        pytest.skip("tests/data/e2e_8k_d1.jxl not made (python oracle/make_e2e_stream.py)")
    for ext, extra in (("npy", []), ("pfm", ["--num_threads", "8"])):
        run(djxl_ref, [jxl, str(tmp_path / f"r.{ext}")] + extra)
        err = run(djxl_hip, [jxl, str(tmp_path / f"h.{ext}")] + extra, verbose=True)
        assert "jxlhip seam: frame 7680x4320" in err, err[-1500:]
        compare(str(tmp_path / f"r.{ext}"), str(tmp_path / f"h.{ext}"), ext)
        os.remove(str(tmp_path / f"r.{ext}"))
        os.remove(str(tmp_path / f"h.{ext}"))


@pytest.mark.gpu
def test_conformance_mini_corpus_through_djxl_hip(tools, ref, tmp_path):
    """Expectations = the reference decoder's float pixels (djxl_ref -> reference_image.npy, reference.icc,
    test.json with rms_error 2e-5 / peak_error 1e-4: tighter than any threshold of the ISO/IEC 18181-3 corpus, which
    is not available offline); decoder under test = djxl_hip; every test must pass AND have run on the device."""
    import conformance_hip as ch
    djxl_ref, djxl_hip = tools
    inputs = []
    for i, (kw, original, orientation, _) in enumerate(CASES):
        p = tmp_path / f"case{i}_{kw['xsize']}x{kw['ysize']}.jxl"
        p.write_bytes(stream(ref, original=original, orientation=orientation, **kw))
        inputs.append(str(p))
    p = tmp_path / "real4k.jxl"
    p.write_bytes(np.load(os.path.join(ROOT, "tests", "data", "real_4k_d1.npz"))["codestream"].tobytes())
    inputs.append(str(p))
    if e2e_stream_path():  # BASELINE's 8K frame size as a corpus entry
        p8 = tmp_path / "real8k.jxl"
        os.symlink(e2e_stream_path(), p8)
        inputs.append(str(p8))
    corpus = str(tmp_path / "corpus")
    ch.generate(djxl_ref, corpus, inputs, peak_error=1e-4, rmse=2e-5)
    log = []
    res, errs = ch.run_corpus(djxl_hip, corpus, log=log.append, env=dict(os.environ, JXLHIP_SEAM_VERBOSE="1"))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "conformance_mini_corpus.log"), "w").write("\n".join(log) + "\n")
    assert len(res) == len(inputs) and all(res.values()), (res, log[-20:])
    for tid, err in errs.items():
        assert "jxlhip seam: frame" in err, (tid, err[-800:])



# ---- frames the seam declines (round 5) -----------------------------------------------------------------------------
# PreparePipeline adds noise / patches / splines stages when the frame header asks (lib/jxl/dec_cache.cc:124,193-200); a
# Modular frame, a frame that is not the file's only regular frame and passes that arrive one by one take other paths of
# the reference.  The seam hands such frames back to libjxl's CPU path (integration/hip_seam.cc: decline): the pixels must
# be the reference's, with no device (CPU suite) and WITH a device present (GPU suite), and the log must say why.
# feature -> (decline lines expected, frames that run on the device when there is one)
FEATURES = {
    "noise": (["noise / patches / splines / DC frame"], 0),
    "splines": (["noise / patches / splines / DC frame"], 0),
    # the encoder writes the patch sources as a reference-only frame in front of the frame that uses them
    "patches": (["not an XYB VarDCT frame", "noise / patches / splines / DC frame"], 0),
    "modular": (["not an XYB VarDCT frame"], 0),
    "animation": (["not a single regular frame"], 1),   # the last frame of the two is an ordinary VarDCT frame
    "progressive": ([], 1),                              # whole file at hand: every pass at once, on the device
    "plain": ([], 1),
}


def feature_file(ref, tmp_path, feature):
    p = tmp_path / f"{feature}.jxl"
    p.write_bytes(ref.feature_stream(feature, xsize=600, ysize=400, seed=5, distance=1.0))
    return str(p)


def check_feature(tools, ref, tmp_path, feature, device):
    djxl_ref, djxl_hip = tools
    declines, on_device = FEATURES[feature]
    jxl = feature_file(ref, tmp_path, feature)
    for ext in ("ppm", "npy"):
        run(djxl_ref, [jxl, str(tmp_path / f"r.{ext}")])
        err = run(djxl_hip, [jxl, str(tmp_path / f"h.{ext}")], verbose=True)
        said = [l.split("declines the frame: ", 1)[1] for l in err.splitlines() if "jxlhip seam declines the frame" in l]
        took = [l for l in err.splitlines() if l.startswith("jxlhip seam: frame")]
        if device:
            assert said == declines, (feature, err[-1500:])
            assert len(took) == on_device, (feature, err[-1500:])
        else:  # without a device the frames the seam would take are declined last of all: "no device"
            assert said == declines + ["no device"] * on_device and not took, (feature, err[-1500:])
        if device and on_device:
            compare(str(tmp_path / f"r.{ext}"), str(tmp_path / f"h.{ext}"), ext)
        else:  # libjxl's own path in both tools: the same bytes
            assert (tmp_path / f"r.{ext}").read_bytes() == (tmp_path / f"h.{ext}").read_bytes(), (feature, ext)


@pytest.mark.parametrize("feature", sorted(FEATURES))
def test_declined_features_without_a_device(tools, ref, tmp_path, feature):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: see the GPU suite")
    check_feature(tools, ref, tmp_path, feature, device=False)


@pytest.mark.gpu
@pytest.mark.parametrize("feature", sorted(FEATURES))
def test_declined_features_with_a_device_present(tools, ref, tmp_path, feature):
    """The same files through djxl_hip on a box WITH a device: the declined frames still come out as djxl_ref's bytes,
    the frames beside them (the animation's last frame) run on the HIP back-end in the same process."""
    check_feature(tools, ref, tmp_path, feature, device=True)
