"""jxlhip_decode_codestream / jxlhip_codestream_basic_info (include/jxl_hip_codestream.h): from the BYTES of a
genuine libjxl stream (written by the reference's own encoder, oracle.RealStream) to pixels.

  CPU suite : headers + container walk (jxlc, in- and out-of-order jxlp) against the stream's known geometry;
              damaged containers refused.
  GPU suite : the pixels of the HIP path -- host parsers, DC groups and AC groups on the JxlParallelRunner of
              libjxl_threads_hip.so, DequantDC + smoothing, dequant tables and the whole back-end on the device --
              against the pixels the REFERENCE decoder produced for the same bytes (the GPU twin of
              tests/test_front_end_chain.py, which renders through the C oracle)."""
import ctypes as C
import struct

import numpy as np
import pytest

from libjxl_amd import abi

TIGHT = 2e-5

CASES = [
    dict(xsize=520, ysize=300, distance=1.0, speed_tier=3),
    dict(xsize=776, ysize=520, distance=3.0, speed_tier=3, progressive=1),   # 3 AC passes
    dict(xsize=640, ysize=264, distance=0.5, speed_tier=5),
    dict(xsize=2200, ysize=264, distance=1.5, speed_tier=4),                 # two DC groups
    dict(xsize=200, ysize=120, distance=1.0, speed_tier=3),                  # one section: no TOC permutation, bit-chained
    dict(xsize=384, ysize=520, distance=2.0, speed_tier=3, epf=1),
]


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    return oracle


@pytest.fixture(scope="module")
def L():
    return abi.load_library()


def box(kind, payload):
    return struct.pack(">I", 8 + len(payload)) + kind + payload


SIG = b"\0\0\0\x0cJXL \r\n\x87\n"
FTYP = box(b"ftyp", b"jxl \0\0\0\0jxl ")


def containers(cs):
    """The same codestream in the three container shapes the reference reads (decode.cc:1922-1990)."""
    half = len(cs) // 2
    yield "jxlc", SIG + FTYP + box(b"jxlc", cs)
    yield "jxlp", SIG + FTYP + box(b"jxlp", struct.pack(">I", 0) + cs[:half]) + box(b"Exif", b"\0" * 12) + \
        box(b"jxlp", struct.pack(">I", 0x80000001) + cs[half:])
    yield "jxlp out of order", SIG + FTYP + box(b"jxlp", struct.pack(">I", 0x80000001) + cs[half:]) + \
        box(b"jxlp", struct.pack(">I", 0) + cs[:half])


def test_basic_info_and_container_walk(L, ref):
    rs = ref.RealStream(seed=5, xsize=520, ysize=300, distance=1.0)
    cs = rs.codestream.tobytes()
    info = abi.CodestreamInfo()
    assert L.jxlhip_codestream_basic_info(cs, len(cs), C.byref(info)) == 0
    assert (info.xsize, info.ysize, info.container) == (520, 300, 0)
    for name, blob in containers(cs):
        info = abi.CodestreamInfo()
        assert L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)) == 0, name
        assert (info.xsize, info.ysize, info.container) == (520, 300, 1), name
    bad = [
        b"\xff\x0b" + cs[2:],                                             # not the codestream signature
        SIG[:11] + b"\x0b" + FTYP + box(b"jxlc", cs),                     # not the container signature
        SIG + FTYP + box(b"jxlc", cs) + box(b"jxlc", cs),                 # "there can only be one jxlc box"
        SIG + FTYP + box(b"jxlp", struct.pack(">I", 1) + cs),             # first jxlp index missing
        SIG + FTYP + struct.pack(">I", 4000000) + b"jxlc" + cs[:64],      # box longer than the file
        SIG + FTYP,                                                       # no codestream at all
        cs[:5],                                                           # truncated headers
    ]
    for blob in bad:
        assert L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)) == -5, blob[:24]  # JXLHIP_ERR_BAD_STREAM


def test_decode_without_device_fails_loudly(L, ref):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    ctx = C.c_void_p()
    assert L.jxlhip_create(0, C.byref(ctx)) == -2  # JXLHIP_ERR_NO_DEVICE: no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [0, 6])
@pytest.mark.parametrize("kw", CASES)
def test_decode_codestream_matches_the_reference_decoder(L, ref, kw, workers):
    import torch
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(seed=23, **kw)
    cs = rs.codestream.tobytes()
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    pool = R.JxlThreadParallelRunnerCreate(None, workers) if workers else None
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p) if workers else None
    dec = VarDctDecoder(0)
    try:
        blobs = [("bare", cs)] + (list(containers(cs))[1:] if workers else [])
        for name, blob in blobs:
            info = abi.CodestreamInfo()
            assert L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)) == 0
            out = torch.empty((info.ysize, info.xsize, 3), dtype=torch.float32, device="cuda")
            rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, blob, len(blob), 1, None, out.data_ptr(),
                                            info.xsize * 12, 0, C.byref(info))
            assert rc == 0, (name, rc, L.jxlhip_last_error(dec.ctx))
            assert (info.num_passes, info.num_groups, info.num_dc_groups) == (rs.num_passes, rs.num_groups, rs.num_dc_groups)
            assert info.used_acs == rs.used_acs
            got = out.cpu().numpy()
            scale = max(1.0, float(np.abs(rs.rgb).max()))
            assert float(np.abs(got - rs.rgb).max()) / scale <= TIGHT, name
        # packed 8-bit sRGB output of the same stream: decodes, and agrees with the float output to within a code
        fmt = abi.OutputFormat(1, 1, 4, 8, 0, 0.0, (C.c_float * 3)(0.2126, 0.7152, 0.0722))
        out8 = torch.empty((info.ysize, info.xsize, 4), dtype=torch.uint8, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, cs, len(cs), 2, C.byref(fmt), out8.data_ptr(),
                                        info.xsize * 4, 0, None)
        assert rc == 0, L.jxlhip_last_error(dec.ctx)
        lin = np.clip(rs.rgb, 0, 1)
        srgb = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1 / 2.4) - 0.055) * 255.0
        d = np.abs(out8.cpu().numpy()[..., :3].astype(np.float32) - srgb)
        assert d.max() <= 1.6 and (out8.cpu().numpy()[..., 3] == 255).all()
    finally:
        dec.close()
        if pool:
            R.JxlThreadParallelRunnerDestroy(pool)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(xsize=2200, ysize=520, distance=1.5, speed_tier=4),                  # two DC groups, 27 AC groups
    dict(xsize=776, ysize=520, distance=3.0, speed_tier=3, progressive=1),   # 3 AC passes
    dict(xsize=2300, ysize=2100, distance=0.6, speed_tier=4),                 # four DC groups of four sizes
    dict(xsize=640, ysize=520, distance=0.05, speed_tier=4),                  # coefficients beyond 16 bits, if any are
])
def test_one_runner_call_equals_three_barriers(L, ref, kw, monkeypatch):
    """With a runner the DC groups, AC global and the AC groups are ONE pool of work units in one runner call, the AC
    groups under a DC group starting when its block info is in (codestream.inc: PipelineJob); JXLHIP_NO_PIPELINE=1 runs
    the three phases of FrameDecoder one after the other (dec_frame.cc:596-703).  Same pixels, bit for bit, same
    coefficient type, for several worker counts -- and equal to the reference decoder's."""
    import torch
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(seed=31, **kw)
    cs = rs.codestream.tobytes()
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
    dec = VarDctDecoder(0)
    results = []
    try:
        for workers, barriers in ((2, False), (13, False), (13, True), (40, False)):
            if barriers:
                monkeypatch.setenv("JXLHIP_NO_PIPELINE", "1")
                abi.load_library().jxlhip_debug_reload_env()
            else:
                monkeypatch.delenv("JXLHIP_NO_PIPELINE", raising=False)
                abi.load_library().jxlhip_debug_reload_env()
            pool = R.JxlThreadParallelRunnerCreate(None, workers)
            try:
                for rep in range(3):  # (a context decodes frame after frame: the offset tables and arenas alternate)
                    info = abi.CodestreamInfo()
                    out = torch.zeros((rs.ysize, rs.xsize, 3), dtype=torch.float32, device="cuda")
                    rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, cs, len(cs), 1, None, out.data_ptr(), rs.xsize * 12, 0,
                                                    C.byref(info))
                    assert rc == 0, (workers, barriers, rc, L.jxlhip_last_error(dec.ctx))
                    results.append((workers, barriers, info.coeff_type, out.cpu().numpy()))
            finally:
                R.JxlThreadParallelRunnerDestroy(pool)
    finally:
        dec.close()
        monkeypatch.delenv("JXLHIP_NO_PIPELINE", raising=False)
        abi.load_library().jxlhip_debug_reload_env()
    first = results[0]
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    assert float(np.abs(first[3] - rs.rgb).max()) / scale <= TIGHT
    for r in results[1:]:
        assert r[2] == first[2], (r[:3], first[:3])
        assert np.array_equal(r[3], first[3]), r[:3]


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("barriers,workers,group", [(False, 9, "11"), (True, 9, "11"),
                                                   # two worker threads, the reporting group under either DC group of
                                                   # the frame (9 x 3 AC groups, 2 DC groups: columns 0-7 and 8): whichever
                                                   # DC group ends last, the tickets waiting on it must be woken (the
                                                   # wake-up is sent under the job's mutex since round 5)
                                                   (False, 2, "8"), (False, 2, "0"), (False, 2, "26"), (False, 1, "17")])
def test_redo_with_int32_coefficients(L, ref, barriers, workers, group, monkeypatch):
    """A coefficient beyond 16 bits on the optimistic 16-bit attempt: every AC group is decoded again into int32 buffers
    (jxl_hip_entropy.h) -- from inside the single runner call (the AC groups stop, the DC groups go on) and from the
    three-barrier path.  No stream of the reference encoder at ordinary settings gets there: the test hook
    JXLHIP_TEST_RANGE_GROUP makes one group report it."""
    import torch
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(seed=31, xsize=2200, ysize=520, distance=1.5, speed_tier=4)
    cs = rs.codestream.tobytes()
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
    pool = R.JxlThreadParallelRunnerCreate(None, workers)
    dec = VarDctDecoder(0)
    if barriers:
        monkeypatch.setenv("JXLHIP_NO_PIPELINE", "1")
        abi.load_library().jxlhip_debug_reload_env()
    try:
        got = []
        for hook in (None, group, None):
            if hook:
                monkeypatch.setenv("JXLHIP_TEST_RANGE_GROUP", hook)
                abi.load_library().jxlhip_debug_reload_env()
            else:
                monkeypatch.delenv("JXLHIP_TEST_RANGE_GROUP", raising=False)
                abi.load_library().jxlhip_debug_reload_env()
            info = abi.CodestreamInfo()
            out = torch.zeros((rs.ysize, rs.xsize, 3), dtype=torch.float32, device="cuda")
            rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, cs, len(cs), 1, None, out.data_ptr(), rs.xsize * 12, 0, C.byref(info))
            assert rc == 0, (hook, rc, L.jxlhip_last_error(dec.ctx))
            got.append((info.coeff_type, out.cpu().numpy()))
    finally:
        dec.close()
        R.JxlThreadParallelRunnerDestroy(pool)
        monkeypatch.delenv("JXLHIP_TEST_RANGE_GROUP", raising=False)
        abi.load_library().jxlhip_debug_reload_env()
        monkeypatch.delenv("JXLHIP_NO_PIPELINE", raising=False)
        abi.load_library().jxlhip_debug_reload_env()
    assert [g[0] for g in got] == [0, 1, 0]  # JXLHIP_COEFF_I16, _I32, _I16 again on the same context
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    for _, px in got:
        assert float(np.abs(px - rs.rgb).max()) / scale <= TIGHT
    assert np.array_equal(got[0][1], got[2][1])


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("barriers", [False, True])
def test_damaged_streams_end_the_runner_call_and_leave_the_context_usable(L, ref, barriers, monkeypatch):
    """Damage anywhere behind the headers -- DC groups, AC global, AC groups -- while the sections are being decoded on
    the runner: the call returns (an error or, for damage the entropy coder cannot see, pixels), the tickets of the
    single runner call drain (a DC group that fails after it announced its block info, an AC-global section that fails
    while DC groups wait for it, AC groups that fail beside running DC groups), nothing hangs, and the same context
    decodes the intact stream afterwards, bit for bit as before."""
    import torch
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(seed=31, xsize=2200, ysize=520, distance=1.5, speed_tier=4)
    cs = np.frombuffer(rs.codestream.tobytes(), np.uint8)
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
    pool = R.JxlThreadParallelRunnerCreate(None, 12)
    dec = VarDctDecoder(0)
    if barriers:
        monkeypatch.setenv("JXLHIP_NO_PIPELINE", "1")
        abi.load_library().jxlhip_debug_reload_env()
    rng = np.random.default_rng(17)

    def run(blob):
        info = abi.CodestreamInfo()
        out = torch.zeros((rs.ysize, rs.xsize, 3), dtype=torch.float32, device="cuda")
        raw = blob.tobytes()
        rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, raw, len(raw), 1, None, out.data_ptr(), rs.xsize * 12, 0, C.byref(info))
        return rc, out

    try:
        rc, good = run(cs)
        assert rc == 0
        good = good.cpu().numpy()
        failed = passed = 0
        first = 200  # (behind the image / frame headers and the TOC of this stream: the sections)
        for trial in range(60):
            b = cs.copy()
            where = trial % 3
            lo, hi = [(first, len(b) // 6), (len(b) // 6, len(b) // 3), (len(b) // 3, len(b))][where]
            for _ in range(int(rng.integers(1, 6))):
                k = int(rng.integers(lo * 8, hi * 8))
                b[k // 8] ^= np.uint8(1 << (k % 8))
            if trial % 7 == 0:
                b = b[: int(rng.integers(first, len(b)))].copy()
            rc, _ = run(b)
            failed += rc != 0
            passed += rc == 0
            if trial % 10 == 9:  # the context is still good for the intact stream
                rc, again = run(cs)
                assert rc == 0 and np.array_equal(again.cpu().numpy(), good), trial
        assert failed > 20, (failed, passed)
        rc, again = run(cs)
        assert rc == 0 and np.array_equal(again.cpu().numpy(), good)
    finally:
        dec.close()
        R.JxlThreadParallelRunnerDestroy(pool)
        monkeypatch.delenv("JXLHIP_NO_PIPELINE", raising=False)
        abi.load_library().jxlhip_debug_reload_env()


@pytest.mark.gpu
def test_unsupported_streams_are_refused_not_misdecoded(L, ref):
    """A damaged section must surface as an error, never as pixels."""
    from libjxl_amd import VarDctDecoder
    import torch
    rs = ref.RealStream(seed=3, xsize=520, ysize=300, distance=1.0)
    cs = bytearray(rs.codestream.tobytes())
    o = int(rs.section_offset[2 + rs.num_dc_groups]) + 3   # inside the first AC group
    for k in range(12):
        cs[o + k] ^= 0x5A
    dec = VarDctDecoder(0)
    try:
        out = torch.empty((300, 520, 3), dtype=torch.float32, device="cuda")
        blob = bytes(cs)
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, blob, len(blob), 1, None, out.data_ptr(), 520 * 12, 0, None)
        assert rc != 0
    finally:
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("orientation", [2, 5, 6, 8])
def test_display_orientation_like_jxldecoder(L, ref, orientation, monkeypatch):
    """A stream whose metadata carries an EXIF orientation: JxlDecoder writes DISPLAY orientation unless asked to keep
    the coded one (decode.h JxlDecoderSetKeepOrientation); so does jxlhip_decode_codestream with
    JXLHIP_OUT_UNDO_ORIENTATION -- against the reference's public decoder (oracle/_ref/libjxl_dec_ref.so, the
    unpatched half of the seam build), and coded orientation without the flag against FrameDecoder's pixels."""
    import sys, os
    import torch
    from libjxl_amd import VarDctDecoder
    import test_seam
    sys.path.insert(0, os.path.join(test_seam.ROOT, "integration"))
    import build_seam
    try:
        ref_so, _ = build_seam.build()
    except RuntimeError as e:
        pytest.skip(str(e))
    monkeypatch.setenv("JXR_ORIENTATION", str(orientation))
    rs = ref.RealStream(seed=5, xsize=328, ysize=200, distance=1.0, speed_tier=3)
    monkeypatch.delenv("JXR_ORIENTATION")
    cs = rs.codestream.tobytes()
    want = test_seam.jxl_decode(test_seam.load(ref_so), cs)  # display orientation
    dec = VarDctDecoder(0)
    try:
        info = abi.CodestreamInfo()
        assert L.jxlhip_codestream_basic_info(cs, len(cs), C.byref(info)) == 0
        assert info.orientation == orientation and (info.xsize, info.ysize) == (328, 200)
        oh, ow = (328, 200) if orientation >= 5 else (200, 328)
        assert want.shape == (oh, ow, 3)
        out = torch.empty((oh, ow, 3), dtype=torch.float32, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 1 | 0x100, None, out.data_ptr(), ow * 12, 0, None)
        assert rc == 0, L.jxlhip_last_error(dec.ctx)
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(out.cpu().numpy() - want).max()) / scale <= TIGHT
        coded = torch.empty((200, 328, 3), dtype=torch.float32, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 1, None, coded.data_ptr(), 328 * 12, 0, None)
        assert rc == 0, L.jxlhip_last_error(dec.ctx)
        assert float(np.abs(coded.cpu().numpy() - rs.rgb.reshape(200, 328, 3)).max()) / scale <= TIGHT
    finally:
        dec.close()


ORIGINALS = [None, "srgb8", "p3", "rec2100pq", "customxy", "gray8"]


@pytest.mark.parametrize("original", ORIGINALS)
def test_inverse_opsin_matrix_for_the_original_colour_space(L, ref, original):
    """jxlhip_output_opsin_matrix = OutputEncodingInfo::SetFromMetadata / SetColorEncoding (dec_xyb.cc:144-249): for an
    original with other primaries or white point than sRGB / D65 the coded matrix is followed by sRGB -> XYZ(D50) ->
    original.  Held, bit for bit, to the matrix the REFERENCE decoder derived for the same stream (Display P3,
    Rec.2100 PQ at 1000 nits, custom primaries with a D50 white point and a gamma curve)."""
    rs = ref.RealStream(264, 200, seed=3, original=original)
    cs = np.ascontiguousarray(rs.codestream)
    ih, pos = abi.ImageHeader(), C.c_size_t(0)
    assert L.jxlhip_image_header_decode(cs.ctypes.data, len(cs), C.byref(pos), None, 0, C.byref(ih)) == 0
    m, lum = (C.c_float * 9)(), (C.c_float * 3)()
    assert L.jxlhip_output_opsin_matrix(C.byref(ih), m, lum) == 0
    scale = np.float32(255.0) / np.float32(ih.intensity_target)
    mine = np.array([np.float32(v) * scale for v in m], np.float32)
    assert np.array_equal(mine, np.array(rs.frame_params.inverse_opsin_matrix, np.float32))
    assert abs(sum(lum) - 1.0) < 1e-5
    info = abi.CodestreamInfo()
    blob = cs.tobytes()
    assert L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)) == 0
    want = {None: (8, 1, 1, 0.0), "srgb8": (13, 1, 1, 0.0), "p3": (13, 11, 1, 0.0), "rec2100pq": (16, 9, 1, 0.0),
            "customxy": (None, 2, 2, 1 / 2.2), "gray8": (13, None, 1, 0.0)}[original]
    assert info.white_point == want[2] and abs(info.gamma - want[3]) < 1e-6
    if want[1] is not None:
        assert info.primaries == want[1]
    if want[0] is not None:
        assert info.transfer_function == want[0]
    assert list(info.luminances) == list(lum)


@pytest.mark.gpu
@pytest.mark.parametrize("original", ["p3", "rec2100pq", "customxy", "gray8"])
def test_pixels_in_the_original_colour_space(L, ref, original):
    """Display P3, Rec.2100 PQ, custom-primaries and grey (R = G = B) originals: jxlhip_decode_codestream with the transfer function the
    info struct names -> the pixels the reference decoder wrote (its default: the original space)."""
    import torch
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(520, 300, seed=19, original=original, distance=1.0, speed_tier=3)
    cs = rs.codestream.tobytes()
    info = abi.CodestreamInfo()
    assert L.jxlhip_codestream_basic_info(cs, len(cs), C.byref(info)) == 0
    tf, par = {"p3": (1, 0.0), "rec2100pq": (2, info.intensity_target), "customxy": (4, info.gamma), "gray8": (1, 0.0)}[original]
    fmt = abi.OutputFormat(tf, 0, 3, 32, 0, par, info.luminances)
    dec = VarDctDecoder(0)
    try:
        out = torch.full((300, 520, 3), -7.0, dtype=torch.float32, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 2, C.byref(fmt), out.data_ptr(), 520 * 12, 0, None)
        assert rc == 0, L.jxlhip_last_error(dec.ctx)
        got = out.cpu().numpy()
        # the encoded samples: steep curves near zero amplify the float pipeline's 2e-5 (as in the packed-format tests)
        assert float(np.abs(got - rs.rgb).max()) <= (2e-3 if original == "rec2100pq" else 2e-4)
        assert float(np.abs(got - rs.rgb).mean()) <= 2e-5
    finally:
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", ["1", "0"])
def test_first_frame_of_a_context_behind_a_busy_stream(L, ref, sparse, monkeypatch):
    """The hand-over of a context's FIRST frame (fresh upload buffers, the sparse arena just allocated) while the stream
    it shares with the caller still has ~50 ms of the caller's work queued: the uploads travel on other streams and
    must not depend on anything that only runs once the shared stream drains.  (Round 3: the arena used to be cleared
    by a NULL-stream hipMemset at allocation, which then ran AFTER the first batch of coefficients had landed; the
    frame came out as its DC image.)"""
    import torch
    from libjxl_amd import VarDctDecoder
    monkeypatch.setenv("JXLHIP_SPARSE_UPLOAD", sparse)   # both forms of the coefficient hand-over
    rs = ref.RealStream(seed=31, xsize=776, ysize=520, distance=2.0, speed_tier=3, epf=1)
    cs = rs.codestream.tobytes()
    a = torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    for workers_none in (True, False):
        dec = VarDctDecoder(0)          # a fresh context per run: its first frame
        try:
            b = a
            for _ in range(40):           # queued, not waited for
                b = b @ a
                b = b / b.abs().max()
            out = torch.full((520, 776, 3), -7.0, dtype=torch.float32, device="cuda")
            rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 1, None, out.data_ptr(), 776 * 12, 0, None)
            assert rc == 0, L.jxlhip_last_error(dec.ctx)
            got = out.cpu().numpy()
            assert float(np.abs(got - rs.rgb).max()) / max(1.0, float(np.abs(rs.rgb).max())) <= TIGHT
        finally:
            dec.close()
