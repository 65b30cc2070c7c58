"""Pins the CPU oracle (oracle/*.c, a restatement) against the libjxl REFERENCE
ITSELF: oracle/_ref/libjxl_ref.so is lib/jxl's decoder sources compiled in place
(oracle/build_ref.py, single-lane Highway shim) and driven through the
reference's own DecodeGroupForRoundtrip + ComputeSigma + render pipeline
(oracle/ref_driver.cc).  Same in-memory inputs as the product's C ABI.

The restatement follows the reference's operation order (explicit fmaf where
the reference uses MulAdd), so the bar here is BIT-EXACT equality, for both of
the reference's executors (LowMemoryRenderPipeline = what djxl runs, and
SimpleRenderPipeline)."""
import numpy as np
import pytest

import frames
from libjxl_amd import synth


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    oracle.ref_lib()
    return oracle


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_dequant_tables_bit_identical(ref):
    # DequantMatrices::EnsureComputed, lib/jxl/quant_weights.cc:1211-1271
    a, b = ref.default_dequant_tables(), ref.ref_default_dequant_tables()
    assert np.array_equal(bits(a), bits(b))


STAGE_LISTS = [(g, e) for g in (False, True) for e in (0, 1, 2, 3)]


@pytest.mark.parametrize("gab,epf", STAGE_LISTS)
def test_all_strategies_every_stage_list(ref, gab, epf):
    _, _, fr = frames.make_case(328, 264, mix=synth.MIX_ALL, gab=gab, epf_iters=epf, seed=7 + epf)
    o = fr.decode(threads=4)
    assert np.array_equal(bits(o), bits(fr.decode_ref(threads=4)))
    assert np.array_equal(bits(o), bits(fr.decode_ref(threads=1, simple_pipeline=True)))


@pytest.mark.parametrize("xs,ys", [(1, 1), (3, 8), (8, 8), (9, 17), (255, 257), (256, 256), (257, 255),
                                   (513, 64), (64, 520)])
def test_ragged_sizes(ref, xs, ys):
    _, _, fr = frames.make_case(xs, ys, mix=synth.MIX_ALL, gab=True, epf_iters=3, seed=xs * 31 + ys)
    assert np.array_equal(bits(fr.decode(threads=2)), bits(fr.decode_ref(threads=2)))


def test_each_strategy_alone(ref):
    for s in range(27):
        cx, cy = ref.covered_blocks(s)
        _, _, fr = frames.make_case(max(64, 16 * cx), max(64, 16 * cy), mix={s: 1}, gab=True, epf_iters=1,
                                    seed=100 + s)
        assert np.array_equal(bits(fr.decode()), bits(fr.decode_ref())), f"strategy {s}"


def test_int32_coefficients_and_hdr_intensity(ref):
    # BASELINE configs[4]: DCT32x32 forced, int32 coefficients, intensity_target > 255
    _, _, fr = frames.make_case(520, 264, mix=synth.MIX_DCT32, gab=False, epf_iters=0, coeff_type=1,
                                intensity_target=4000.0, quant_mul=2.0)
    assert np.array_equal(bits(fr.decode(threads=2)), bits(fr.decode_ref(threads=2)))


def test_custom_loop_filter_fields(ref):
    _, _, fr = frames.make_case(300, 200, mix=synth.MIX_D1, gab=True, epf_iters=3, custom_lf=True, seed=5)
    assert np.array_equal(bits(fr.decode(threads=2)), bits(fr.decode_ref(threads=2)))


def test_xyb_planar_output(ref):
    # phase 1 only (no filters): the IDCT planes as the render pipeline receives them
    _, _, fr = frames.make_case(264, 136, mix=synth.MIX_ALL, gab=False, epf_iters=0, output_kind=0)
    o, r = fr.decode(), fr.decode_ref()
    assert o.shape == r.shape == (3, 136, 264)
    assert np.array_equal(bits(o), bits(r))
    _, _, fr = frames.make_case(264, 136, mix=synth.MIX_D1, gab=True, epf_iters=2, output_kind=0)
    assert np.array_equal(bits(fr.decode()), bits(fr.decode_ref()))


def test_d1_mix_1024_plumbing_config(ref):
    # BASELINE configs[0]: 1024x1024 d1.0 full pipeline on the CPU
    _, _, fr = frames.make_case(1024, 1024, mix=synth.MIX_D1, gab=True, epf_iters=1)
    assert np.array_equal(bits(fr.decode(threads=8)), bits(fr.decode_ref(threads=8)))


@pytest.mark.parametrize("smooth", [0, 1])
def test_dequant_dc_and_smoothing(ref, smooth):
    # DequantDC + AdaptiveDCSmoothing, lib/jxl/compressed_dc.cc:128-250
    import ctypes as C
    rng = np.random.default_rng(3)
    ysb, xsb = 37, 53
    base = rng.integers(-40, 40, size=(3, ysb, xsb))
    base[:, 10:20, 10:30] = base[:, 10:11, 10:11]  # flat areas exercise the smoothing gate
    q = [np.ascontiguousarray(base[c], np.int32) for c in range(3)]
    mul = np.array([1.0 / 4096 * 3.1, 1.0 / 512 * 3.1, 1.0 / 256 * 3.1], np.float32)
    want = ref.ref_dequant_dc(q, mul, 0.0357, 1.0119, smooth)
    got = [np.zeros((ysb, xsb), np.float32) for _ in range(3)]
    L = ref.lib()
    p3 = lambda a: (C.c_void_p * 3)(*[x.ctypes.data for x in a])  # noqa: E731
    L.jxo_dequant_dc(xsb, ysb, p3(q), p3(got), mul.ctypes.data_as(C.c_void_p), 0.0357, 1.0119)
    if smooth:
        L.jxo_adaptive_dc_smoothing(xsb, ysb, mul.ctypes.data_as(C.c_void_p), p3(got))
    for c in range(3):
        assert np.array_equal(bits(got[c]), bits(want[c])), c


PACKED_FORMATS = [
    # (transfer, sample_type, bits, channels, swap)
    (1, 1, 8, 3, 0),   # sRGB u8 RGB  (djxl -> PPM/PNG default)
    (1, 1, 8, 4, 0),   # sRGB u8 RGBA
    (0, 1, 5, 3, 0),   # linear, 5-bit samples in bytes
    (1, 2, 16, 3, 0),  # sRGB u16
    (1, 2, 12, 4, 1),  # sRGB 12-bit in u16, big-endian, RGBA
    (1, 3, 0, 3, 0),   # sRGB f16
    (0, 3, 0, 4, 1),   # linear f16 RGBA, byte-swapped
    (1, 0, 0, 3, 0),   # sRGB f32
    (0, 0, 0, 4, 1),   # linear f32 RGBA, byte-swapped
]


@pytest.mark.parametrize("tf,par,it", [(2, 1000.0, 1000.0), (2, 255.0, 255.0), (3, 0.0, 255.0), (4, 1 / 2.6, 255.0),
                                       (4, 0.45455, 80.0),
                                       (5, 1000.0, 1000.0),   # HLG at 1000 nits: OOTF gamma 1/1.2
                                       (5, 255.0, 255.0),     # ... at the SDR default
                                       (5, 334.0, 334.0)])    # ... where the system gamma is ~1: OOTF skipped
@pytest.mark.parametrize("st,bits_", [(1, 8), (2, 16), (0, 0)])
def test_packed_output_pq_709_gamma(ref, tf, par, it, st, bits_):
    """FromLinearStage's OpPq (two rational polynomials in x^(1/4)), Op709 and OpGamma
    (FastPowf) followed by the write stage: restatement == reference, byte for byte."""
    _, _, fr = frames.make_case(264, 136, mix=synth.MIX_D1, gab=True, epf_iters=1, output_kind=2,
                                intensity_target=it, seed=23,
                                out_format=dict(transfer=tf, sample_type=st, num_channels=3, bits_per_sample=bits_,
                                                tf_param=par))
    o, r = fr.decode(threads=2), fr.decode_ref(threads=2)
    assert np.array_equal(o.view(np.uint8), r.view(np.uint8))


@pytest.mark.parametrize("tf,st,bits_,nc,sw", PACKED_FORMATS)
def test_packed_output_stages(ref, tf, st, bits_, nc, sw):
    """f3: FromLinearStage (TF_SRGB) + WriteToOutputStage (scale, ordered dither,
    clamp, round, interleave, alpha = 1, endianness): restatement == reference, byte for byte."""
    _, _, fr = frames.make_case(264, 136, mix=synth.MIX_D1, gab=True, epf_iters=1, output_kind=2,
                                intensity_target=80.0 if tf else 255.0, seed=17 + st,
                                out_format=dict(transfer=tf, sample_type=st, num_channels=nc,
                                                bits_per_sample=bits_, swap_endianness=sw))
    o, r = fr.decode(threads=2), fr.decode_ref(threads=2)
    assert o.shape == r.shape == (136, 264, nc) and o.dtype == r.dtype
    assert np.array_equal(o.view(np.uint8), r.view(np.uint8))
    if st in (1, 2) and tf:  # the frame spans a useful range, not a constant
        assert int(r.max()) - int(r.min()) > (1 << bits_) // 4


def test_srgb_transfer_function_known_values(ref):
    # TF_SRGB::EncodedFromDisplay: 12.92 x below the threshold, within 5e-7 of the
    # exact sRGB curve above it (transfer_functions-inl.h:244), odd symmetry
    import ctypes as C
    L = ref.lib()
    L.jxo_srgb_from_linear.restype = C.c_float
    L.jxo_srgb_from_linear.argtypes = [C.c_float]
    assert L.jxo_srgb_from_linear(0.0) == 0.0
    assert L.jxo_srgb_from_linear(0.001) == np.float32(np.float32(0.001) * np.float32(12.92))
    for x in (0.0031309, 0.01, 0.18, 0.5, 1.0):  # the approximation is fitted on [0, 1]
        exact = 1.055 * x ** (1 / 2.4) - 0.055
        assert abs(L.jxo_srgb_from_linear(x) - exact) < 2e-6 * max(1.0, exact)
        assert L.jxo_srgb_from_linear(-x) == -L.jxo_srgb_from_linear(x)


def _decode_ref_in_order(ref, fr, order, threads=1):
    """decode_ref with the workaround of oracle.ref_threads switched off and the groups handed out in `order`
    (JXR_GROUP_ORDER, oracle/ref_driver.cc)."""
    import os
    keep = ref.ref_threads
    ref.ref_threads = lambda x, y, t: t
    try:
        if order:
            os.environ["JXR_GROUP_ORDER"] = order
        return fr.decode_ref(threads=threads)
    finally:
        os.environ.pop("JXR_GROUP_ORDER", None)
        ref.ref_threads = keep


def test_reference_group_order_dependence_is_the_mirroring_test(ref):
    """Round 1's "NaN column now and then with threads"  is not a data race in the driver: ONE
    thread reproduces it whenever the narrow last group column (530 = 2 * 256 + 18 px: narrower than the 16-px
    border strip + the 3-px filter border) is finished before its left neighbour.  The reference then renders the
    strip [496, 528) alone, Gaborish runs 2 columns past it (xextra_right) and reads column 530 -- past the image
    edge, unmirrored, because ApplyXMirroring only looks at rect.x1 + border (low_memory_render_pipeline.cc:496,
    :510).  In index order the reference equals the restatement bit for bit."""
    _, _, fr = frames.make_case(530, 300, mix=synth.MIX_ALL, gab=True, epf_iters=1, seed=24)
    o = fr.decode(threads=4)
    assert np.array_equal(bits(o), bits(_decode_ref_in_order(ref, fr, None)))
    assert np.array_equal(bits(o), bits(fr.decode_ref(threads=1, simple_pipeline=True)))
    swapped = _decode_ref_in_order(ref, fr, "0,2,1")
    d = np.argwhere(swapped != o)
    assert len(d) > 0, "the reference no longer depends on the group order: oracle.ref_threads can go"
    assert set(d[:, 1].tolist()) <= {526, 527}, "only the strip's last columns read the unmirrored column"
    # a last column of 16 + 3 px or more has no such rect
    _, _, fr2 = frames.make_case(531, 300, mix=synth.MIX_ALL, gab=True, epf_iters=1, seed=24)
    assert np.array_equal(bits(fr2.decode(threads=4)), bits(_decode_ref_in_order(ref, fr2, "0,2,1")))


@pytest.mark.parametrize("xs,ys,gab,epf", [(540, 300, True, 3), (768, 520, True, 1), (535, 300, True, 3)])
def test_reference_threads_bit_identical_when_last_column_is_wide(ref, xs, ys, gab, epf):
    """... and with a last group column of at least 16 + 7 px (or none: widths that are multiples of 256, like the
    4K / 8K bench frames) any order and any thread count give the in-order bits."""
    _, _, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=gab, epf_iters=epf, seed=xs)
    one = _decode_ref_in_order(ref, fr, None)
    assert np.array_equal(bits(one), bits(_decode_ref_in_order(ref, fr, "2,5,1,4,0,3")))
    for _ in range(3):
        assert np.array_equal(bits(one), bits(_decode_ref_in_order(ref, fr, None, threads=8)))


def test_fma_build_of_the_reference_is_bit_identical(ref):
    """bench.py's cpu_baseline times oracle/_ref/libjxl_ref_fma.so: the same reference sources and Highway shim
    compiled -O3 -mavx2 -mfma.  Same IEEE arithmetic (MulAdd = fmaf either way, -ffp-contract=off): same bits."""
    if ref.ref_lib_fma() is None:
        pytest.skip("no AVX2 / FMA on this host")
    import frames
    from libjxl_amd import synth
    for (w, h, gab, epf) in ((520, 300, True, 1), (333, 268, True, 3), (256, 128, False, 0)):
        _, _, fr = frames.make_case(w, h, mix=synth.MIX_ALL, gab=gab, epf_iters=epf, seed=4)
        assert np.array_equal(fr.decode_ref(threads=1), fr.decode_ref(threads=1, fma_build=True))


def test_eight_lane_build_of_the_reference_hot_path_matches_the_checker(ref):
    """bench.py's cpu_baseline times oracle/_ref/libjxl_ref_v8.so: the reference's decode hot path (dec_group.cc with the
    inverse transforms, the Gaborish / EPF / XYB / write stages) compiled IN PLACE against oracle/hwy_shim_v -- 256-bit
    vectors, 8 float lanes, HWY_TARGET = HWY_AVX2: libjxl's vector DCTs, register transposes and 8-pixel filter steps
    instead of its one-lane paths.  Held to the single-lane checker within the reference's own executor tolerance
    (2e-4, render_pipeline_test.cc:321-327) on every strategy, every stage list, ragged sizes, int32 coefficients and
    an HDR intensity target -- it exercises every lane-crossing operation of the stand-in (Interleave / Concat /
    Broadcast / LoadDup128 / StoreInterleaved3 / conversions).  In practice the pixels are identical: the vector code
    performs the same operations per lane."""
    if ref.ref_lib_v8() is None:
        pytest.skip("no AVX2 / FMA on this host")
    import frames
    from libjxl_amd import synth
    cases = [(520, 300, synth.MIX_ALL, True, 1, {}), (333, 268, synth.MIX_ALL, True, 3, {}), (256, 128, synth.MIX_D1, False, 0, {}),
             (700, 500, synth.MIX_ALL, True, 2, {}), (129, 67, synth.MIX_ALL, True, 1, {}), (17, 9, synth.MIX_DCT8, True, 1, {}),
             (600, 300, synth.MIX_D1, True, 1, dict(coeff_type=1, amp=200000.0, decay=3.0)),
             (400, 300, synth.MIX_D1, True, 1, dict(intensity_target=4000.0, output_kind=0))]
    for (w, h, mix, gab, epf, kw) in cases:
        _, _, fr = frames.make_case(w, h, mix=mix, gab=gab, epf_iters=epf, seed=4 + w, **kw)
        a, b = fr.decode_ref(threads=2), fr.decode_ref(threads=2, v8_build=True)
        scale = max(1.0, float(np.abs(a).max()))
        assert float(np.abs(a - b).max()) / scale <= 2e-4, (w, h, gab, epf)
