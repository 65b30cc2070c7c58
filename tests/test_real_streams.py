"""The product on GENUINE libjxl streams: the reference's full encoder
(jxl::EncodeFrame -- ac-strategy search, adaptive quantisation, CfL, loop-filter
choice, coefficient orders, clustered ANS histograms; compiled in place,
oracle/ref_real_stream.cc) writes a VarDCT codestream of a procedural image, the
reference's FrameDecoder decodes it, and the side info it parsed is handed to the
product's boundary.  The AC-global / AC-group SECTION BYTES of that codestream go
through the product's own entropy decoder (include/jxl_hip_entropy.h); the
coefficients it returns are rendered by the C oracle here (CPU) and by the HIP
path in test_gpu_vs_reference.py.  The pixels must be the reference decoder's."""
import ctypes as C

import numpy as np
import pytest

from libjxl_amd import abi


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    return oracle


@pytest.fixture(scope="module")
def L():
    return abi.load_library()


def ceil_log2(n):
    return (n - 1).bit_length()


def open_ac_global(L, rs):
    """ProcessACGlobal (dec_frame.cc:372-421) on the stream's AC-global section: the block
    context map from DC global, then jxlhip_ac_global_decode.  Returns (pass handles,
    coefficient type)."""
    bctx = abi.BlockCtxMap()
    pos = C.c_size_t(0)
    b = rs.block_ctx_bytes
    assert L.jxlhip_block_ctx_map_decode(b.ctypes.data, len(b), C.byref(pos), C.byref(bctx)) == 0
    glob = np.frombuffer(rs.ac_global(), np.uint8)
    encs = abi.QuantEncodings()
    nh, used, hs = C.c_uint32(0), C.c_size_t(0), (C.c_void_p * rs.num_passes)()
    rc = L.jxlhip_ac_global_decode(glob.ctypes.data, len(glob), rs.num_groups, rs.num_passes, rs.used_acs,
                                   C.byref(bctx), C.byref(encs), C.byref(nh), hs, C.byref(used))
    assert rc == 0, rc
    assert nh.value == rs.num_histograms
    assert (used.value + 7) // 8 == len(glob), (used.value, len(glob))
    assert all(e.mode == abi.QUANT_LIBRARY for e in encs)  # the encoder never writes custom matrices
    # dec_frame.cc:414-421: 16-bit coefficient buffers only when no token, summed over the passes,
    # can carry 16 bits (a flat histogram in the stream is enough to make the reference pick int32)
    mx = max(L.jxlhip_ac_pass_max_num_bits(h) for h in hs) + ceil_log2(rs.num_passes)
    return list(hs), 0 if mx < 16 else 1


def entropy_decode(L, rs, force_ct=None):
    """AC global + every pass of every AC group section of rs through the product's
    entropy decoder.  Returns (coeff_type, the three coefficient buffers)."""
    hs, ct = open_ac_global(L, rs)
    if force_ct is not None:
        ct = force_ct
    try:
        out = [np.zeros(rs.num_groups * 65536, np.int32 if ct else np.int16) for _ in range(3)]
        xsb, ysb, xsg = (rs.xsize + 7) // 8, (rs.ysize + 7) // 8, (rs.xsize + 255) // 256
        for g in range(rs.num_groups):
            ptrs = (C.c_void_p * 3)(*[o[g * 65536:].ctypes.data for o in out])
            for p in range(rs.num_passes):  # passes accumulate, each << its shift (dec_group.cc:520-526)
                d = np.frombuffer(rs.ac_group(g, p), np.uint8)
                gp, n = C.c_size_t(0), C.c_size_t(0)
                rc = L.jxlhip_ac_group_decode(hs[p], xsb, ysb, g % xsg, g // xsg, rs.ac_strategy.ctypes.data,
                                              rs.raw_quant.ctypes.data, rs.quant_dc.ctypes.data, d.ctypes.data,
                                              len(d), C.byref(gp), rs.shift[p], ct, ptrs, C.byref(n))
                assert rc == 0, (g, p, rc)
                assert (gp.value + 7) // 8 == len(d), (g, p, gp.value, len(d))
    finally:
        for h in hs:
            L.jxlhip_ac_pass_destroy(h)
    return ct, out


@pytest.mark.parametrize("xs,ys,distance,tier,epf", [
    (512, 384, 1.0, 3, -1),   # squirrel d1: every strategy family, Gaborish + 2 EPF iterations
    (520, 300, 3.0, 3, -1),   # ragged, d3: 3 EPF iterations
    (640, 264, 0.5, 5, -1),   # hare d0.5
    (384, 520, 2.0, 2, 1),    # kitten, EPF forced to 1 iteration
    (300, 300, 1.0, 7, 0),    # falcon (DCT8 only), no EPF
    (1024, 1280, 1.0, 5, -1), # 20 groups; a flat histogram cluster -> the reference's int32 buffers
])
def test_reference_encoded_stream_decodes_to_reference_pixels(L, ref, xs, ys, distance, tier, epf):
    rs = ref.RealStream(xs, ys, seed=xs + ys, distance=distance, speed_tier=tier, epf=epf)
    check_cpu(L, rs)


@pytest.mark.parametrize("progressive,passes", [(1, 3), (2, 2)])
def test_progressive_passes_accumulate_to_reference_pixels(L, ref, progressive, passes):
    """--progressive_ac (3 passes split by frequency band) and --qprogressive_ac (2 passes,
    the first one shifted by 1): every pass has its own coefficient orders and histograms."""
    rs = ref.RealStream(520, 392, seed=11, distance=1.5, speed_tier=3, progressive=progressive)
    assert rs.num_passes == passes
    assert (max(rs.shift) > 0) == (progressive == 2)
    check_cpu(L, rs)


def test_optimistic_16_bit_buffers_on_a_flat_histogram_stream(L, ref):
    """The reference switches to int32 coefficient buffers as soon as a token COULD carry 16 bits
    (a flat histogram does that); the values themselves fit 16 bits, and the decoder says so:
    decoding into int16 succeeds (no JXLHIP_ERR_RANGE) and gives the same coefficients."""
    rs = ref.RealStream(1024, 1280, seed=2304, distance=1.0, speed_tier=5)
    ct, c32 = entropy_decode(L, rs)
    assert ct == 1  # what dec_frame.cc:414-421 would pick
    _, c16 = entropy_decode(L, rs, force_ct=0)
    for a, b in zip(c16, c32):
        assert a.dtype == np.int16 and np.array_equal(a.astype(np.int32), b)


def check_cpu(L, rs):
    ct, coeffs = entropy_decode(L, rs)
    assert any(np.any(c) for c in coeffs)
    fr = rs.frame(coeffs)
    fr.c.p.output_kind = 1
    fr.c.p.coeff_type = ct
    out = fr.decode(threads=4)
    assert np.array_equal(out, rs.rgb), float(np.abs(out - rs.rgb).max())


@pytest.mark.gpu
@pytest.mark.parametrize("xs,ys,distance,tier,epf,progressive", [
    (512, 384, 1.0, 3, -1, 0),
    (520, 300, 3.0, 3, -1, 0),
    (384, 520, 2.0, 2, 1, 0),
    (300, 300, 1.0, 7, 0, 0),
    (2048, 1280, 1.0, 5, -1, 0),   # 40 groups: the runner's threads race for the staging slots; int32
    (520, 392, 1.5, 3, -1, 1),     # 3 passes by frequency band
    (520, 392, 1.5, 3, -1, 2),     # 2 passes, the first shifted by 1
])
def test_reference_encoded_stream_through_hip_path(L, ref, xs, ys, distance, tier, epf, progressive):
    """The whole product on a genuine stream, as a libjxl maintainer would wire it:
    side info to the device, AC sections entropy-decoded by runner threads straight
    into the pinned staging slots (jxlhip_ac_group_decode_submit_passes), HIP decode."""
    import threading

    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(xs, ys, seed=xs + ys, distance=distance, speed_tier=tier, epf=epf, progressive=progressive)
    hs, ct = open_ac_global(L, rs)
    d = VarDctDecoder(0)
    params = abi.FrameParams.from_buffer_copy(rs.params.tobytes())
    params.output_kind = 1
    params.coeff_type = ct
    d.begin_frame(params)
    dq = d.dequant_tables(None)   # the stream's encodings are all-library (checked in open_ac_global)
    d.sync()
    dqh = dq.cpu().numpy()
    assert np.array_equal(dqh, rs.dequant_table)
    dc = [rs.dc_x, rs.dc_y, rs.dc_b]
    dc3 = (C.c_void_p * 3)(*[x.ctypes.data for x in dc])
    assert L.jxlhip_upload_side_info(d.ctx, rs.ac_strategy.ctypes.data, rs.raw_quant.ctypes.data,
                                     rs.epf_sharpness.ctypes.data, rs.ytox_map.ctypes.data,
                                     rs.ytob_map.ctypes.data, dc3, dqh.ctypes.data) == 0
    errs = []
    n = rs.num_passes
    pass_arr = (C.c_void_p * n)(*hs)
    shifts = (C.c_uint32 * n)(*rs.shift)

    def worker(tid, nthreads):
        for g in range(tid, rs.num_groups, nthreads):
            secs = [np.frombuffer(rs.ac_group(g, p), np.uint8) for p in range(n)]
            datas = (C.c_void_p * n)(*[x.ctypes.data for x in secs])
            sizes = (C.c_size_t * n)(*[len(x) for x in secs])
            pos = (C.c_size_t * n)()
            rc = L.jxlhip_ac_group_decode_submit_passes(d.ctx, n, pass_arr, shifts, g, rs.ac_strategy.ctypes.data,
                                                        rs.raw_quant.ctypes.data, rs.quant_dc.ctypes.data, datas,
                                                        sizes, pos)
            if rc != 0 or any((pos[p] + 7) // 8 != len(secs[p]) for p in range(n)):
                errs.append((g, rc, list(pos), [len(x) for x in secs]))

    threads = [threading.Thread(target=worker, args=(i, 12)) for i in range(12)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    got = d.decode_frame()
    d.sync()
    for h in hs:
        L.jxlhip_ac_pass_destroy(h)
    got = got.cpu().numpy()
    d.close()
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    err = float(np.abs(got - rs.rgb).max()) / scale
    assert err <= 2e-5, err


@pytest.mark.gpu
def test_all_groups_on_the_parallel_runner(L, ref):
    """jxlhip_ac_groups_decode_submit: every AC group of a genuine two-pass stream on the
    JxlParallelRunner of libjxl_threads_hip.so (and on the calling thread), one C call."""
    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(776, 520, seed=21, distance=1.5, speed_tier=3, progressive=2)
    hs, ct = open_ac_global(L, rs)
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    pool = R.JxlThreadParallelRunnerCreate(None, 6)
    n, ng = rs.num_passes, rs.num_groups
    secs = [np.frombuffer(rs.ac_group(g, p), np.uint8) for p in range(n) for g in range(ng)]
    ptrs = (C.c_void_p * len(secs))(*[x.ctypes.data for x in secs])
    sizes = (C.c_size_t * len(secs))(*[len(x) for x in secs])
    pass_arr = (C.c_void_p * n)(*hs)
    shifts = (C.c_uint32 * n)(*rs.shift)
    d = VarDctDecoder(0)
    params = abi.FrameParams.from_buffer_copy(rs.params.tobytes())
    params.output_kind = 1
    params.coeff_type = 0  # optimistic 16-bit buffers whatever max_num_bits says
    outs = []
    for runner in (C.cast(R.JxlThreadParallelRunner, C.c_void_p), None):
        d.begin_frame(params)
        dc3 = (C.c_void_p * 3)(rs.dc_x.ctypes.data, rs.dc_y.ctypes.data, rs.dc_b.ctypes.data)
        assert L.jxlhip_upload_side_info(d.ctx, rs.ac_strategy.ctypes.data, rs.raw_quant.ctypes.data,
                                         rs.epf_sharpness.ctypes.data, rs.ytox_map.ctypes.data,
                                         rs.ytob_map.ctypes.data, dc3, rs.dequant_table.ctypes.data) == 0
        rc = L.jxlhip_ac_groups_decode_submit(d.ctx, runner, pool, n, pass_arr, shifts, rs.ac_strategy.ctypes.data,
                                              rs.raw_quant.ctypes.data, rs.quant_dc.ctypes.data, ptrs, sizes)
        assert rc == 0, rc
        out = d.decode_frame()
        d.sync()
        outs.append(out.cpu().numpy())
    R.JxlThreadParallelRunnerDestroy(pool)
    for h in hs:
        L.jxlhip_ac_pass_destroy(h)
    d.close()
    assert np.array_equal(outs[0], outs[1])
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    assert float(np.abs(outs[0] - rs.rgb).max()) / scale <= 2e-5


@pytest.mark.gpu
def test_range_error_reaches_the_caller_of_the_runner_path(L, ref):
    """A coefficient beyond 16 bits in an optimistically 16-bit frame: jxlhip_ac_groups_decode_submit
    returns JXLHIP_ERR_RANGE, and the same frame redone with int32 buffers decodes."""
    import frames
    from libjxl_amd import VarDctDecoder, synth
    xs, ys = 520, 264
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_DCT32, gab=False, epf_iters=0, coeff_type=1, quant_mul=2.0,
                                     amp=40.0)
    coeffs = [c.numpy() for c in t["coeffs"]]
    coeffs[1][5] = 40000
    glob, groups, used_acs, _ = fr.encode_ac_ref()
    g = np.frombuffer(glob, np.uint8)
    pos, h = C.c_size_t(0), C.c_void_p()
    assert L.jxlhip_ac_pass_decode(g.ctypes.data, len(g), C.byref(pos), used_acs, 1, None, C.byref(h)) == 0
    secs = [np.frombuffer(x, np.uint8) if len(x) else np.zeros(1, np.uint8) for x in groups]
    ptrs = (C.c_void_p * len(secs))(*[x.ctypes.data for x in secs])
    sizes = (C.c_size_t * len(secs))(*[len(x) for x in groups])
    pass_arr = (C.c_void_p * 1)(h)
    npy = {k: (v.numpy() if not isinstance(v, list) else [x.numpy() for x in v]) for k, v in t.items()}
    d = VarDctDecoder(0)
    dq = d.default_dequant_tables().cpu().numpy()
    results = []
    for ct in (0, 1):
        d.begin_frame(dict(params, coeff_type=ct, output_kind=1))
        dc3 = (C.c_void_p * 3)(*[x.ctypes.data for x in npy["dc"]])
        assert L.jxlhip_upload_side_info(d.ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                         npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                         npy["ytob_map"].ctypes.data, dc3, dq.ctypes.data) == 0
        results.append(L.jxlhip_ac_groups_decode_submit(d.ctx, None, None, 1, pass_arr, None, npy["ac_strategy"].ctypes.data,
                                                        npy["raw_quant"].ctypes.data, None, ptrs, sizes))
    assert results == [-8, 0]
    out = d.decode_frame()
    d.sync()
    want = fr.decode_ref(threads=1)
    L.jxlhip_ac_pass_destroy(h)
    d.close()
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(out.cpu().numpy() - want).max()) / scale <= 2e-5
