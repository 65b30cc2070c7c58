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
    lib = C.CDLL(abi.library_path())
    lib.jxlhip_block_ctx_map_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.jxlhip_ac_pass_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.POINTER(C.c_void_p)]
    lib.jxlhip_ac_pass_destroy.argtypes = [C.c_void_p]
    lib.jxlhip_ac_pass_destroy.restype = None
    lib.jxlhip_ac_pass_max_num_bits.argtypes = [C.c_void_p]
    lib.jxlhip_ac_pass_max_num_bits.restype = C.c_uint32
    lib.jxlhip_ac_group_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                           C.c_uint32, C.c_uint32, C.c_void_p * 3, C.POINTER(C.c_size_t)]
    return lib


def ceil_log2(n):
    return (n - 1).bit_length()


def entropy_decode(L, rs):
    """AC global + AC group sections of rs through the product's entropy decoder.
    Returns (coeff_type, the three coefficient buffers)."""
    bctx = abi.BlockCtxMap()
    pos = C.c_size_t(0)
    b = rs.block_ctx_bytes
    assert L.jxlhip_block_ctx_map_decode(b.ctypes.data, len(b), C.byref(pos), C.byref(bctx)) == 0
    glob = np.frombuffer(rs.ac_global(), np.uint8)
    # ProcessACGlobal (dec_frame.cc:368-386): DequantMatrices::Decode -- one bit "all default" --
    # then num_histograms - 1 in CeilLog2Nonzero(num_groups) bits
    assert glob[0] & 1, "custom dequant matrices are not in scope"
    nbits = ceil_log2(rs.num_groups)
    num_histo = 1 + ((int(glob[0]) | int(glob[1]) << 8 | int(glob[2]) << 16) >> 1 & ((1 << nbits) - 1))
    assert num_histo == rs.num_histograms
    pos = C.c_size_t(1 + nbits)
    h = C.c_void_p()
    rc = L.jxlhip_ac_pass_decode(glob.ctypes.data, len(glob), C.byref(pos), rs.used_acs, num_histo,
                                 C.byref(bctx), C.byref(h))
    assert rc == 0, rc
    assert (pos.value + 7) // 8 == len(glob), (pos.value, len(glob))
    try:
        # dec_frame.cc:414-421: 16-bit coefficient buffers only when no token can carry 16 bits
        # (a flat histogram in the stream is enough to make the reference pick int32)
        ct = 0 if L.jxlhip_ac_pass_max_num_bits(h) < 16 else 1
        out = [np.zeros(rs.num_groups * 65536, np.int32 if ct else np.int16) for _ in range(3)]
        xsb, ysb, xsg = (rs.xsize + 7) // 8, (rs.ysize + 7) // 8, (rs.xsize + 255) // 256
        for g in range(rs.num_groups):
            d = np.frombuffer(rs.ac_group(g), np.uint8)
            gp, n = C.c_size_t(0), C.c_size_t(0)
            ptrs = (C.c_void_p * 3)(*[o[g * 65536:].ctypes.data for o in out])
            rc = L.jxlhip_ac_group_decode(h, xsb, ysb, g % xsg, g // xsg, rs.ac_strategy.ctypes.data,
                                          rs.raw_quant.ctypes.data, rs.quant_dc.ctypes.data, d.ctypes.data, len(d),
                                          C.byref(gp), 0, ct, ptrs, C.byref(n))
            assert rc == 0, (g, rc)
            assert (gp.value + 7) // 8 == len(d), (g, gp.value, len(d))
    finally:
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
    ct, coeffs = entropy_decode(L, rs)
    assert any(np.any(c) for c in coeffs)
    fr = rs.frame(coeffs)
    fr.c.p.output_kind = 1
    fr.c.p.coeff_type = ct
    out = fr.decode(threads=4)
    assert np.array_equal(out, rs.rgb), float(np.abs(out - rs.rgb).max())


@pytest.mark.gpu
@pytest.mark.parametrize("xs,ys,distance,tier,epf", [
    (512, 384, 1.0, 3, -1),
    (520, 300, 3.0, 3, -1),
    (384, 520, 2.0, 2, 1),
    (300, 300, 1.0, 7, 0),
    (2048, 1280, 1.0, 5, -1),   # 40 groups: the runner's threads race for the staging slots
])
def test_reference_encoded_stream_through_hip_path(ref, xs, ys, distance, tier, epf):
    """The whole product on a genuine stream, as a libjxl maintainer would wire it:
    side info to the device, AC sections entropy-decoded by runner threads straight
    into the pinned staging slots (jxlhip_ac_group_decode_submit), HIP decode."""
    import threading

    import torch

    from libjxl_amd import VarDctDecoder
    rs = ref.RealStream(xs, ys, seed=xs + ys, distance=distance, speed_tier=tier, epf=epf)
    d = VarDctDecoder(0)
    L = d.L
    params = abi.FrameParams.from_buffer_copy(rs.params.tobytes())
    params.output_kind = 1
    bctx = abi.BlockCtxMap()
    pos = C.c_size_t(0)
    b = rs.block_ctx_bytes
    assert L.jxlhip_block_ctx_map_decode(b.ctypes.data, len(b), C.byref(pos), C.byref(bctx)) == 0
    glob = np.frombuffer(rs.ac_global(), np.uint8)
    assert glob[0] & 1
    nbits = ceil_log2(rs.num_groups)
    num_histo = 1 + ((int(glob[0]) | int(glob[1]) << 8 | int(glob[2]) << 16) >> 1 & ((1 << nbits) - 1))
    pos = C.c_size_t(1 + nbits)
    h = C.c_void_p()
    assert L.jxlhip_ac_pass_decode(glob.ctypes.data, len(glob), C.byref(pos), rs.used_acs, num_histo,
                                   C.byref(bctx), C.byref(h)) == 0
    params.coeff_type = 0 if L.jxlhip_ac_pass_max_num_bits(h) < 16 else 1  # dec_frame.cc:414-421
    d.begin_frame(params)
    dc = [rs.dc_x, rs.dc_y, rs.dc_b]
    dc3 = (C.c_void_p * 3)(*[x.ctypes.data for x in dc])
    assert L.jxlhip_upload_side_info(d.ctx, rs.ac_strategy.ctypes.data, rs.raw_quant.ctypes.data,
                                     rs.epf_sharpness.ctypes.data, rs.ytox_map.ctypes.data,
                                     rs.ytob_map.ctypes.data, dc3, rs.dequant_table.ctypes.data) == 0
    errs = []

    def worker(tid, nthreads):
        for g in range(tid, rs.num_groups, nthreads):
            sec = np.frombuffer(rs.ac_group(g), np.uint8)
            gp = C.c_size_t(0)
            rc = L.jxlhip_ac_group_decode_submit(d.ctx, h, g, rs.ac_strategy.ctypes.data, rs.raw_quant.ctypes.data,
                                                 rs.quant_dc.ctypes.data, sec.ctypes.data, len(sec), C.byref(gp))
            if rc != 0 or (gp.value + 7) // 8 != len(sec):
                errs.append((g, rc, gp.value, len(sec)))

    threads = [threading.Thread(target=worker, args=(i, 12)) for i in range(12)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    got = d.decode_frame()
    d.sync()
    L.jxlhip_ac_pass_destroy(h)
    got = got.cpu().numpy()
    d.close()
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    err = float(np.abs(got - rs.rgb).max()) / scale
    assert err <= 2e-5, err
