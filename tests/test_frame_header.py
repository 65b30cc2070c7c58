"""f4: the frame header (FrameHeader / Passes / BlendingInfo / AnimationFrame / LoopFilter bundles,
lib/jxl/frame_header.cc, loop_filter.cc, fields.cc) -- differential test against the reference's
ReadFrameHeader: the same bytes (headers of genuine codestreams, and thousands of random bit
strings, which walk every conditional branch, every U32 / U64 / F16 coder and every rejection)
under the same image metadata must give the same verdict, the same number of bits and the same
fields.  CPU only."""
import ctypes as C
import struct

import numpy as np
import pytest

from libjxl_amd import abi


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    R = oracle.ref_lib()
    R.jxr_frame_header_read.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    return R


@pytest.fixture(scope="module")
def L():
    return abi.load_library()


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def flatten(h):
    """jxlhip_frame_header in the order oracle/ref_driver.cc jxr_frame_header_read writes."""
    kH, kV = (0, 1, 1, 0), (0, 1, 0, 1)
    o = [h.all_default, h.frame_type, h.is_modular, h.color_transform, h.flags]
    o += [kH[m] | (kV[m] << 4) for m in h.chroma_mode]
    o += [h.upsampling, h.group_size_shift, h.x_qm_scale, h.b_qm_scale, h.num_passes, h.num_downsample]
    o += list(h.shift[:h.num_passes]) + list(h.downsample[:h.num_downsample]) + list(h.last_pass[:h.num_downsample])
    o += [h.dc_level, h.custom_size_or_origin, h.x0 & 0xFFFFFFFFFFFFFFFF, h.y0 & 0xFFFFFFFFFFFFFFFF, h.coded_xsize,
          h.coded_ysize, h.blend_mode, h.blend_alpha_channel, h.blend_clamp, h.blend_source, h.duration, h.timecode,
          h.is_last, h.save_as_reference, h.save_before_color_transform, h.name_length, h.extensions]
    lf = h.lf
    o += [h.lf_all_default, lf.gab, h.gab_custom] + [f32bits(w) for w in lf.gab_weights]
    o += [lf.epf_iters, h.epf_sharp_custom] + [f32bits(v) for v in lf.epf_sharp_lut]
    o += [h.epf_weight_custom] + [f32bits(v) for v in lf.epf_channel_scale]
    o += [f32bits(h.epf_pass1_zeroflush), f32bits(h.epf_pass2_zeroflush), h.epf_sigma_custom]
    o += [f32bits(lf.epf_quant_mul), f32bits(lf.epf_pass0_sigma_scale), f32bits(lf.epf_pass2_sigma_scale),
          f32bits(lf.epf_border_sad_mul), f32bits(h.epf_sigma_for_modular), h.lf_extensions]
    o += [h.xsize, h.ysize, h.xsize_blocks, h.ysize_blocks, h.group_dim, h.xsize_groups, h.ysize_groups, h.num_groups,
          h.num_dc_groups]
    return [int(v) & 0xFFFFFFFFFFFFFFFF for v in o]


def both(L, R, data, xs, ys, xyb=1, num_ec=0, dim_shift=None, anim=0, tc=0, preview=0, bit_pos=0):
    d = np.frombuffer(data, np.uint8)
    ds = None if dim_shift is None else np.ascontiguousarray(dim_shift, np.uint8)
    info = abi.ImageInfo(xs, ys, xyb, num_ec, None if ds is None else ds.ctypes.data, anim, tc, preview)
    h = abi.FrameHeader()
    pos = C.c_size_t(bit_pos)
    rc = L.jxlhip_frame_header_decode(d.ctypes.data, len(d), C.byref(pos), C.byref(info), C.byref(h))
    out = np.zeros(160, np.uint64)
    bits = C.c_size_t(0)
    assert bit_pos == 0
    want_rc = R.jxr_frame_header_read(d.ctypes.data, len(d), xs, ys, xyb, num_ec, None if ds is None else ds.ctypes.data,
                                      anim, tc, preview, out.ctypes.data, C.byref(bits))
    return rc, h, pos.value, want_rc, out, bits.value


def test_headers_of_genuine_codestreams(L, ref, oracle):
    for kw in (dict(xsize=520, ysize=300, distance=1.0), dict(xsize=776, ysize=520, distance=3.0, progressive=1),
               dict(xsize=264, ysize=200, distance=1.5, progressive=2, epf=3), dict(xsize=200, ysize=136, epf=0)):
        rs = oracle.RealStream(seed=9, speed_tier=3, **kw)
        frame = rs.codestream[rs.frame_offset:].tobytes()
        rc, h, pos, want_rc, out, bits = both(L, ref, frame, rs.xsize, rs.ysize)
        assert rc == 0 and want_rc == 0
        assert pos == bits == rs.toc_bit_offset
        got = flatten(h)
        assert got == [int(v) for v in out[:len(got)]]
        # ... and they are what the decoder state of the harness says
        p = rs.frame_params
        assert h.num_passes == rs.num_passes and list(h.shift[:h.num_passes]) == rs.shift
        assert (h.num_groups, h.num_dc_groups) == (rs.num_groups, rs.num_dc_groups)
        assert h.num_toc_entries == len(rs.section_size)
        assert bytes(h.lf) == bytes(p.lf)
        assert (h.x_dm_multiplier, h.b_dm_multiplier) == (p.x_dm_multiplier, p.b_dm_multiplier)


@pytest.mark.parametrize("cfg", [
    dict(xyb=1), dict(xyb=0), dict(xyb=1, num_ec=1), dict(xyb=0, num_ec=3, dim_shift=[0, 1, 2]),
    dict(xyb=1, anim=1), dict(xyb=1, anim=1, tc=1, num_ec=2), dict(xyb=1, preview=1), dict(xyb=0, preview=1, num_ec=1),
])
def test_random_bit_strings_against_the_reference(L, ref, cfg):
    rng = np.random.default_rng(sum(ord(c) for c in str(sorted(cfg.items()))))
    agree_ok = agree_bad = 0
    for trial in range(2500):
        n = int(rng.integers(2, 96))
        b = rng.integers(0, 256, n, dtype=np.uint8)
        if trial % 3 == 0:
            b[0] &= 0xFE  # not all_default: go down the long path
        if trial % 5 == 0:
            b[: n // 2] &= rng.integers(0, 256, dtype=np.uint8)  # sparser bits: smaller selectors, shorter paths
        data = b.tobytes()
        xs, ys = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        rc, h, pos, want_rc, out, bits = both(L, ref, data, xs, ys, **cfg)
        assert (rc == 0) == (want_rc == 0), (trial, rc, want_rc, data.hex())
        if rc == 0:
            agree_ok += 1
            assert pos == bits, (trial, pos, bits, data.hex())
            got = flatten(h)
            assert got == [int(v) for v in out[:len(got)]], (trial, data.hex())
        else:
            agree_bad += 1
    assert agree_ok > 200 and agree_bad > 200, (agree_ok, agree_bad)


def dc_both(L, R, data):
    R.jxr_dc_global_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    d = np.frombuffer(data, np.uint8)
    g = abi.DcGlobal()
    pos = C.c_size_t(0)
    rc = L.jxlhip_dc_global_decode(d.ctypes.data, len(d), C.byref(pos), 0, C.byref(g))
    out = np.zeros(16 + 3 * 13 * 64, np.uint64)
    bits = C.c_size_t(0)
    want_rc = R.jxr_dc_global_read(d.ctypes.data, len(d), out.ctypes.data, C.byref(bits))
    m = g.block_ctx_map
    got = [f32bits(v) for v in g.dc_quant] + [g.global_scale, g.quant_dc, g.cfl_color_factor, f32bits(g.cfl_base_x),
                                              f32bits(g.cfl_base_b), g.ytox_dc & 0xFFFFFFFFFFFFFFFF,
                                              g.ytob_dc & 0xFFFFFFFFFFFFFFFF, m.num_dc_ctxs, m.num_qf_thresholds,
                                              m.ctx_map_size] + list(m.ctx_map[:m.ctx_map_size])
    return rc, got, pos.value, want_rc, [int(v) for v in out[:len(got)]], bits.value, g


def test_dc_global_of_genuine_codestreams(L, ref, oracle):
    for kw in (dict(xsize=520, ysize=300, distance=1.0), dict(xsize=776, ysize=520, distance=3.0)):
        rs = oracle.RealStream(seed=9, speed_tier=3, **kw)
        rc, got, pos, want_rc, want, bits, g = dc_both(L, ref, rs.section(0))
        assert rc == 0 and want_rc == 0 and pos == bits and got == want
        p = rs.frame_params
        assert (g.global_scale, g.quant_dc) == (p.global_scale, p.quant_dc)
        assert (g.cfl_base_x, g.cfl_base_b, g.cfl_color_factor) == (p.cfl_base_x, p.cfl_base_b, p.cfl_color_factor)
        # the block context map is the one the AC decoder of this stream was given
        m = abi.BlockCtxMap()
        q = C.c_size_t(0)
        b = rs.block_ctx_bytes
        assert L.jxlhip_block_ctx_map_decode(b.ctypes.data, len(b), C.byref(q), C.byref(m)) == 0
        assert bytes(m) == bytes(g.block_ctx_map)
    assert L.jxlhip_dc_global_decode(b.ctypes.data, len(b), C.byref(C.c_size_t(0)), 2, C.byref(abi.DcGlobal())) == -7


def test_dc_global_random_bit_strings_against_the_reference(L, ref):
    rng = np.random.default_rng(77)
    ok = bad = 0
    for trial in range(4000):
        n = int(rng.integers(2, 64))
        b = rng.integers(0, 256, n, dtype=np.uint8)
        if trial % 2 == 0:
            b[: n // 2] &= rng.integers(0, 256, dtype=np.uint8)
        if trial % 4 == 1:
            b[0] |= 1  # default DC quant
        rc, got, pos, want_rc, want, bits, _ = dc_both(L, ref, b.tobytes())
        assert (rc == 0) == (want_rc == 0), (trial, rc, want_rc, b.tobytes().hex())
        if rc == 0:
            ok += 1
            assert pos == bits and got == want, (trial, b.tobytes().hex())
        else:
            bad += 1
    assert ok > 100 and bad > 100, (ok, bad)
