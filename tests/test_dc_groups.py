"""f4: the Modular-coded parts of a VarDCT frame -- the global MA tree in the DC-global section and
the DC groups (quantized DC, strategy map, quant field, EPF sharpness, colour-correlation maps):
jxlhip_modular_global_decode + jxlhip_dc_group_decode on the sections of genuine codestreams
(written by the reference's encoder: learned MA trees, its choice of predictors incl. the
self-correcting one, clustered histograms) must reproduce, bit for bit, what the reference's
ModularFrameDecoder left in PassesSharedState for the same stream.  CPU only."""
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


def parse_to_sections(L, rs):
    cs = np.ascontiguousarray(rs.codestream)
    base, n = cs.ctypes.data, len(cs)
    ih, pos = abi.ImageHeader(), C.c_size_t(0)
    assert L.jxlhip_image_header_decode(base, n, C.byref(pos), None, 0, C.byref(ih)) == 0
    info = abi.ImageInfo(ih.xsize, ih.ysize, ih.xyb_encoded, 0, None, 0, 0, 0)
    fh = abi.FrameHeader()
    assert L.jxlhip_frame_header_decode(base, n, C.byref(pos), C.byref(info), C.byref(fh)) == 0
    nt = int(fh.num_toc_entries)
    off, sz, total = np.zeros(nt, np.uint64), np.zeros(nt, np.uint32), C.c_uint64(0)
    assert L.jxlhip_toc_decode(base, n, C.byref(pos), nt, off.ctypes.data, sz.ctypes.data, C.byref(total)) == 0
    start = pos.value // 8
    return cs, ih, fh, [cs[start + int(o): start + int(o) + int(s)] for o, s in zip(off, sz)]


def decode_side_info(L, fh, sections):
    """DC global (both halves) and every DC group; returns the frame-level arrays."""
    s0 = sections[0]
    dcg, dpos = abi.DcGlobal(), C.c_size_t(0)
    assert L.jxlhip_dc_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), fh.flags, C.byref(dcg)) == 0
    tree = C.c_void_p()
    assert L.jxlhip_modular_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), C.byref(fh), C.byref(tree)) == 0
    assert (dpos.value + 7) // 8 == len(s0)  # the DC-global section is consumed exactly
    xsb, ysb = fh.xsize_blocks, fh.ysize_blocks
    qdc = [np.zeros(xsb * ysb, np.int32) for _ in range(3)]
    acs = np.zeros(xsb * ysb, np.uint8)
    rq = np.zeros(xsb * ysb, np.int32)
    sharp = np.zeros(xsb * ysb, np.uint8)
    cw, chh = (xsb + 7) // 8, (ysb + 7) // 8
    ytox, ytob = np.zeros(cw * chh, np.int8), np.zeros(cw * chh, np.int8)
    used, prec = C.c_uint32(0), []
    try:
        for g in range(int(fh.num_dc_groups)):
            d = sections[1 + g]
            gp, ep = C.c_size_t(0), C.c_uint32(0)
            ptrs = (C.c_void_p * 3)(*[q.ctypes.data for q in qdc])
            rc = L.jxlhip_dc_group_decode(tree, d.ctypes.data, len(d), C.byref(gp), C.byref(fh), g, ptrs, C.byref(ep),
                                          acs.ctypes.data, rq.ctypes.data, sharp.ctypes.data, ytox.ctypes.data,
                                          ytob.ctypes.data, C.byref(used))
            assert rc == 0, (g, rc)
            assert (gp.value + 7) // 8 == len(d), (g, gp.value, len(d))
            prec.append(ep.value)
    finally:
        L.jxlhip_modular_tree_destroy(tree)
    return dcg, qdc, prec, acs, rq, sharp, ytox, ytob, used.value


@pytest.mark.parametrize("kw", [
    dict(xsize=520, ysize=300, distance=1.0, speed_tier=3),
    dict(xsize=776, ysize=520, distance=3.0, speed_tier=3),
    dict(xsize=640, ysize=264, distance=0.5, speed_tier=5),
    dict(xsize=384, ysize=520, distance=2.0, speed_tier=2),
    dict(xsize=300, ysize=300, distance=1.0, speed_tier=7),
    dict(xsize=2200, ysize=264, distance=1.5, speed_tier=4),   # two DC groups side by side
])
def test_dc_groups_of_genuine_codestreams(L, ref, kw):
    rs = ref.RealStream(seed=13, **kw)
    cs, ih, fh, sections = parse_to_sections(L, rs)
    dcg, qdc, prec, acs, rq, sharp, ytox, ytob, used = decode_side_info(L, fh, sections)
    assert np.array_equal(acs, rs.ac_strategy.ravel())
    first = (acs & 1) == 1
    assert np.array_equal(rq[first], rs.raw_quant.ravel()[first])
    assert np.array_equal(sharp, rs.epf_sharpness.ravel())
    assert np.array_equal(ytox, rs.ytox_map.ravel()) and np.array_equal(ytob, rs.ytob_map.ravel())
    assert used == rs.used_acs
    # the quantized DC through the reference's own DequantDC + AdaptiveDCSmoothing = its DC image
    assert len(set(prec)) == 1
    xsb, ysb = fh.xsize_blocks, fh.ysize_blocks
    smooth = not (fh.flags & 128)
    f32 = np.float32
    mul = f32(1.0) / f32(1 << prec[0])
    inv_quant_dc = (f32(65536.0) / f32(dcg.global_scale)) / f32(dcg.quant_dc)   # quantizer.h:133-139
    mul_dc = [f32(inv_quant_dc * f32(dcg.dc_quant[c])) for c in range(3)]
    color_scale = f32(1.0) / f32(dcg.cfl_color_factor)                           # chroma_from_luma.h
    cfl_x = float(f32(dcg.cfl_base_x) + f32(dcg.ytox_dc) * color_scale)
    cfl_b = float(f32(dcg.cfl_base_b) + f32(dcg.ytob_dc) * color_scale)
    got = ref.ref_dequant_dc([q.reshape(ysb, xsb) for q in qdc], mul_dc, cfl_x, cfl_b, smooth, mul=float(mul))
    for c, want in enumerate((rs.dc_x, rs.dc_y, rs.dc_b)):
        assert np.array_equal(got[c].ravel(), want.ravel()), c


def _dc_groups_both_ways(L, fh, sections, monkeypatch, damage=None):
    """Every DC group through the channel loops of their own (modular.inc: DecodeWpPureChannel, DecodeSimpleChannel) and
    through the general loop (JXLHIP_WP_GENERAL=1): status, bits consumed and every output array."""
    import os
    s0 = sections[0]
    dcg, dpos = abi.DcGlobal(), C.c_size_t(0)
    assert L.jxlhip_dc_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), fh.flags, C.byref(dcg)) == 0
    tree = C.c_void_p()
    assert L.jxlhip_modular_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), C.byref(fh), C.byref(tree)) == 0
    xsb, ysb = fh.xsize_blocks, fh.ysize_blocks
    cw, chh = (xsb + 7) // 8, (ysb + 7) // 8
    results = []
    try:
        for general in (False, True):
            if general:
                monkeypatch.setenv("JXLHIP_WP_GENERAL", "1")
                abi.load_library().jxlhip_debug_reload_env()
            else:
                monkeypatch.delenv("JXLHIP_WP_GENERAL", raising=False)
                abi.load_library().jxlhip_debug_reload_env()
            qdc = [np.zeros(xsb * ysb, np.int32) for _ in range(3)]
            acs, rq, sharp = np.zeros(xsb * ysb, np.uint8), np.zeros(xsb * ysb, np.int32), np.zeros(xsb * ysb, np.uint8)
            ytox, ytob = np.zeros(cw * chh, np.int8), np.zeros(cw * chh, np.int8)
            used, status = C.c_uint32(0), []
            for g in range(int(fh.num_dc_groups)):
                d = sections[1 + g] if damage is None else damage(g, sections[1 + g])
                gp, ep = C.c_size_t(0), C.c_uint32(0)
                ptrs = (C.c_void_p * 3)(*[q.ctypes.data for q in qdc])
                rc = L.jxlhip_dc_group_decode(tree, d.ctypes.data, len(d), C.byref(gp), C.byref(fh), g, ptrs, C.byref(ep),
                                              acs.ctypes.data, rq.ctypes.data, sharp.ctypes.data, ytox.ctypes.data,
                                              ytob.ctypes.data, C.byref(used))
                status.append((rc, gp.value if rc == 0 else -1, ep.value if rc == 0 else -1))
            results.append((status, [q.copy() for q in qdc], acs, rq, sharp, ytox, ytob, used.value))
    finally:
        L.jxlhip_modular_tree_destroy(tree)
        monkeypatch.delenv("JXLHIP_WP_GENERAL", raising=False)
        abi.load_library().jxlhip_debug_reload_env()
    return results


def _same(a, b):
    assert a[0] == b[0], (a[0], b[0])
    ok = [g for g, st in enumerate(a[0]) if st[0] == 0]
    if len(ok) == len(a[0]):  # (a failed group leaves its rectangle half written: compare the frames only when all went through)
        for x, y in zip(a[1], b[1]):
            assert np.array_equal(x, y)
        for x, y in zip(a[2:7], b[2:7]):
            assert np.array_equal(x, y)
        assert a[7] == b[7]
    return len(ok)


def test_channel_loops_of_their_own_equal_the_general_loop(L, ref, monkeypatch):
    """The self-correcting predictor's loop and the no-predictor-state loop are special cases of the general channel loop:
    same samples, same bits consumed -- on genuine streams (the 8K d1.0 stream of tests/data included) and on damaged DC
    group sections, where both must stop (or not) at the same place."""
    import os
    streams = []
    for kw in (dict(xsize=520, ysize=300, distance=1.0, speed_tier=3), dict(xsize=2200, ysize=264, distance=1.5, speed_tier=4),
               dict(xsize=300, ysize=300, distance=1.0, speed_tier=7)):
        streams.append(ref.RealStream(seed=13, **kw))
    path = os.path.join(os.path.dirname(__file__), "data", "e2e_8k_d1.jxl")
    if os.path.exists(path):
        class _File:
            codestream = np.fromfile(path, np.uint8)
        streams.append(_File())
    rng = np.random.default_rng(5)
    went_through = stopped = 0
    for rs in streams:
        cs, ih, fh, sections = parse_to_sections(L, rs)
        a, b = _dc_groups_both_ways(L, fh, sections, monkeypatch)
        assert _same(a, b) == int(fh.num_dc_groups)
        if len(cs) > (1 << 20):
            continue  # (the 8K stream: intact only -- a damaged copy of each of its groups costs seconds)
        for trial in range(40):
            flips = {}

            def damage(g, d, flips=flips):
                if g not in flips:
                    e = np.array(d)
                    for _ in range(int(rng.integers(1, 4))):
                        k = int(rng.integers(0, len(e) * 8))
                        e[k // 8] ^= np.uint8(1 << (k % 8))
                    if trial % 5 == 0:
                        e = e[: int(rng.integers(1, len(e) + 1))].copy()
                    flips[g] = e
                return flips[g]

            a, b = _dc_groups_both_ways(L, fh, sections, monkeypatch, damage)
            n = _same(a, b)
            went_through += n
            stopped += int(fh.num_dc_groups) - n
    assert stopped > 50, (went_through, stopped)  # (nearly every flip derails the ANS stream: both loops must say so)


def test_staged_dc_group_announces_the_block_info_before_the_sharpness_channel(L, ref):
    """jxlhip_dc_group_decode_staged: the callback fires once per group, on the calling thread, when the group's rectangles
    of quant_dc / ac_strategy / raw_quant (and its used_acs bits) already hold their final values -- what the AC groups
    under it read -- and the call then ends with the same outputs as jxlhip_dc_group_decode."""
    rs = ref.RealStream(seed=13, xsize=2200, ysize=264, distance=1.5, speed_tier=4)
    cs, ih, fh, sections = parse_to_sections(L, rs)
    want = decode_side_info(L, fh, sections)
    s0 = sections[0]
    dcg, dpos = abi.DcGlobal(), C.c_size_t(0)
    assert L.jxlhip_dc_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), fh.flags, C.byref(dcg)) == 0
    tree = C.c_void_p()
    assert L.jxlhip_modular_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), C.byref(fh), C.byref(tree)) == 0
    xsb, ysb = fh.xsize_blocks, fh.ysize_blocks
    qdc = [np.zeros(xsb * ysb, np.int32) for _ in range(3)]
    acs, rq, sharp = np.zeros(xsb * ysb, np.uint8), np.zeros(xsb * ysb, np.int32), np.full(xsb * ysb, 99, np.uint8)
    cw, chh = (xsb + 7) // 8, (ysb + 7) // 8
    ytox, ytob = np.zeros(cw * chh, np.int8), np.zeros(cw * chh, np.int8)
    used = C.c_uint32(0)
    seen = []
    gd = fh.group_dim
    xdg = (xsb + gd - 1) // gd

    @C.CFUNCTYPE(None, C.c_void_p)
    def ready(opaque):
        g = C.cast(opaque, C.POINTER(C.c_uint32))[0]
        x0, y0 = (g % xdg) * gd, (g // xdg) * gd
        rect = np.zeros((ysb, xsb), bool)
        rect[y0:y0 + gd, x0:x0 + gd] = True
        rect = rect.ravel()
        seen.append((g, np.array_equal(acs[rect], want[3][rect]), np.array_equal(rq[rect], want[4][rect]),
                     all(np.array_equal(q[rect], w[rect]) for q, w in zip(qdc, want[1])),
                     bool((sharp[rect] == 99).all())))

    try:
        for g in range(int(fh.num_dc_groups)):
            d = sections[1 + g]
            gp, ep, gid = C.c_size_t(0), C.c_uint32(0), C.c_uint32(g)
            ptrs = (C.c_void_p * 3)(*[q.ctypes.data for q in qdc])
            rc = L.jxlhip_dc_group_decode_staged(tree, d.ctypes.data, len(d), C.byref(gp), C.byref(fh), g, ptrs, C.byref(ep),
                                                 acs.ctypes.data, rq.ctypes.data, sharp.ctypes.data, ytox.ctypes.data,
                                                 ytob.ctypes.data, C.byref(used), C.cast(ready, C.c_void_p), C.addressof(gid))
            assert rc == 0 and (gp.value + 7) // 8 == len(d)
    finally:
        L.jxlhip_modular_tree_destroy(tree)
    assert [s[0] for s in seen] == list(range(int(fh.num_dc_groups)))
    # at the callback: strategy map, quant field and quantized DC final; not one sharpness value of the rectangle written yet
    assert all(s[1] and s[2] and s[3] and s[4] for s in seen), seen
    assert np.array_equal(acs, want[3]) and np.array_equal(sharp, want[5]) and used.value == want[8]
    assert np.array_equal(ytox, want[6]) and np.array_equal(ytob, want[7])


def test_damaged_global_trees_fail_like_the_reference(L, ref):
    """DecodeTree + ValidateTree + DecodeHistograms (modular/encoding/dec_ma.cc) on damaged DC-global
    sections: same verdict and, when accepted, the same number of bits as the reference."""
    R = ref.ref_lib()
    R.jxr_modular_tree_read.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(77)
    agree_ok = agree_bad = 0
    for kw in (dict(xsize=520, ysize=300, distance=1.0, speed_tier=3), dict(xsize=640, ysize=264, distance=0.5, speed_tier=5),
               dict(xsize=384, ysize=520, distance=2.0, speed_tier=2)):
        rs = ref.RealStream(seed=13, **kw)
        cs, ih, fh, sections = parse_to_sections(L, rs)
        s0 = np.array(sections[0])
        dcg, dpos = abi.DcGlobal(), C.c_size_t(0)
        assert L.jxlhip_dc_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), fh.flags, C.byref(dcg)) == 0
        start = dpos.value
        assert (s0[start // 8] >> (start % 8)) & 1  # has_tree
        limit = min(1 << 22, 1024 + fh.xsize * fh.ysize * 3 // 16)
        for trial in range(400):
            b = s0.copy()
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(start + 1, len(b) * 8))
                b[k // 8] ^= np.uint8(1 << (k % 8))
            if trial % 7 == 0:
                b = b[: int(rng.integers(start // 8 + 2, len(b) + 1))].copy()
            pos, tree = C.c_size_t(start), C.c_void_p()
            rc = L.jxlhip_modular_global_decode(b.ctypes.data, len(b), C.byref(pos), C.byref(fh), C.byref(tree))
            L.jxlhip_modular_tree_destroy(tree)
            out = np.zeros(8, np.uint64)
            want = R.jxr_modular_tree_read(b.ctypes.data, len(b), start + 1, limit, out.ctypes.data)
            if rc == -7:
                continue  # LZ77 in a damaged code: outside the slice (include/jxl_hip_frame.h)
            assert (rc == 0) == (want == 0), (trial, rc, want, out[:5])
            if rc == 0:
                agree_ok += 1
                assert pos.value == int(out[2]), (trial, pos.value, out[:3])
            else:
                agree_bad += 1
    assert agree_ok > 100 and agree_bad > 100, (agree_ok, agree_bad)
