"""f4, first piece: the frame table of contents (toc.cc ReadGroupOffsets) -- where the sections of
a frame are.  TOCs written by the reference (enc_toc.cc), with and without a permutation, and the
TOC of genuine codestreams; damaged TOCs must fail like the reference.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from libjxl_amd import abi


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    return oracle.ref_lib()


@pytest.fixture(scope="module")
def L():
    return abi.load_library()


def decode(L, data, n, bit_pos=0):
    d = np.frombuffer(data, np.uint8)
    off, sz = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    pos, total = C.c_size_t(bit_pos), C.c_uint64(0)
    rc = L.jxlhip_toc_decode(d.ctypes.data, len(d), C.byref(pos), n, off.ctypes.data, sz.ctypes.data, C.byref(total))
    return rc, off, sz, pos.value, total.value


def ref_write(R, sizes, perm=None):
    R.jxr_toc_write.restype = C.c_int64
    R.jxr_toc_write.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
    s = np.ascontiguousarray(sizes, np.uint32)
    p = None if perm is None else np.ascontiguousarray(perm, np.uint32)
    buf = np.zeros(1 << 20, np.uint8)
    n = R.jxr_toc_write(s.ctypes.data, len(s), None if p is None else p.ctypes.data, buf.ctypes.data, len(buf))
    assert n > 0
    return bytes(buf[:n])


def ref_read(R, data, n):
    R.jxr_toc_read.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    d = np.frombuffer(data, np.uint8)
    off, sz, bits = np.zeros(n, np.uint64), np.zeros(n, np.uint32), C.c_size_t(0)
    rc = R.jxr_toc_read(d.ctypes.data, len(d), n, off.ctypes.data, sz.ctypes.data, C.byref(bits))
    return rc, off, sz, bits.value


@pytest.mark.parametrize("n,permuted", [(1, False), (7, False), (7, True), (138, True), (2000, True), (65536, False)])
def test_toc_written_by_the_reference(L, ref, n, permuted):
    rng = np.random.default_rng(n)
    # every U32 bucket of kTocDist: < 1024, < 17408, < 4211712, up to 2^30
    sizes = np.concatenate([rng.integers(0, 1024, n // 2 + 1), rng.integers(0, 1 << 22, n)])[:n]
    if n > 3:
        sizes[1], sizes[2], sizes[3] = 17408 + 5, 4211712 + 9, (1 << 30) + 4211711
    perm = rng.permutation(n) if permuted else None
    data = ref_write(ref, sizes, perm)
    rc, off, sz, pos, total = decode(L, data + b"\0" * 8, n)
    assert rc == 0
    want_rc, want_off, want_sz, want_bits = ref_read(ref, data + b"\0" * 8, n)
    assert want_rc == 0 and pos == want_bits == len(data) * 8
    assert np.array_equal(off, want_off) and np.array_equal(sz, want_sz)
    assert total == int(sizes.astype(np.uint64).sum())
    if not permuted:
        assert np.array_equal(sz, sizes)


def test_toc_of_genuine_codestreams(L, oracle, ref):
    for kw in (dict(xsize=520, ysize=300), dict(xsize=776, ysize=520, progressive=1),
               dict(xsize=200, ysize=136)):  # the last one is a single-section frame
        rs = oracle.RealStream(seed=9, distance=1.5, speed_tier=3, **kw)
        n = L.jxlhip_num_toc_entries(rs.num_groups, rs.num_dc_groups, rs.num_passes)
        assert n == len(rs.section_size)
        frame = rs.codestream[rs.frame_offset:].tobytes()
        rc, off, sz, pos, total = decode(L, frame, n, rs.toc_bit_offset)
        assert rc == 0
        assert pos == (rs.sections_offset - rs.frame_offset) * 8
        assert np.array_equal(sz, rs.section_size.astype(np.uint32))
        assert np.array_equal(off + np.uint64(rs.sections_offset), rs.section_offset)
        assert rs.frame_offset + pos // 8 + total == len(rs.codestream)


def test_damaged_tocs_fail_like_the_reference(L, ref):
    rng = np.random.default_rng(5)
    n = 40
    data = bytearray(ref_write(ref, rng.integers(0, 5000, n), rng.permutation(n)))
    agree = 0
    for trial in range(300):
        d = bytearray(data)
        for _ in range(1 + trial % 3):
            d[rng.integers(0, len(d))] ^= 1 << rng.integers(0, 8)
        if trial % 7 == 0:
            d = d[:rng.integers(1, len(d))]
        b = bytes(d) + b"\0" * 4
        rc, off, sz, pos, _ = decode(L, b, n)
        want_rc, want_off, want_sz, want_bits = ref_read(ref, b, n)
        assert (rc == 0) == (want_rc == 0), trial
        if rc == 0:
            agree += 1
            assert pos == want_bits and np.array_equal(off, want_off) and np.array_equal(sz, want_sz)
    assert 0 < agree < 300
    assert decode(L, bytes(data), 65537)[0] != 0
