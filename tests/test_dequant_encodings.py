"""a5 beyond the default library: custom dequant-matrix encodings
(DequantMatrices::Decode + ComputeQuantTable, lib/jxl/quant_weights.cc:163-511).
The REFERENCE's DequantMatricesEncode writes the bits, the product's parser
(jxlhip_dequant_encodings_decode) reads them, and the tables -- C restatement on
CPU, k_dequant_tables on the GPU -- must equal the reference's
DequantMatrices::EnsureComputed float for float.  Damaged / out-of-range
encodings must fail where the reference fails."""
import ctypes as C

import numpy as np
import pytest

from libjxl_amd import abi

BAD_STREAM, UNSUPPORTED = -5, -7
SINGLE = (0, 1, 2, 3, 9, 10)  # table kinds of one 8x8 block


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    oracle.ref_lib()
    return oracle


@pytest.fixture(scope="module")
def L():
    return abi.load_library()


def f16(rng, lo, hi, n):
    """n values in [lo, hi) exactly representable as IEEE half."""
    return rng.uniform(lo, hi, n).astype(np.float16).astype(np.float32)


def fill_bands(rng, nb, bands, seed_lo=2.0, seed_hi=60.0):
    for c in range(3):
        bands[c][0] = float(f16(rng, seed_lo, seed_hi, 1)[0]) * 64.0
        rest = f16(rng, -1.5, 1.5, nb - 1)
        for i in range(1, nb):
            bands[c][i] = float(rest[i - 1])


def random_encodings(rng, modes=None):
    """Every table gets a non-library mode valid for its size (DCT for the large ones)."""
    encs = abi.QuantEncodings()
    for k in range(abi.NUM_QUANT_TABLES):
        e = encs[k]
        if modes is not None:
            mode = modes[k]
        elif k in SINGLE:
            mode = int(rng.choice([abi.QUANT_ID, abi.QUANT_DCT2, abi.QUANT_DCT4, abi.QUANT_DCT4X8, abi.QUANT_AFV,
                                   abi.QUANT_DCT, abi.QUANT_LIBRARY]))
        else:
            mode = int(rng.choice([abi.QUANT_DCT, abi.QUANT_LIBRARY]))
        e.mode = mode
        if mode == abi.QUANT_ID:
            for c in range(3):
                w = f16(rng, 0.5, 60.0, 3) * 64.0
                for i in range(3):
                    e.weights[c][i] = float(w[i])
        elif mode == abi.QUANT_DCT2:
            for c in range(3):
                w = f16(rng, 0.25, 60.0, 6) * 64.0
                for i in range(6):
                    e.weights[c][i] = float(w[i])
        elif mode == abi.QUANT_DCT4:
            for c in range(3):
                w = f16(rng, 0.5, 2.0, 2)
                for i in range(2):
                    e.weights[c][i] = float(w[i])
            e.num_bands = int(rng.integers(1, 17))
            fill_bands(rng, e.num_bands, e.bands)
        elif mode == abi.QUANT_DCT4X8:
            for c in range(3):
                e.weights[c][0] = float(f16(rng, 0.5, 2.0, 1)[0])
            e.num_bands = int(rng.integers(1, 17))
            fill_bands(rng, e.num_bands, e.bands)
        elif mode == abi.QUANT_AFV:
            for c in range(3):
                w = f16(rng, 0.5, 50.0, 6) * 64.0
                for i in range(6):
                    e.weights[c][i] = float(w[i])
                r = f16(rng, -1.0, 1.0, 3)
                for i in range(3):
                    e.weights[c][6 + i] = float(r[i])
            e.num_bands = int(rng.integers(1, 17))
            fill_bands(rng, e.num_bands, e.bands)
            e.num_bands_afv_4x4 = int(rng.integers(1, 17))
            fill_bands(rng, e.num_bands_afv_4x4, e.bands_afv_4x4)
        elif mode == abi.QUANT_DCT:
            e.num_bands = int(rng.integers(1, 17))
            fill_bands(rng, e.num_bands, e.bands, 20.0, 400.0)
    return encs


def parse(L, data, bit_pos=0):
    d = np.frombuffer(data, np.uint8)
    encs = abi.QuantEncodings()
    pos = C.c_size_t(bit_pos)
    rc = L.jxlhip_dequant_encodings_decode(d.ctypes.data, len(d), C.byref(pos), C.byref(encs))
    return rc, encs, pos.value


def same(a, b):
    return bytes(a) == bytes(b)


@pytest.mark.parametrize("seed", range(8))
def test_parser_and_tables_match_reference(L, ref, seed):
    rng = np.random.default_rng(1000 + seed)
    encs = random_encodings(rng)
    data = ref.ref_dequant_encode(encs)
    rc, got, pos = parse(L, data)
    assert rc == 0
    assert same(got, encs)  # F16 parameters survive exactly, scalings included
    st, want, bits = ref.ref_dequant_decode(data)
    assert pos == bits
    mine = ref.dequant_tables(got)
    if st == 2:
        assert mine is None  # "Invalid quantization table" / "Invalid distance bands"
        return
    assert st == 0
    assert mine is not None and np.array_equal(mine.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode", [abi.QUANT_ID, abi.QUANT_DCT2, abi.QUANT_DCT4, abi.QUANT_DCT4X8, abi.QUANT_AFV,
                                  abi.QUANT_DCT])
def test_each_mode_on_every_8x8_table(L, ref, mode):
    rng = np.random.default_rng(50 + mode)
    modes = [mode if (k in SINGLE) else abi.QUANT_DCT for k in range(17)]
    encs = random_encodings(rng, modes)
    data = ref.ref_dequant_encode(encs)
    rc, got, _ = parse(L, data)
    assert rc == 0 and same(got, encs)
    st, want, _ = ref.ref_dequant_decode(data)
    assert st == 0
    assert np.array_equal(ref.dequant_tables(got).view(np.uint32), want.view(np.uint32))


def test_all_default_bit(L, ref):
    rc, encs, pos = parse(L, b"\x01")
    assert rc == 0 and pos == 1 and all(e.mode == abi.QUANT_LIBRARY for e in encs)
    assert np.array_equal(ref.dequant_tables(encs), ref.ref_default_dequant_tables())
    # library written explicitly: 1 + 17 * 3 bits
    data = ref.ref_dequant_encode(abi.QuantEncodings())
    assert data == b"\x01"  # the reference collapses an all-library set to the flag
    rc, encs, pos = parse(L, bytes(8))  # all_default = 0, then 17 x mode 0
    assert rc == 0 and pos == 1 + 17 * 3 and all(e.mode == abi.QUANT_LIBRARY for e in encs)
    assert ref.ref_dequant_decode(bytes(8))[2] == pos


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, nbits, value):
        self.v |= (int(value) & ((1 << nbits) - 1)) << self.n
        self.n += nbits

    def f16(self, x):
        self.put(16, int(np.float16(x).view(np.uint16)))

    def bytes(self, pad=0):
        return self.v.to_bytes((self.n + 7) // 8 + pad, "little")


def stream_with(kind, body):
    """all_default = 0, library for every table but `kind`, whose encoding body() writes."""
    b = Bits()
    b.put(1, 0)
    for k in range(17):
        if k == kind:
            body(b)
        else:
            b.put(3, abi.QUANT_LIBRARY)
    return b


def agree_on_failure(L, ref, data):
    rc, _, _ = parse(L, data)
    st, _, _ = ref.ref_dequant_decode(data)
    assert st == 1, st
    assert rc == BAD_STREAM, rc


def test_reference_decode_failures(L, ref):
    # ID on a 16x16 table
    agree_on_failure(L, ref, stream_with(4, lambda b: b.put(3, abi.QUANT_ID)).bytes(64))
    # DCT4 on 32x32
    agree_on_failure(L, ref, stream_with(5, lambda b: b.put(3, abi.QUANT_DCT4)).bytes(64))

    def tiny_id(b):
        b.put(3, abi.QUANT_ID)
        for i in range(9):
            b.f16(0.0 if i == 4 else 1.0)
    agree_on_failure(L, ref, stream_with(1, tiny_id).bytes(8))

    def nan_band(b):
        b.put(3, abi.QUANT_DCT)
        b.put(4, 1)
        b.put(16, 0x7E00)  # NaN
        for _ in range(5):
            b.f16(1.0)
    agree_on_failure(L, ref, stream_with(0, nan_band).bytes(8))

    def negative_seed(b):
        b.put(3, abi.QUANT_DCT)
        b.put(4, 0)
        b.f16(-3.0)
        b.f16(1.0)
        b.f16(1.0)
    agree_on_failure(L, ref, stream_with(6, negative_seed).bytes(8))

    def small_mul(b):
        b.put(3, abi.QUANT_DCT4X8)
        b.f16(0.0)
        b.f16(1.0)
        b.f16(1.0)
    agree_on_failure(L, ref, stream_with(9, small_mul).bytes(64))
    # truncated: a DCT encoding announcing 17 bands with no bytes behind it
    b = stream_with(16, lambda b: (b.put(3, abi.QUANT_DCT), b.put(4, 15)))
    agree_on_failure(L, ref, b.bytes(0))


def test_compute_failures_match_reference(L, ref):
    # band product underflows below 1e-8: Decode accepts, ComputeQuantTable rejects
    def collapsing(b):
        b.put(3, abi.QUANT_DCT)
        b.put(4, 15)
        for c in range(3):
            b.f16(1.0 / 64)
            for _ in range(15):
                b.f16(-60000.0)
    data = stream_with(0, collapsing).bytes(8)
    rc, encs, _ = parse(L, data)
    assert rc == 0
    assert ref.ref_dequant_decode(data)[0] == 2
    assert ref.dequant_tables(encs) is None

    # huge weights: 1/w < 1e-8 is fine, w >= 1e8 is not
    def huge(b):
        b.put(3, abi.QUANT_DCT)
        b.put(4, 3)
        for c in range(3):
            b.f16(60000.0)
            for _ in range(3):
                b.f16(60000.0)
    data = stream_with(4, huge).bytes(8)
    rc, encs, _ = parse(L, data)
    assert rc == 0
    st = ref.ref_dequant_decode(data)[0]
    assert (ref.dequant_tables(encs) is None) == (st == 2)
    assert st == 2


def test_raw_mode_is_reported_unsupported(L):
    rc, _, _ = parse(L, stream_with(0, lambda b: b.put(3, abi.QUANT_RAW)).bytes(64))
    assert rc == UNSUPPORTED


def test_ac_global_decode_whole_section(L, ref):
    """FrameDecoder::ProcessACGlobal in one call, on a genuine stream's AC-global section."""
    rs = ref.RealStream(520, 300, seed=820, distance=3.0, speed_tier=3)
    bctx = abi.BlockCtxMap()
    pos = C.c_size_t(0)
    b = rs.block_ctx_bytes
    assert L.jxlhip_block_ctx_map_decode(b.ctypes.data, len(b), C.byref(pos), C.byref(bctx)) == 0
    glob = np.frombuffer(rs.ac_global(), np.uint8)
    encs = abi.QuantEncodings()
    nh, used, h = C.c_uint32(0), C.c_size_t(0), (C.c_void_p * 1)()
    rc = L.jxlhip_ac_global_decode(glob.ctypes.data, len(glob), rs.num_groups, 1, rs.used_acs, C.byref(bctx),
                                   C.byref(encs), C.byref(nh), h, C.byref(used))
    assert rc == 0
    assert nh.value == rs.num_histograms and (used.value + 7) // 8 == len(glob)
    assert all(e.mode == abi.QUANT_LIBRARY for e in encs)
    assert np.array_equal(ref.dequant_tables(encs), rs.dequant_table)
    L.jxlhip_ac_pass_destroy(h[0])
    # truncated section: no handles leak out
    rc = L.jxlhip_ac_global_decode(glob.ctypes.data, len(glob) // 2, rs.num_groups, 1, rs.used_acs, C.byref(bctx),
                                   C.byref(encs), C.byref(nh), h, C.byref(used))
    assert rc == BAD_STREAM and not h[0]


def test_oracle_frame_with_custom_tables_matches_reference(ref):
    """The C restatement's full decode under custom tables == the reference's, bit for bit."""
    import frames
    from libjxl_amd import synth
    rng = np.random.default_rng(7)
    while True:
        encs = random_encodings(rng)
        table = ref.dequant_tables(encs)
        if table is not None:
            break
    params, t, _ = frames.make_case(264, 200, mix=synth.MIX_ALL, gab=True, epf_iters=2, seed=5)
    npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"], table)
    want = fr.decode_ref(threads=4, quant_encodings=encs)
    got = fr.decode(threads=4)
    assert np.array_equal(got, want)
    assert not np.array_equal(want, fr.decode_ref(threads=4))  # and the tables do matter


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_device_tables_match_reference(ref, seed):
    import torch

    from libjxl_amd import VarDctDecoder
    rng = np.random.default_rng(1000 + seed)
    encs = random_encodings(rng)
    st, want, _ = ref.ref_dequant_decode(ref.ref_dequant_encode(encs))
    d = VarDctDecoder(0)
    t = d.dequant_tables(encs)
    if st == 2:
        with pytest.raises(abi.JxlHipError):
            d.sync()
    else:
        d.sync()
        assert np.array_equal(t.cpu().numpy().view(np.uint32), want.view(np.uint32))
    d.close()


@pytest.mark.gpu
def test_frame_with_custom_tables_matches_reference(ref):
    """A full decode under custom dequant matrices, HIP path vs the reference's pixels."""
    import os

    import frames
    from libjxl_amd import VarDctDecoder, synth
    rng = np.random.default_rng(7)
    while True:
        encs = random_encodings(rng)
        st, want_t, _ = ref.ref_dequant_decode(ref.ref_dequant_encode(encs))
        if st == 0:
            break
    d = VarDctDecoder(0)
    params, t = synth.synth_frame(520, 392, device="cuda", mix=synth.MIX_ALL, gab=True, epf_iters=2, seed=5)
    d.begin_frame(params)
    dq = d.dequant_tables(encs)
    d.set_inputs(t, dq)
    got = d.decode_frame()
    d.sync()
    got = got.cpu().numpy()
    npy = {k: ([x.cpu().numpy() for x in v] if isinstance(v, list) else v.cpu().numpy()) for k, v in t.items()}
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"], want_t)
    want = fr.decode_ref(threads=min(32, os.cpu_count() or 1), quant_encodings=encs)
    d.close()
    finite = np.isfinite(want)
    assert finite.all()
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got - want).max()) / scale <= 2e-5
