"""GPU parity against the libjxl REFERENCE ITSELF (oracle/_ref/libjxl_ref.so:
lib/jxl's decoder sources compiled in place, driven through
DecodeGroupForRoundtrip + the real render pipeline; prebuilt in the build
container, it travels to the GPU box).  The HIP path is called through the C
ABI.  Includes BASELINE.json's configs at their FULL sizes (4K filters-off, 8K
full pipeline), compared pixel for pixel.

Tolerance: the reference holds its own two executors to 2e-4 relative
(lib/jxl/render_pipeline/render_pipeline_test.cc:321-327); the kernels keep the
reference's operation order, the only deviations being v_rcp_f32 in
AdjustQuantBias and in the EPF normalisation, so the assertion is 2e-5 of the
output range."""
import os

import numpy as np
import pytest
import torch

import frames
from libjxl_amd import VarDctDecoder, abi, synth

pytestmark = pytest.mark.gpu

TIGHT = 2e-5
THREADS = min(64, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.fail("oracle/_ref/libjxl_ref.so missing: run __graft_entry__.build() in the build container")
    oracle.ref_lib()
    return oracle


@pytest.fixture(scope="module")
def dec():
    d = VarDctDecoder(0)
    yield d
    d.close()


def run_case(dec, ref, xs, ys, **kw):
    params, t = synth.synth_frame(xs, ys, device="cuda", **kw)
    dec.begin_frame(params)
    dq = dec.default_dequant_tables()
    dec.set_inputs(t, dq)
    out = dec.decode_frame()
    dec.sync()
    got = out.cpu().numpy()
    npy = {k: ([x.cpu().numpy() for x in v] if isinstance(v, list) else v.cpu().numpy()) for k, v in t.items()}
    # the reference decodes with ITS OWN dequant tables (DequantMatrices::EnsureComputed), not with the product's: a
    # wrong table on the device cannot cancel out (test_dequant_tables_bit_identical_to_reference pins the two besides)
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"],
                   ref.ref_default_dequant_tables())
    want = fr.decode_ref(threads=THREADS)
    assert got.shape == want.shape
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err <= TIGHT, err
    return err


@pytest.mark.parametrize("gab,epf", [(g, e) for g in (False, True) for e in (0, 1, 2, 3)])
def test_all_strategies_every_stage_list(dec, ref, gab, epf):
    run_case(dec, ref, 533, 401, mix=synth.MIX_ALL, gab=gab, epf_iters=epf, seed=40 + epf)


def test_dequant_tables_bit_identical_to_reference(dec, ref):
    params, _ = synth.synth_frame(8, 8, mix=synth.MIX_DCT8)
    dec.begin_frame(params)
    got = dec.default_dequant_tables()
    dec.sync()
    want = ref.ref_default_dequant_tables()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_config0_1024_d1_full_pipeline(dec, ref):
    run_case(dec, ref, 1024, 1024, mix=synth.MIX_D1, gab=True, epf_iters=1)


def test_config1_4k_filters_off_full_size(dec, ref):
    """BASELINE configs[1]: 3840x2160 d1.0, IDCT + XYB only."""
    run_case(dec, ref, 3840, 2160, mix=synth.MIX_D1, gab=False, epf_iters=0)


def test_config2_8k_full_pipeline_full_size(dec, ref):
    """BASELINE configs[2] (the bench workload): 7680x4320 d1.0, Gaborish + EPF1."""
    run_case(dec, ref, 7680, 4320, mix=synth.MIX_D1, gab=True, epf_iters=1)


def test_config4_hdr_dct32_int32(dec, ref):
    """BASELINE configs[4] at reduced size: every block DCT32X32, int32
    coefficients, d0.5-like quantisation, intensity_target 4000."""
    run_case(dec, ref, 2048, 1024, mix=synth.MIX_DCT32, gab=False, epf_iters=0, coeff_type=1,
             intensity_target=4000.0, quant_mul=2.0)


@pytest.mark.parametrize("mfma", ["1", "0"])
def test_config4_8k_full_size(ref, mfma, monkeypatch):
    """BASELINE configs[4] at its FULL size, as bench.py --config c5 runs it: 7680x4320, every block DCT32X32,
    int32 coefficients, d0.5-like quantisation, intensity_target 1000 -- through the matrix-core IDCT
    (kernels_mfma.hip, what the context picks for this frame) and through the row-per-lane butterflies."""
    monkeypatch.setenv("JXLHIP_MFMA", mfma)
    d = VarDctDecoder(0)
    try:
        run_case(d, ref, 7680, 4320, mix=synth.MIX_DCT32, gab=False, epf_iters=0, coeff_type=1,
                 intensity_target=1000.0, quant_mul=2.0)
    finally:
        d.close()


@pytest.mark.parametrize("coeff_type", [0, 1])
def test_dct32_only_frame_written_by_the_class_kernel(ref, coeff_type):
    """All-DCT32X32, no loop filter, float RGB: k_transform_mfma32<EMIT> writes the pixels itself (no planes, no
    filter kernel).  A size whose last block column / row is clipped by the image (1020 x 508 = 128 x 64 blocks)."""
    d = VarDctDecoder(0)
    try:
        kw = dict(coeff_type=1, quant_mul=2.0) if coeff_type else {}
        run_case(d, ref, 1020, 508, mix=synth.MIX_DCT32, gab=False, epf_iters=0, intensity_target=1000.0, seed=32, **kw)
    finally:
        d.close()


def test_config3_16k_striped_below_the_abi_full_size(ref):
    """BASELINE configs[3] at its FULL size (15360x8640 d1.0, Gaborish + EPF1), decoded the way --gpus N decodes it:
    jxlhip_create_multi splits the frame into stripes of AC-group rows (one per visible device; on a 1-GPU box
    four stripes share device 0), halo rows and the gather are peer copies below the C ABI.  Against the
    reference's own decode of the whole frame, pixel for pixel."""
    import ctypes as C
    xs, ys = 15360, 8640
    params, t = synth.synth_frame(xs, ys, device="cpu", mix=synth.MIX_D1, gab=True, epf_iters=1, seed=16)
    L = abi.load_library()
    one = VarDctDecoder(0)
    one.begin_frame(params)
    table_host = one.default_dequant_tables().cpu().numpy()
    one.sync()
    one.close()
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(max(4, ndev))]
    ctx = C.c_void_p()
    assert L.jxlhip_create_multi((C.c_int * len(devices))(*devices), len(devices), None, C.byref(ctx)) == 0
    try:
        p = abi.make_params(params)
        assert L.jxlhip_frame_begin(ctx, C.byref(p)) == 0, L.jxlhip_last_error(ctx)
        npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
        dc3 = (C.c_void_p * 3)(*[a.ctypes.data for a in npy["dc"]])
        assert L.jxlhip_upload_side_info(ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                         npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                         npy["ytob_map"].ctypes.data, dc3, table_host.ctypes.data) == 0
        for g in range(((xs + 255) // 256) * ((ys + 255) // 256)):
            ptrs = (C.c_void_p * 3)(*[c[g * 65536:].ctypes.data for c in npy["coeffs"]])
            assert L.jxlhip_submit_group(ctx, g, ptrs, 65536) == 0, L.jxlhip_last_error(ctx)
        got = np.zeros((ys, xs, 3), np.float32)
        assert L.jxlhip_decode_frame_host(ctx, got.ctypes.data, xs * 12, 0) == 0, L.jxlhip_last_error(ctx)
    finally:
        L.jxlhip_destroy(ctx)
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"], table_host)
    want = fr.decode_ref(threads=THREADS)
    scale = max(1.0, float(np.abs(want).max()))
    # row blocks: the difference of two 1.6 GB arrays in one piece is another 1.6 GB
    err = max(float(np.abs(got[y:y + 540] - want[y:y + 540]).max()) for y in range(0, ys, 540)) / scale
    assert err <= TIGHT, err


def test_xyb_planar_output(dec, ref):
    run_case(dec, ref, 600, 300, mix=synth.MIX_ALL, gab=True, epf_iters=2, output_kind=0)


# ---- f3: colour-encoding + packing stages fused into the last kernel ----------
def run_packed(dec, ref, xs, ys, fmt, **kw):
    params, t = synth.synth_frame(xs, ys, device="cuda", output_kind=2, out_format=fmt, **kw)
    dec.begin_frame(params)
    dq = dec.default_dequant_tables()
    dec.set_inputs(t, dq)
    out = dec.decode_frame()
    dec.sync()
    got = out.cpu().numpy()
    npy = {k: ([x.cpu().numpy() for x in v] if isinstance(v, list) else v.cpu().numpy()) for k, v in t.items()}
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"],
                   dq.cpu().numpy())
    want = fr.decode_ref(threads=THREADS)
    assert got.shape == want.shape
    if os.environ.get("JXLHIP_TEST_ARBITRATE"):  # debugging aid: who is off, the kernels or the reference run?
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        if d.max() > 1:
            mine = fr.decode(threads=8)
            want1 = fr.decode_ref(threads=1)
            out2 = dec.decode_frame()
            dec.sync()
            got2 = out2.cpu().numpy()
            ys, xs = np.nonzero(d.max(axis=2) > 1)
            y, x = int(ys[0]), int(xs[0])
            print("ARBITRATE at", y, x, "got", got[y, x - 2:x + 3].tolist(), "want", want[y, x - 2:x + 3].tolist(),
                  "oracle", mine[y, x - 2:x + 3].tolist(), "ref1", want1[y, x - 2:x + 3].tolist(),
                  "gpu again", got2[y, x - 2:x + 3].tolist(), "| ref==ref1", np.array_equal(want, want1),
                  "oracle==ref1", np.array_equal(mine, want1), "gpu==gpu2", np.array_equal(got, got2))
    return got, want


@pytest.mark.parametrize("gab,epf", [(True, 1), (False, 0), (True, 3), (True, 2), (False, 2)])
@pytest.mark.parametrize("nc", [3, 4])
def test_packed_srgb_u8(dec, ref, gab, epf, nc):
    """What djxl writes by default (8-bit sRGB): the float pipeline differs from the
    reference by <= 2e-5 of the range, so after x255 + dither + rounding a sample may land on
    the other side of a rounding boundary: at most 1 LSB, for at most 0.1 % of the samples."""
    got, want = run_packed(dec, ref, 533, 401, dict(transfer=1, sample_type=1, num_channels=nc, bits_per_sample=8),
                           mix=synth.MIX_ALL, gab=gab, epf_iters=epf, intensity_target=80.0)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    if d.max() > 1:  # say which side is off before failing
        ys, xs, cs = np.nonzero(d > 1)
        print("samples off by > 1:", len(ys), "rows", ys.min(), "-", ys.max(), "cols", xs.min(), "-", xs.max(),
              "channels", sorted(set(cs.tolist())), "got", got[ys[0], xs[0]], "want", want[ys[0], xs[0]])
    assert d.max() <= 1
    assert (d != 0).mean() < 1e-3
    if nc == 4:
        assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("tf,st,bits,nc,sw", [(1, 2, 16, 3, 0), (1, 2, 12, 4, 1), (0, 2, 16, 3, 0)])
def test_packed_u16(dec, ref, tf, st, bits, nc, sw):
    got, want = run_packed(dec, ref, 520, 264, dict(transfer=tf, sample_type=st, num_channels=nc,
                                                    bits_per_sample=bits, swap_endianness=sw),
                           mix=synth.MIX_D1, gab=True, epf_iters=1, intensity_target=80.0 if tf else 255.0)
    g, w = got.view(np.uint16), want.view(np.uint16)
    if sw:
        g, w = g.byteswap(), w.byteswap()
    d = np.abs(g.astype(np.int32) - w.astype(np.int32))
    assert d.max() <= max(2, int(TIGHT * 8 * (1 << bits)))


@pytest.mark.parametrize("st,sw", [(3, 0), (3, 1), (0, 0), (0, 1)])
def test_packed_float_srgb(dec, ref, st, sw):
    got, want = run_packed(dec, ref, 520, 264, dict(transfer=1, sample_type=st, num_channels=4, swap_endianness=sw),
                           mix=synth.MIX_D1, gab=True, epf_iters=1, intensity_target=80.0)
    if st == 3:
        g, w = got.view(np.uint16), want.view(np.uint16)
        if sw:
            g, w = g.byteswap(), w.byteswap()
        g, w = g.view(np.float16).astype(np.float32), w.view(np.float16).astype(np.float32)
        tol = 2e-3  # one f16 ulp at 1.0 is 9.8e-4
    else:
        g, w = got.view(np.uint32), want.view(np.uint32)
        if sw:
            g, w = g.byteswap(), w.byteswap()
        g, w = g.view(np.float32), w.view(np.float32)
        tol = 1e-4  # the sRGB curve amplifies dark-end differences (slope 12.92)
    assert float(np.abs(g - w).max()) <= tol * max(1.0, float(np.abs(w).max()))
    assert (g[..., 3] == 1.0).all()


@pytest.mark.parametrize("gab,epf", [(True, 1), (False, 0), (True, 2), (False, 1), (True, 3)])
@pytest.mark.parametrize("tf,st,bits,nc,sw", [(1, 2, 16, 4, 0), (1, 2, 16, 3, 1), (1, 2, 10, 4, 1), (1, 0, 0, 3, 0), (1, 0, 0, 4, 0),
                                              (0, 0, 0, 4, 0), (1, 3, 0, 4, 0), (0, 3, 0, 4, 0), (2, 2, 16, 3, 1), (2, 2, 16, 4, 1)])
def test_packed_formats_with_a_kernel_of_their_own(dec, ref, tf, st, bits, nc, sw, gab, epf):
    """Round 3's fixed-format instantiations of the row march (kernels_filters_fast_{b,c,d}.hip: 16-bit sRGB RGBA and
    the big-endian 16-bit forms, float sRGB / linear, half-float RGBA, big-endian 16-bit PQ) against the reference's
    FromLinearStage + WriteToOutputStage, over the stage lists."""
    hdr = tf == 2
    got, want = run_packed(dec, ref, 520, 264, dict(transfer=tf, sample_type=st, num_channels=nc, bits_per_sample=bits,
                                                    swap_endianness=sw, tf_param=1000.0 if hdr else 0.0),
                           mix=synth.MIX_D1, gab=gab, epf_iters=epf, intensity_target=1000.0 if hdr else (80.0 if tf else 255.0))
    if st == 2:
        g, w = got.view(np.uint16), want.view(np.uint16)
        if sw:
            g, w = g.byteswap(), w.byteswap()
        d = np.abs(g.astype(np.int32) - w.astype(np.int32))
        if hdr:  # PQ's slope near zero (~1e3 at 1e-4) amplifies the float pipeline's 2e-5
            assert d[..., :3].max() <= 140 and (d[..., :3] > 8).mean() < 2e-3
        else:
            assert d.max() <= max(2, int(TIGHT * 8 * (1 << bits)))
        if nc == 4:
            assert (g[..., 3] == (1 << bits) - 1).all()
        return
    if st == 3:
        g, w = got.view(np.uint16).view(np.float16).astype(np.float32), want.view(np.uint16).view(np.float16).astype(np.float32)
        tol = 2e-3
    else:
        g, w = got.view(np.float32), want.view(np.float32)
        tol = 1e-4 if tf else TIGHT
    assert float(np.abs(g - w).max()) <= tol * max(1.0, float(np.abs(w).max()))
    if nc == 4:
        assert (g[..., 3] == 1.0).all()


@pytest.mark.parametrize("tf,par,it", [(2, 1000.0, 1000.0), (3, 0.0, 255.0), (4, 1 / 2.6, 255.0), (5, 1000.0, 1000.0),
                                       (5, 334.0, 334.0)])
@pytest.mark.parametrize("st,bits", [(1, 8), (2, 16), (0, 0)])
def test_packed_pq_709_gamma(dec, ref, tf, par, it, st, bits):
    """HDR / video transfer functions (PQ at 1000 nits, BT.709, DCI gamma) through the
    kernels' general packed path."""
    got, want = run_packed(dec, ref, 520, 264, dict(transfer=tf, sample_type=st, num_channels=3,
                                                    bits_per_sample=bits, tf_param=par),
                           mix=synth.MIX_D1, gab=True, epf_iters=1, intensity_target=it)
    if st == 0:
        # steep curves near zero (PQ: slope ~1e3 at 1e-4) amplify the float pipeline's 2e-5
        assert float(np.abs(got - want).max()) <= 2e-3 * max(1.0, float(np.abs(want).max()))
        assert float(np.abs(got - want).mean()) <= 2e-5
    else:
        if st == 2:
            got, want = got.view(np.uint16), want.view(np.uint16)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= (1 if st == 1 else 140)
        assert (d > (0 if st == 1 else 8)).mean() < 2e-3


def test_packed_8k_srgb_u8_rgba_full_size(dec, ref):
    """The bench workload with djxl's default output: 7680x4320 d1.0 -> sRGB RGBA8."""
    got, want = run_packed(dec, ref, 7680, 4320, dict(transfer=1, sample_type=1, num_channels=4, bits_per_sample=8),
                           mix=synth.MIX_D1, gab=True, epf_iters=1, intensity_target=80.0)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


# ---- f1: bytes -> pixels.  The reference's entropy ENCODER writes the AC streams;
# the product decodes them on host threads into pinned staging, uploads them group
# by group and renders; the result must equal the decode from device-resident
# coefficients bit for bit (entropy coding is lossless).
@pytest.mark.parametrize("sparse", ["1", "0"])
def test_entropy_decode_submit_end_to_end(dec, ref, sparse, monkeypatch):
    """sparse = 1 (the default): a group crosses PCIe as its non-zero coefficients and k_expand_sparse rebuilds the
    dense block stream on the device; 0: the dense staging slot.  Either way the pixels of the device-resident path,
    bit for bit -- and a later frame that is handed over DENSELY (jxlhip_submit_group) in the same context must not
    see the earlier frame's sparse groups."""
    import ctypes as C
    import threading
    monkeypatch.setenv("JXLHIP_SPARSE_UPLOAD", sparse)
    xs, ys = 1000, 700
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=77)
    dq = dec.default_dequant_tables()
    dec.begin_frame(params)
    dec.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}, dq)
    want = dec.decode_frame().clone()
    dec.sync()

    glob, groups, used_acs, _ = fr.encode_ac_ref(histo_sets=2)
    d2 = VarDctDecoder(0)
    d2.begin_frame(params)
    L = d2.L
    npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
    dqh = dq.cpu().numpy()
    dc3 = (C.c_void_p * 3)(*[x.ctypes.data for x in npy["dc"]])
    assert L.jxlhip_upload_side_info(d2.ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                     npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                     npy["ytob_map"].ctypes.data, dc3, dqh.ctypes.data) == 0
    g = np.frombuffer(glob, np.uint8)
    pos, h = C.c_size_t(0), C.c_void_p()
    assert L.jxlhip_ac_pass_decode(g.ctypes.data, len(g), C.byref(pos), used_acs, 2, None, C.byref(h)) == 0
    assert L.jxlhip_ac_pass_max_num_bits(h) < 16  # int16 coefficients, as the frame was set up
    ng = len(groups)
    errs = []

    def worker(tid, nthreads):  # the JxlParallelRunner's role: groups in any order, concurrently
        for gi in range(tid, ng, nthreads):
            d = np.frombuffer(groups[gi], np.uint8)
            gp = C.c_size_t(0)
            rc = L.jxlhip_ac_group_decode_submit(d2.ctx, h, gi, npy["ac_strategy"].ctypes.data,
                                                 npy["raw_quant"].ctypes.data, None, d.ctypes.data, len(d),
                                                 C.byref(gp))
            if rc != 0:
                errs.append((gi, rc))

    threads = [threading.Thread(target=worker, args=(i, 12)) for i in range(12)]  # more threads than staging slots
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    got = d2.decode_frame()
    d2.sync()
    L.jxlhip_ac_pass_destroy(h)
    assert torch.equal(got, want)
    assert torch.equal(d2.decode_frame(), want)  # a second decode of the same frame
    d2.sync()
    # another frame, same context, dense hand-over of (other) coefficients
    params2, t2, _ = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=78)
    dec.begin_frame(params2)
    dec.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t2.items()}, dq)
    want2 = dec.decode_frame().clone()
    dec.sync()
    assert not torch.equal(want2, want)
    d2.begin_frame(params2)
    npy2 = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t2.items()}
    dc3b = (C.c_void_p * 3)(*[x.ctypes.data for x in npy2["dc"]])
    assert L.jxlhip_upload_side_info(d2.ctx, npy2["ac_strategy"].ctypes.data, npy2["raw_quant"].ctypes.data,
                                     npy2["epf_sharpness"].ctypes.data, npy2["ytox_map"].ctypes.data,
                                     npy2["ytob_map"].ctypes.data, dc3b, dqh.ctypes.data) == 0
    for gi in range(ng):
        ptrs = (C.c_void_p * 3)(*[npy2["coeffs"][c][gi * 65536:].ctypes.data for c in range(3)])
        assert L.jxlhip_submit_group(d2.ctx, gi, ptrs, 65536) == 0
    got2 = d2.decode_frame()
    d2.sync()
    assert torch.equal(got2, want2)
    d2.close()


# ---- f3: undo_orientation in the write stage ----------------------------------
@pytest.mark.parametrize("orientation", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("kind", ["f32", "u8", "u16"])
def test_undo_orientation_matches_the_reference_write_stage(ref, orientation, kind):
    """jxlhip_frame_params::undo_orientation: the frame in DISPLAY orientation like WriteToOutputStage writes it
    (stage_write.cc:441-457 flips / transpose, and the 8-bit dither pattern at the FLIPPED coordinates,
    :486-492) -- every EXIF orientation, float and packed outputs, a ragged size."""
    xs, ys = 333, 212
    if kind == "f32":
        kw = dict(output_kind=1)
    else:
        fmt = dict(transfer=1, sample_type=1 if kind == "u8" else 2, num_channels=4 if kind == "u8" else 3,
                   bits_per_sample=8 if kind == "u8" else 16, swap_endianness=0, tf_param=0.0, luminances=[0, 0, 0])
        kw = dict(output_kind=2, out_format=fmt)
    params, t = synth.synth_frame(xs, ys, device="cuda", mix=synth.MIX_D1, gab=True, epf_iters=1, seed=70 + orientation,
                                  undo_orientation=orientation, **kw)
    d = VarDctDecoder(0)
    try:
        d.begin_frame(params)
        dq = d.default_dequant_tables()
        d.set_inputs(t, dq)
        oh, ow = (xs, ys) if orientation >= 5 else (ys, xs)
        got = d.decode_frame().cpu().numpy()
        d.sync()
    finally:
        d.close()
    npy = {k: ([x.cpu().numpy() for x in v] if isinstance(v, list) else v.cpu().numpy()) for k, v in t.items()}
    fr = ref.Frame(frames.to_oracle_params(abi.make_params(params)), npy["coeffs"], npy["ac_strategy"],
                   npy["raw_quant"], npy["epf_sharpness"], npy["ytox_map"], npy["ytob_map"], npy["dc"],
                   dq.cpu().numpy())
    want = fr.decode_ref(threads=1)
    assert got.shape == want.shape == (oh, ow, want.shape[2])
    if kind == "f32":
        assert float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max())) <= TIGHT
    else:
        diff = np.abs(got.view(want.dtype).astype(np.int64) - want.astype(np.int64))
        # the float pipeline in front differs by <= 2e-5 of the range: a sample may land on the other side of a
        # rounding boundary -- 1 of 255 codes for a few samples in a thousand, 1-2 of 65535 codes more often
        if kind == "u8":
            assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())
        else:
            assert diff.max() <= 2 and (diff > 0).mean() < 2e-2, (diff.max(), (diff > 0).mean())
