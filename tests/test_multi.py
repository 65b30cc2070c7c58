"""jxlhip_create_multi (include/jxl_hip.h): one context over several devices, stripes of AC-group rows, halo rows
and the final gather as stream-ordered peer copies below the C ABI.  A 1-GPU box lists device 0 several times
(several stripes on one GPU: same code path, peer copies degenerate to same-device copies); with >= 2 devices
visible the same test spreads the stripes over them."""
import ctypes as C

import numpy as np
import pytest
import torch

import frames
from libjxl_amd import VarDctDecoder, abi, synth

pytestmark = pytest.mark.gpu
TIGHT = 2e-5


def rel_err(got, ref):
    return float(np.abs(got.astype(np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))


def run_multi(L, devices, params, t, table_host, host_out, monkeypatch=None):
    ctx = C.c_void_p()
    devs = (C.c_int * len(devices))(*devices)
    assert L.jxlhip_create_multi(devs, len(devices), None, C.byref(ctx)) == 0
    try:
        p = abi.make_params(params)
        assert L.jxlhip_frame_begin(ctx, C.byref(p)) == 0, L.jxlhip_last_error(ctx)
        npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
        dc3 = (C.c_void_p * 3)(*[a.ctypes.data for a in npy["dc"]])
        assert L.jxlhip_upload_side_info(ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                         npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                         npy["ytob_map"].ctypes.data, dc3, table_host.ctypes.data) == 0
        xs, ys = params["xsize"], params["ysize"]
        ng = ((xs + 255) // 256) * ((ys + 255) // 256)
        for g in range(ng):
            ptrs = (C.c_void_p * 3)(*[c[g * 65536:].ctypes.data for c in npy["coeffs"]])
            assert L.jxlhip_submit_group(ctx, g, ptrs, 65536) == 0, L.jxlhip_last_error(ctx)
        if host_out:
            out = np.zeros((ys, xs, 3), np.float32)
            assert L.jxlhip_decode_frame_host(ctx, out.ctypes.data, xs * 12, 0) == 0, L.jxlhip_last_error(ctx)
            return out
        dev_out = torch.empty((ys, xs, 3), dtype=torch.float32, device=f"cuda:{devices[0]}")
        assert L.jxlhip_decode_frame(ctx, dev_out.data_ptr(), xs * 12, 0) == 0, L.jxlhip_last_error(ctx)
        assert L.jxlhip_sync(ctx) == 0, L.jxlhip_last_error(ctx)
        return dev_out.cpu().numpy()
    finally:
        L.jxlhip_destroy(ctx)


@pytest.mark.parametrize("nstripes,gather,host_out", [(2, False, False), (3, True, False), (2, False, True), (4, True, True)])
@pytest.mark.parametrize("gab,epf", [(1, 1), (1, 2), (0, 0)])
def test_multi_context_matches_single_context(oracle, monkeypatch, nstripes, gather, host_out, gab, epf):
    L = abi.load_library()
    xs, ys = 600, 1100  # 5 group rows
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=91)
    ref = fr.decode(threads=4)
    # the dequant table and the single-context result (two-phase: what the stripes run)
    monkeypatch.setenv("JXLHIP_FUSE", "0")
    d = VarDctDecoder(0)
    d.begin_frame(params)
    dq = d.default_dequant_tables()
    d.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}, dq)
    single = d.decode_frame().cpu().numpy()
    d.sync()
    table_host = dq.cpu().numpy()
    d.close()
    if gather:
        monkeypatch.setenv("JXLHIP_MULTI_FORCE_GATHER", "1")
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(nstripes)]
    got = run_multi(L, devices, params, t, table_host, host_out)
    assert rel_err(got, ref) <= TIGHT
    assert np.array_equal(got, single)  # stripes + halo exchange reproduce the whole-frame two-phase result bit for bit
    # interior rows first (the default: they are filtered while the halo rows travel) or everything behind the exchange
    # (round 3's order): the same pixels
    monkeypatch.setenv("JXLHIP_MULTI_INTERIOR_FIRST", "0")
    abi.load_library().jxlhip_debug_reload_env()
    assert np.array_equal(run_multi(L, devices, params, t, table_host, host_out), single)


def test_multi_context_refuses_what_it_does_not_do():
    L = abi.load_library()
    ctx = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert L.jxlhip_create_multi(devs, 2, None, C.byref(ctx)) == 0
    try:
        assert L.jxlhip_decode_blocks(ctx) == -7  # JXLHIP_ERR_UNSUPPORTED
        params, _ = synth.synth_frame(300, 200, mix=synth.MIX_DCT8)  # one group row: cannot be split in two
        p = abi.make_params(params)
        assert L.jxlhip_frame_begin(ctx, C.byref(p)) == -1
    finally:
        L.jxlhip_destroy(ctx)
    bad = (C.c_int * 1)(99)
    assert L.jxlhip_create_multi(bad, 1, None, C.byref(ctx)) == -1


def _upload(L, ctx, params, t, table_host):
    p = abi.make_params(params)
    assert L.jxlhip_frame_begin(ctx, C.byref(p)) == 0, L.jxlhip_last_error(ctx)
    npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
    dc3 = (C.c_void_p * 3)(*[a.ctypes.data for a in npy["dc"]])
    assert L.jxlhip_upload_side_info(ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                     npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                     npy["ytob_map"].ctypes.data, dc3, table_host.ctypes.data) == 0
    xs, ys = params["xsize"], params["ysize"]
    for g in range(((xs + 255) // 256) * ((ys + 255) // 256)):
        ptrs = (C.c_void_p * 3)(*[c[g * 65536:].ctypes.data for c in npy["coeffs"]])
        assert L.jxlhip_submit_group(ctx, g, ptrs, 65536) == 0, L.jxlhip_last_error(ctx)


@pytest.mark.parametrize("padded", [False, True])
def test_multi_context_frames_back_to_back_without_sync(oracle, monkeypatch, padded):
    """Ten frames (two different ones, alternating) through one multi context with NO jxlhip_sync in between: the
    persistent halo staging of a stripe may only be overwritten by the next frame's export once the neighbour has
    pulled the previous frame's rows (ev_pull, multi.inc), and the next frame's uploads must not overtake the
    previous frame's kernels (frame_ev, jxlhip_frame_begin).  padded: the caller's rows are wider than the pixels --
    the gather then is a 2-D copy and the padding bytes stay the caller's."""
    L = abi.load_library()
    xs, ys = 600, 1100
    monkeypatch.setenv("JXLHIP_FUSE", "0")
    monkeypatch.setenv("JXLHIP_MULTI_FORCE_GATHER", "1")
    cases, singles = [], []
    table_host = None
    for seed in (91, 17):
        params, t, _ = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=seed)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        dq = d.default_dequant_tables()
        d.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}, dq)
        singles.append(d.decode_frame().cpu().numpy())
        d.sync()
        table_host = dq.cpu().numpy()
        d.close()
        cases.append((params, t))
    assert not np.array_equal(singles[0], singles[1])
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(3)]
    ctx = C.c_void_p()
    assert L.jxlhip_create_multi((C.c_int * 3)(*devices), 3, None, C.byref(ctx)) == 0
    row_floats = xs * 3 + (16 if padded else 0)
    outs = []
    try:
        for i in range(10):
            params, t = cases[i % 2]
            _upload(L, ctx, params, t, table_host)
            out = torch.full((ys, row_floats), -7.0, dtype=torch.float32, device=f"cuda:{devices[0]}")
            assert L.jxlhip_decode_frame(ctx, out.data_ptr(), row_floats * 4, 0) == 0, L.jxlhip_last_error(ctx)
            outs.append(out)
        assert L.jxlhip_sync(ctx) == 0, L.jxlhip_last_error(ctx)
        for i, out in enumerate(outs):
            a = out.cpu().numpy()
            assert np.array_equal(a[:, :xs * 3].reshape(ys, xs, 3), singles[i % 2]), i
            assert (a[:, xs * 3:] == -7.0).all(), i  # row padding untouched
    finally:
        L.jxlhip_destroy(ctx)


@pytest.mark.parametrize("gab,epf", [(1, 1), (0, 0), (1, 2)])
@pytest.mark.parametrize("nstripes,host_out", [(3, False), (2, True)])
def test_multi_context_stripes_take_the_fused_kernel(oracle, monkeypatch, nstripes, host_out, gab, epf):
    """Stripes of a frame that qualifies for the fused kernel (12 Mpx and more; forced here with JXLHIP_FUSE=1) run it
    too: a stripe's DCT8 blocks are decoded inside its filter march, only those of its first / last block row also
    reach the planes -- the halo rows the neighbours pull.  Bit-equal to the whole frame through the fused kernel."""
    L = abi.load_library()
    xs, ys = 600, 1100
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=57)
    ref = fr.decode(threads=4)
    monkeypatch.setenv("JXLHIP_FUSE", "1")
    monkeypatch.setenv("JXLHIP_MULTI_FORCE_GATHER", "1")
    d = VarDctDecoder(0)
    d.begin_frame(params)
    dq = d.default_dequant_tables()
    d.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}, dq)
    single = d.decode_frame().cpu().numpy()
    d.sync()
    prof_single = None
    table_host = dq.cpu().numpy()
    d.close()
    ndev = torch.cuda.device_count()
    got = run_multi(L, [i % ndev for i in range(nstripes)], params, t, table_host, host_out)
    assert rel_err(got, ref) <= TIGHT
    assert rel_err(single, ref) <= TIGHT
    assert np.array_equal(got, single)


@pytest.mark.parametrize("nstripes,host_out", [(2, False), (3, True)])
@pytest.mark.parametrize("st,bits", [(1, 8), (0, 0)])
def test_multi_context_alpha_plane(oracle, monkeypatch, nstripes, host_out, st, bits):
    """jxlhip_set_alpha on a multi-device context: every stripe takes its own rows of the plane (through the
    context's pinned staging plane, jxlhip_alpha_staging); RGBA out equals the single-context result bit for bit."""
    L = abi.load_library()
    xs, ys = 600, 1100
    fmt = dict(transfer=1, sample_type=st, num_channels=4, bits_per_sample=bits)
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=93, output_kind=2, out_format=fmt,
                                     intensity_target=80.0)
    rng = np.random.default_rng(5)
    alpha = rng.integers(0, 256, size=(ys, xs)).astype(np.float32) * np.float32(1 / 255)
    d = VarDctDecoder(0)
    d.begin_frame(params)
    dq = d.default_dequant_tables()
    d.set_inputs({k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}, dq)
    d.set_alpha(alpha)
    single = d.decode_frame().cpu().numpy()
    d.sync()
    table_host = dq.cpu().numpy()
    d.close()
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(nstripes)]
    ctx = C.c_void_p()
    devs = (C.c_int * len(devices))(*devices)
    assert L.jxlhip_create_multi(devs, len(devices), None, C.byref(ctx)) == 0
    try:
        p = abi.make_params(params)
        assert L.jxlhip_frame_begin(ctx, C.byref(p)) == 0, L.jxlhip_last_error(ctx)
        npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
        dc3 = (C.c_void_p * 3)(*[a.ctypes.data for a in npy["dc"]])
        assert L.jxlhip_upload_side_info(ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data,
                                         npy["epf_sharpness"].ctypes.data, npy["ytox_map"].ctypes.data,
                                         npy["ytob_map"].ctypes.data, dc3, table_host.ctypes.data) == 0
        ng = ((xs + 255) // 256) * ((ys + 255) // 256)
        for g in range(ng):
            ptrs = (C.c_void_p * 3)(*[c[g * 65536:].ctypes.data for c in npy["coeffs"]])
            assert L.jxlhip_submit_group(ctx, g, ptrs, 65536) == 0, L.jxlhip_last_error(ctx)
        plane, stride = C.c_void_p(), C.c_size_t(0)
        assert L.jxlhip_alpha_staging(ctx, C.byref(plane), C.byref(stride)) == 0, L.jxlhip_last_error(ctx)
        assert stride.value == xs
        C.memmove(plane, alpha.ctypes.data, alpha.nbytes)
        assert L.jxlhip_set_alpha(ctx, plane, stride.value) == 0, L.jxlhip_last_error(ctx)
        px = 4 * (1 if st == 1 else 4)
        if host_out:
            out = np.zeros((ys, xs, 4), np.uint8 if st == 1 else np.float32)
            assert L.jxlhip_decode_frame_host(ctx, out.ctypes.data, xs * px, 0) == 0, L.jxlhip_last_error(ctx)
            got = out
        else:
            dev_out = torch.empty((ys, xs, 4), dtype=torch.uint8 if st == 1 else torch.float32, device=f"cuda:{devices[0]}")
            assert L.jxlhip_decode_frame(ctx, dev_out.data_ptr(), xs * px, 0) == 0, L.jxlhip_last_error(ctx)
            assert L.jxlhip_sync(ctx) == 0, L.jxlhip_last_error(ctx)
            got = dev_out.cpu().numpy()
    finally:
        L.jxlhip_destroy(ctx)
    assert np.array_equal(got, single)
    a = got[..., 3]
    assert np.array_equal(a, np.rint(alpha * 255).astype(np.uint8)) if st == 1 else np.array_equal(a, alpha)
