"""GPU parity tests: the HIP path (through the C ABI, libjxl_hip.so) against the
CPU oracle on the same seeded synthetic frames.

Tolerances.  The reference holds its own two executors to a relative error of
2e-4 (lib/jxl/render_pipeline/render_pipeline_test.cc:321-327) and the IDCT to
1e-7*N (lib/jxl/dct_test.cc:191-214).  The HIP kernels follow the oracle's
operation order, so the observed differences are a few ulp (v_rcp_f32 instead
of an exact divide in AdjustQuantBias / the EPF normalisation); the tests
assert a much tighter bound (TIGHT) than the reference's own bar (REF_TOL) so
regressions in operation order are caught.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import frames
from libjxl_amd import VarDctDecoder, abi, synth

pytestmark = pytest.mark.gpu

REF_TOL = 2e-4   # render_pipeline_test.cc:321-327
TIGHT = 2e-5     # what we actually hold the kernels to (relative to the range)


def to_dev(t):
    return {k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda()) for k, v in t.items()}


@pytest.fixture(scope="module")
def dec():
    d = VarDctDecoder(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def dq(dec):
    p, _ = synth.synth_frame(8, 8, mix=synth.MIX_DCT8)
    dec.begin_frame(p)
    t = dec.default_dequant_tables()
    dec.sync()
    return t


def crop_planes(dec, params):
    return dec.export_xyb()


def rel_err(got, ref):
    scale = max(1.0, float(np.abs(ref).max()))
    return float(np.abs(got.astype(np.float64) - ref).max()) / scale


def test_default_dequant_tables_match_oracle(dq, oracle):
    ref = oracle.default_dequant_tables()
    got = dq.cpu().numpy()
    # same arithmetic (FastPowf restated with explicit fma): bit-identical
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
        np.abs(got - ref).max()


@pytest.mark.parametrize("strategy", list(range(27)))
def test_blocks_each_strategy(dec, dq, oracle, strategy):
    """dequant + CfL + LLF + inverse transform for one strategy at a time."""
    cx, cy = synth.COVERED_X[strategy], synth.COVERED_Y[strategy]
    xs = max(272, 8 * cx + 24)   # ragged: last group clipped, not a multiple of the block
    ys = max(264, 8 * cy + 8)
    if max(cx, cy) >= 16:
        xs, ys = 8 * cx + 256, 8 * cy
    params, t, fr = frames.make_case(xs, ys, mix={strategy: 3.0 * cx * cy, 0: 1.0}, gab=False,
                                     epf_iters=0, seed=1000 + strategy)
    acs = t["ac_strategy"].numpy()
    used = set((acs[(acs & 1) == 1] >> 1).tolist())
    assert strategy in used
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    dec.decode_blocks()
    dec.sync()
    ref = fr.decode_groups()
    got = crop_planes(dec, params)
    for c in range(3):
        assert rel_err(got[c], ref[c]) <= TIGHT, (strategy, c)


@pytest.mark.parametrize("size", [(8, 8), (3, 8), (64, 64), (256, 256), (258, 258), (533, 401),
                                  (777, 777), (1024, 1024)])
def test_blocks_mixed_sizes(dec, dq, oracle, size):
    params, t, fr = frames.make_case(*size, mix=synth.MIX_ALL, gab=False, epf_iters=0, seed=7)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    dec.decode_blocks()
    dec.sync()
    ref = fr.decode_groups()
    got = crop_planes(dec, params)
    for c in range(3):
        assert rel_err(got[c], ref[c]) <= TIGHT


def test_blocks_int32_coefficients(dec, dq, oracle):
    params, t, fr = frames.make_case(400, 300, mix=synth.MIX_ALL, gab=False, epf_iters=0,
                                     coeff_type=1, amp=200000.0, decay=3.0, seed=11)
    assert t["coeffs"][1].dtype == torch.int32
    assert int(t["coeffs"][1].abs().max()) > 32767  # really needs 32 bits
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    dec.decode_blocks()
    dec.sync()
    ref = fr.decode_groups()
    got = crop_planes(dec, params)
    for c in range(3):
        assert rel_err(got[c], ref[c]) <= TIGHT


def test_sigma_matches_oracle(dec, dq, oracle):
    params, t, fr = frames.make_case(533, 401, mix=synth.MIX_ALL, gab=True, epf_iters=2, seed=3,
                                     custom_lf=True)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    dec.decode_blocks()
    dec.sync()
    ref = fr.compute_sigma()
    got = dec.sigma().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)


FILTER_CONFIGS = [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3)]


@pytest.mark.parametrize("gab,epf", FILTER_CONFIGS)
@pytest.mark.parametrize("size", [(533, 401), (3, 8), (64, 33)])
def test_full_pipeline_rgb(dec, dq, oracle, gab, epf, size):
    """Whole path -> linear RGB, all stage lists of PreparePipeline, ragged sizes
    (mirror borders at the true image edge, also for images smaller than the halo)."""
    params, t, fr = frames.make_case(*size, mix=synth.MIX_ALL, gab=bool(gab), epf_iters=epf,
                                     seed=21 + gab + 2 * epf)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    out = dec.decode_frame()
    dec.sync()
    ref = fr.decode(threads=4)
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    e = rel_err(got, ref)
    assert e <= REF_TOL
    assert e <= TIGHT, e


@pytest.mark.parametrize("gab,epf", [(1, 0), (0, 1), (1, 1), (1, 2), (0, 2)])
@pytest.mark.parametrize("size", [(533, 401), (61, 70), (1000, 130)])
def test_generic_lds_filter_kernel_also_matches(dq, oracle, monkeypatch, gab, epf, size):
    """Stage lists with <= 2 EPF passes normally take the register/DPP kernel; the
    generic LDS kernel (used for epf_iters == 3 and frames narrower than 16) must agree on
    them too."""
    monkeypatch.setenv("JXLHIP_FILTERS", "generic")
    d = VarDctDecoder(0)
    monkeypatch.delenv("JXLHIP_FILTERS")
    params, t, fr = frames.make_case(*size, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf,
                                     seed=77)
    d.begin_frame(params)
    d.set_inputs(to_dev(t), dq)
    out = d.decode_frame()
    d.sync()
    assert rel_err(out.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    d.close()


@pytest.mark.parametrize("gab,epf", [(1, 1), (1, 3)])
def test_full_pipeline_xyb_output_and_custom_lf(dec, dq, oracle, gab, epf):
    params, t, fr = frames.make_case(300, 270, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf,
                                     seed=5, output_kind=0, custom_lf=True,
                                     intensity_target=4000.0)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    out = dec.decode_frame()
    dec.sync()
    ref = fr.decode(threads=4)
    assert rel_err(out.cpu().numpy(), ref) <= TIGHT


def test_c1_1024_full_pipeline(dec, dq, oracle):
    """BASELINE config 1: 1024x1024 d1.0-like (Gaborish + EPF1)."""
    params, t, fr = frames.make_case(1024, 1024, mix=synth.MIX_D1, gab=True, epf_iters=1)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    out = dec.decode_frame()
    dec.sync()
    ref = fr.decode(threads=8)
    assert rel_err(out.cpu().numpy(), ref) <= TIGHT


@pytest.mark.parametrize("epf", [3, 2, 1])
def test_stripes_with_halo_exchange_equal_whole_frame(dec, dq, oracle, epf):
    """Multi-GPU decomposition on one device: two contexts decode the two
    group-row stripes, swap halo rows, and must reproduce the whole-frame
    result bit for bit (generic LDS kernel for 3 EPF passes, row-march kernel below)."""
    params, t, fr = frames.make_case(600, 700, mix=synth.MIX_ALL, gab=True, epf_iters=epf, seed=9)
    devt = to_dev(t)
    dec.begin_frame(params)
    dec.set_inputs(devt, dq)
    whole = dec.decode_frame().clone()
    dec.sync()
    parts = []
    decs = []
    for (g0, gr) in [(0, 1), (1, 2)]:
        d = VarDctDecoder(0)
        p = dict(params, stripe_group_y0=g0, stripe_group_rows=gr)
        d.begin_frame(p)
        d.set_inputs(devt, dq)
        d.decode_blocks()
        decs.append(d)
    torch.cuda.synchronize()
    # stripe 0 sends its last rows down, stripe 1 sends its first rows up
    decs[1].halo_import(0, decs[0].halo_export(1))
    decs[0].halo_import(1, decs[1].halo_export(0))
    for d in decs:
        out = d.alloc_output()
        d.decode_filters(out)
        d.sync()
        parts.append(out)
    got = torch.cat(parts, dim=0)
    assert got.shape == whole.shape
    assert torch.equal(got, whole)
    assert rel_err(whole.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    for d in decs:
        d.close()


@pytest.mark.parametrize("gab,epf", [(1, 1), (1, 2), (0, 0)])
def test_stripes_through_the_fused_kernel_equal_whole_frame(dq, oracle, gab, epf, monkeypatch):
    """The split calls on a STRIPE take the fused kernel when the frame qualifies (forced here: JXLHIP_FUSE=1):
    k_prepare keeps the DCT8 blocks off the work list except those of the stripe's first / last block row next to a
    neighbour, which reach the planes too -- the halo rows that are exported.  Three stripes (the middle one has
    neighbours on both sides), bit-equal to the whole frame through the fused kernel; the profile slots say which
    kernel ran."""
    monkeypatch.setenv("JXLHIP_FUSE", "1")
    params, t, fr = frames.make_case(600, 1100, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=19)
    devt = to_dev(t)
    d0 = VarDctDecoder(0)
    d0.begin_frame(params)
    d0.set_inputs(devt, dq)
    whole = d0.decode_frame().clone()
    d0.sync()
    d0.close()
    decs = []
    for (g0, gr) in [(0, 2), (2, 1), (3, 2)]:
        d = VarDctDecoder(0)
        d.begin_frame(dict(params, stripe_group_y0=g0, stripe_group_rows=gr))
        d.set_inputs(devt, dq)
        d.profile(True)
        d.decode_blocks()
        decs.append(d)
    torch.cuda.synchronize()
    for i in range(2 if decs[0].halo_rows() else 0):  # (no loop filter: no halo rows to exchange)
        decs[i + 1].halo_import(0, decs[i].halo_export(1))
        decs[i].halo_import(1, decs[i + 1].halo_export(0))
    parts = []
    for d in decs:
        out = d.alloc_output()
        d.decode_filters(out)
        d.sync()
        parts.append(out)
        assert "fused" in d.profile_read() and "filters" not in d.profile_read()
        with pytest.raises(Exception):
            d.export_xyb()  # the planes do not hold the inner DCT8 blocks
    got = torch.cat(parts, dim=0)
    assert torch.equal(got, whole)
    assert rel_err(whole.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    for d in decs:
        d.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_decode_frame_under_stream_capture_replays_bit_identically(dq, oracle, fuse, monkeypatch):
    """A caller may record jxlhip_decode_frame into a hipGraph and replay it (bench.py's `graph_replay`): under capture
    the library keeps its per-frame state inside the graph -- the work-list counters are zeroed by a KERNEL node of the
    graph (a memset node at the root was seen to overtake the previous replay of the same graph: a memory fault), and
    the alternating counter blocks of consecutive direct calls are not used.  Five replays back to back, then a direct
    call again: the same pixels every time."""
    monkeypatch.setenv("JXLHIP_FUSE", fuse)
    params, t, fr = frames.make_case(1000, 520, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=61)
    d = VarDctDecoder(0)
    d.begin_frame(params)
    d.set_inputs(to_dev(t), dq)
    want = d.decode_frame().clone()
    d.sync()
    assert rel_err(want.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    cs = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    cs.wait_stream(main)
    d.set_stream(cs)
    out = d.alloc_output()
    with torch.cuda.stream(cs):
        d.decode_frame(out)  # (everything allocated before the capture)
    cs.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        d.decode_frame(out)
    for _ in range(5):
        out.zero_()
        g.replay()
        g.replay()  # back to back, no synchronisation in between
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    d.set_stream(main)
    out.zero_()
    d.decode_frame(out)
    d.sync()
    assert torch.equal(out, want)
    d.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_graph_replays_interleaved_with_direct_calls_on_one_stream(dq, oracle, fuse, monkeypatch):
    """Replays of a captured frame and direct jxlhip_decode_frame calls on the SAME stream, with no jxlhip_set_stream in
    between (which would reset the host's "counter block is clean" flags): capture, two direct calls (the second leaves
    block 0 marked clean), a replay, another direct call -- round 5 put captured frames on block 0, so that last call
    started k_prepare on the replay's non-zero counters.  Captured frames now use counter blocks of their own
    (context.hip: kCaptureBase).  Every frame must be the same pixels, and the stream must report no fault."""
    monkeypatch.setenv("JXLHIP_FUSE", fuse)
    params, t, fr = frames.make_case(1000, 520, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=62)
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        d = VarDctDecoder(0)  # bound to cs for its whole life
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        out = d.alloc_output()
        want = d.decode_frame().clone()
        d.decode_frame(out)
    cs.synchronize()
    assert rel_err(want.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        d.decode_frame(out)
    with torch.cuda.stream(cs):
        for round_ in range(3):
            for _ in range(2 + round_):  # two or more direct calls: both alternating blocks have been used and re-zeroed
                out.zero_()
                d.decode_frame(out)
                cs.synchronize()
                assert torch.equal(out, want), ("direct", round_)
            out.zero_()
            g.replay()
            cs.synchronize()
            assert torch.equal(out, want), ("replay", round_)
        out.zero_()
        d.decode_frame(out)
        d.sync()
        assert torch.equal(out, want)
    d.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("gab,epf,interior", [(1, 1, True), (1, 2, False), (1, 3, False), (0, 0, True)])
def test_stripe_step_in_three_calls_equals_whole_frame(dq, oracle, gab, epf, interior, fuse, monkeypatch):
    """jxlhip_stripe_begin (phase 1 + both exports) / jxlhip_decode_filters_rows (the interior) / jxlhip_stripe_finish
    (both imports + the boundary block rows): what libjxl_amd.stripes issues per rank since round 5.  Three stripes on
    one device, the send buffers of one handed to the other as its receive buffers: bit-equal to the whole frame."""
    monkeypatch.setenv("JXLHIP_FUSE", fuse)
    params, t, fr = frames.make_case(600, 1100, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=29)
    devt = to_dev(t)
    d0 = VarDctDecoder(0)
    d0.begin_frame(params)
    d0.set_inputs(devt, dq)
    whole = d0.decode_frame().clone()
    d0.sync()
    d0.close()
    parts = [(0, 2), (2, 1), (3, 2)]
    decs, bufs, outs = [], [], []
    for (g0, gr) in parts:
        d = VarDctDecoder(0)
        d.begin_frame(dict(params, stripe_group_y0=g0, stripe_group_rows=gr))
        d.set_inputs(devt, dq)
        h = d.halo_rows()
        mk = lambda: torch.full((3, h, 600), float("nan"), dtype=torch.float32, device="cuda")  # noqa: E731
        up, dn = g0 > 0, g0 + gr < 5
        b = dict(up=mk() if up and h else None, dn=mk() if dn and h else None)
        d.stripe_begin(b["up"], b["dn"])
        out = d.alloc_output()
        y0, y1 = d.stripe_rows()
        rows = (y0 + 8 if up else y0, y1 - 8 if dn else y1)
        if interior:
            d.decode_filters(out, rows=rows)
        decs.append(d), bufs.append(b), outs.append((out, rows if interior else None))
    torch.cuda.synchronize()
    for i, d in enumerate(decs):  # a stripe receives the rows its neighbours exported towards it
        d.stripe_finish(outs[i][0], bufs[i - 1]["dn"] if i > 0 else None, bufs[i + 1]["up"] if i + 1 < len(decs) else None,
                        outs[i][1])
        d.sync()
    got = torch.cat([o for o, _ in outs], dim=0)
    assert torch.equal(got, whole)
    assert rel_err(whole.cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    for d in decs:
        d.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("gab,epf", [(1, 1), (1, 2), (0, 1)])
def test_stripe_interior_rows_before_the_halo_arrives(dq, oracle, gab, epf, fuse, monkeypatch):
    """jxlhip_decode_filters_rows: a stripe filters the rows whose support stays inside it BEFORE its neighbours' halo
    rows are installed (what libjxl_amd.stripes and jxlhip_create_multi do while the halo messages travel), then the
    first / last block row.  Three stripes, fused kernel and two-phase: bit-equal to the whole frame; and the interior
    rows really do not read the halo (they are computed with the halo rows still unset)."""
    monkeypatch.setenv("JXLHIP_FUSE", fuse)
    params, t, fr = frames.make_case(600, 1100, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=23)
    devt = to_dev(t)
    d0 = VarDctDecoder(0)
    d0.begin_frame(params)
    d0.set_inputs(devt, dq)
    whole = d0.decode_frame().clone()
    d0.sync()
    d0.close()
    parts, decs = [], []
    bounds = []
    for (g0, gr) in [(0, 2), (2, 1), (3, 2)]:
        d = VarDctDecoder(0)
        d.begin_frame(dict(params, stripe_group_y0=g0, stripe_group_rows=gr))
        d.set_inputs(devt, dq)
        d.decode_blocks()
        y0, y1 = d.stripe_rows()
        ya = y0 + 8 if g0 > 0 else y0
        yb = y1 - 8 if g0 + gr < 5 else y1
        out = d.alloc_output()
        out.fill_(float("nan"))
        d.decode_filters(out, rows=(ya, yb))  # no halo rows installed yet
        d.sync()
        assert not torch.isnan(out[ya - y0:yb - y0]).any() and torch.isnan(out[:ya - y0]).all() and torch.isnan(out[yb - y0:]).all()
        decs.append(d)
        parts.append(out)
        bounds.append((y0, ya, yb, y1))
    for i in range(2):
        decs[i + 1].halo_import(0, decs[i].halo_export(1))
        decs[i].halo_import(1, decs[i + 1].halo_export(0))
    for d, out, (y0, ya, yb, y1) in zip(decs, parts, bounds):
        d.decode_filters(out, rows=(y0, ya))
        d.decode_filters(out, rows=(yb, y1))
        d.sync()
        with pytest.raises(Exception):
            d.decode_filters(out, rows=(y0 + 4, y1))  # not a block-row multiple
    got = torch.cat(parts, dim=0)
    assert torch.equal(got, whole), (got - whole).abs().max()
    for d in decs:
        d.close()


def test_concurrency_hint_moves_the_fused_threshold(dq, oracle):
    """jxlhip_set_concurrency_hint: a context that runs ALONE fuses from 12 Mpx, one of several in flight from 6 Mpx (the
    device is then bound by HBM traffic, which the fused path has less of).  A 3328x2048 frame (6.8 Mpx): two-phase
    without the hint, fused with it (the profile slots say which), both within the bar of the oracle."""
    params, t, fr = frames.make_case(3328, 2048, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=3)
    ref = fr.decode(threads=8)
    outs = {}
    for hint in (1, 3):
        d = VarDctDecoder(0)
        d.set_concurrency_hint(hint)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        d.profile(True)
        outs[hint] = d.decode_frame().cpu().numpy()
        d.sync()
        slots = d.profile_read()
        assert ("fused" in slots) == (hint == 3) and ("filters" in slots) == (hint == 1), slots
        d.close()
    assert rel_err(outs[1], ref) <= TIGHT and rel_err(outs[3], ref) <= TIGHT
    with pytest.raises(Exception):
        VarDctDecoder(0).set_concurrency_hint(0)


def test_upload_path_equals_device_path(dec, dq, oracle):
    """Host-pointer hand-off (upload_side_info + submit_group per group, as a
    FrameDecoder would call it) gives the same pixels as device-resident inputs."""
    params, t, fr = frames.make_case(520, 300, mix=synth.MIX_ALL, gab=True, epf_iters=1, seed=13)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    a = dec.decode_frame().clone()
    dec.sync()
    d2 = VarDctDecoder(0)
    d2.begin_frame(params)
    L = d2.L
    dqh = dq.cpu().numpy()
    npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
    dc3 = (C.c_void_p * 3)(*[x.ctypes.data for x in npy["dc"]])
    rc = L.jxlhip_upload_side_info(d2.ctx, npy["ac_strategy"].ctypes.data,
                                   npy["raw_quant"].ctypes.data, npy["epf_sharpness"].ctypes.data,
                                   npy["ytox_map"].ctypes.data, npy["ytob_map"].ctypes.data, dc3,
                                   dqh.ctypes.data)
    assert rc == 0
    ngroups = ((520 + 255) // 256) * ((300 + 255) // 256)
    order = np.random.default_rng(0).permutation(ngroups)  # any order (FakeParallelRunner-like)
    for g in order:
        ptrs = (C.c_void_p * 3)(*[npy["coeffs"][c][g * 65536:].ctypes.data for c in range(3)])
        assert L.jxlhip_submit_group(d2.ctx, int(g), ptrs, 65536) == 0
    b = d2.decode_frame()
    d2.sync()
    assert torch.equal(a, b)
    d2.close()


def test_bad_strategy_map_is_reported(dec, dq):
    params, t = synth.synth_frame(256, 256, mix=synth.MIX_DCT8, gab=False, epf_iters=0)
    acs = t["ac_strategy"].clone()
    acs[31, 31] = (5 << 1) | 1   # a 32x32 block starting in the last cell: overflows the group
    t["ac_strategy"] = acs
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    dec.decode_blocks()
    with pytest.raises(abi.JxlHipError, match="format constraint"):
        dec.sync()
    dec.sync()  # flag is cleared, context usable again


def test_used_acs_hint(dec, dq, oracle):
    """jxlhip_frame_params::used_acs: families without a set bit are not launched (same pixels
    as with the mask unknown); a mask that rules out a strategy the frame uses is an error."""
    params, t, fr = frames.make_case(520, 300, mix={0: 3, 4: 1, 12: 0.5}, gab=True, epf_iters=1, seed=3)
    assert params["used_acs"] == (1 << 0) | (1 << 4) | (1 << 12)
    outs = []
    for mask in (params["used_acs"], 0):
        dec.begin_frame(dict(params, used_acs=mask))
        dec.set_inputs(to_dev(t), dq)
        outs.append(dec.decode_frame().clone())
        dec.sync()
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0].cpu().numpy(), fr.decode(threads=4)) <= TIGHT
    dec.begin_frame(dict(params, used_acs=(1 << 0) | (1 << 12)))   # 16x16 ruled out
    dec.set_inputs(to_dev(t), dq)
    dec.decode_frame()
    with pytest.raises(abi.JxlHipError):
        dec.sync()


def test_dequant_dc_and_smoothing(dec, oracle):
    import oracle as O
    xs, ys = 333, 270
    params, _ = synth.synth_frame(xs, ys, mix=synth.MIX_DCT8)
    dec.begin_frame(params)
    xsb, ysb = (xs + 7) // 8, (ys + 7) // 8
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:ysb, 0:xsb]
    q = [(200 * np.sin(xx * 0.05 + c) * np.cos(yy * 0.04) + rng.integers(-3, 4, (ysb, xsb))).astype(np.int32)
         for c in range(3)]
    inv_gs = np.float32(65536.0 / params["global_scale"])
    mul = np.array([np.float32(inv_gs / np.float32(params["quant_dc"])) * np.float32(v)
                    for v in (1 / 4096.0, 1 / 512.0, 1 / 256.0)], np.float32)
    ref = [np.zeros((ysb, xsb), np.float32) for _ in range(3)]
    O.lib().jxo_dequant_dc(xsb, ysb, O._p3(q), O._p3(ref), O._p(mul), 0.1, 0.9)
    O.lib().jxo_adaptive_dc_smoothing(xsb, ysb, O._p(mul), O._p3(ref))
    qd = [torch.from_numpy(a).cuda() for a in q]
    od = [torch.empty((ysb, xsb), dtype=torch.float32, device="cuda") for _ in range(3)]
    rc = dec.L.jxlhip_dequant_dc(dec.ctx, (C.c_void_p * 3)(*[a.data_ptr() for a in qd]),
                                 (C.c_void_p * 3)(*[a.data_ptr() for a in od]), None,
                                 0.1, 0.9, 1)
    assert rc == 0
    dec.sync()
    for c in range(3):
        assert np.array_equal(od[c].cpu().numpy(), ref[c])


@pytest.mark.parametrize("gab,epf", [(1, 1), (0, 0), (1, 2), (0, 1), (1, 0), (0, 2)])
@pytest.mark.parametrize("size", [(1000, 520), (333, 268), (112, 64), (2048, 1029 - 5)])
def test_fused_kernel_matches_two_phase_and_oracle(dq, oracle, gab, epf, size, monkeypatch):
    """jxlhip_decode_frame on a whole frame runs the fused kernel (kernels_fused.hip: DCT8 decoded inside
    the filter march, other classes through the planes); JXLHIP_FUSE=0 forces the two-phase path.  Both
    against the oracle; the fused path must actually have been taken (profile slot names do not tell:
    the kernel count does -- with fusion the DCT8 list stays empty)."""
    xs, ys = size
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=bool(gab), epf_iters=epf, seed=77 + xs)
    ref = fr.decode(threads=4)
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("JXLHIP_FUSE", fuse)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        outs[fuse] = d.decode_frame().cpu().numpy()
        d.sync()
        d.close()
    assert rel_err(outs["0"], ref) <= TIGHT
    assert rel_err(outs["1"], ref) <= TIGHT, np.argwhere(np.abs(outs["1"] - ref) > 1e-3)[:5]


@pytest.mark.parametrize("gab,epf", [(1, 1), (0, 0)])
@pytest.mark.parametrize("size,coeff_type", [((1000, 520), 0), ((333, 268), 1), ((112, 64), 0), ((2048, 1029 - 5), 0),
                                             ((1500, 2100), 0), ((113, 40), 0), ((225, 24), "dct8"), ((964, 300), "dct8")])
def test_fused_producer_consumer_chunking_is_bit_identical(dq, oracle, gab, epf, size, coeff_type, monkeypatch):
    """k_fused_pc (kernels_fused.hip): the window march split over a producing and a marching wave with a
    double-buffered slab.  The pixels do not depend on how the rows are cut into window chunks (JXLHIP_FUSED_PC_RH: the
    interior march with its row tests resolved at compile time against the generic one, chunk heads / tails), and are
    within 2e-5 of the oracle.  Sizes: several row chunks per window (1500x2100), one block row, edge windows, one column
    / one cell past a window (113, 225), a last block row of 4 rows, int32 coefficients, frames of DCT8 only (every cell
    of every block row through the producer's two decode steps, no plane copies).  (Rounds 3-5 also held it bit-equal
    to a single-wave fused kernel, removed in round 6.)"""
    xs, ys = size
    kw = dict(coeff_type=1, amp=200000.0, decay=3.0) if coeff_type == 1 else {}
    mix = {0: 100} if coeff_type == "dct8" else synth.MIX_D1
    params, t, fr = frames.make_case(xs, ys, mix=mix, gab=bool(gab), epf_iters=epf, seed=31 + xs, **kw)
    ref = fr.decode(threads=4)
    outs = {}
    monkeypatch.setenv("JXLHIP_FUSE", "1")
    for rh in ("0", "64", "24"):
        monkeypatch.setenv("JXLHIP_FUSED_PC_RH", rh)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        outs[rh] = d.decode_frame().cpu().numpy()
        d.sync()
        d.close()
    assert rel_err(outs["0"], ref) <= TIGHT
    assert np.array_equal(outs["64"], outs["0"]), np.argwhere(outs["64"] != outs["0"])[:5]
    assert np.array_equal(outs["24"], outs["0"]), np.argwhere(outs["24"] != outs["0"])[:5]


@pytest.mark.parametrize("gab,out", [(1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("size,coeff_type,mix", [((1000, 520), 0, None), ((333, 268), 1, None), ((117, 68), 0, None),
                                                  ((2048, 1029 - 5), 0, "all"), ((1500, 700), 1, "all"), ((258, 2100), 0, None)])
def test_three_epf_iterations_through_the_fused_producer_match_oracle(dq, oracle, gab, out, size, coeff_type, mix, monkeypatch):
    """epf_iters = 3 with the frame in fused mode (forced here; automatic from the whole-frame size rule): k_fused_pc0
    marches [Gaborish] + EPF0 from the slab its producing wave fills -- the DCT8 cells decoded in the wave, never written
    to the first plane set -- into the second plane set; EPF1 + EPF2 + output from there as in the two-phase path
    (kernels_fused.hip part 3, epf0_march.h).  Against the oracle at the bar of every other path, and bit-equal to the
    two-phase path (k_epf0): the same arithmetic on the same values, whichever kernel decoded the DCT8 blocks.  Sizes:
    windows cut by the frame's edges, ragged bottoms, several row chunks per window, int32 coefficients; float RGB and
    planar XYB outputs."""
    xs, ys = size
    kw = dict(coeff_type=1, amp=200000.0, decay=3.0) if coeff_type else {}
    m = {None: synth.MIX_D1, "all": synth.MIX_ALL}[mix]
    params, t, fr = frames.make_case(xs, ys, mix=m, gab=bool(gab), epf_iters=3, seed=57 + xs, output_kind=out, **kw)
    outs, kernels = {}, {}
    for fuse, rh in (("1", "0"), ("1", "40"), ("0", "0")):
        monkeypatch.setenv("JXLHIP_FUSE", fuse)
        monkeypatch.setenv("JXLHIP_FUSED_PC_RH", rh)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        d.profile(True)
        o = d.decode_frame()
        d.sync()
        outs[fuse + rh] = o.cpu().numpy()
        kernels[fuse + rh] = d.profile_read()
        d.close()
    ref = fr.decode(threads=4)
    for k, v in outs.items():
        assert rel_err(v, ref) <= TIGHT, (k, np.argwhere(np.abs(v - ref) > 1e-3)[:5])
    assert np.array_equal(outs["10"], outs["140"])  # the chunking does not change a sample
    assert np.array_equal(outs["10"], outs["00"]), np.argwhere(outs["10"] != outs["00"])[:5]
    assert "epf0" in kernels["10"] and "epf0" in kernels["00"]


@pytest.mark.parametrize("coeff_type", [0, 1])
@pytest.mark.parametrize("size,mix,want", [((8 * 4 + 256, 8 * 4 + 8), {5: 48.0, 0: 1.0}, (5,)),
                                           ((8 * 4 + 256, 8 * 4 + 8), {4: 48.0, 0: 1.0}, (4,)),
                                           ((1000, 520), {4: 8.0, 5: 8.0, 0: 1.0, 6: 1.0}, (4, 5)),
                                           ((1000, 520), None, (4, 5)), ((258, 258), None, (4, 5))])
def test_mfma_dct32_dct16_match_oracle_and_row_lane_path(dq, oracle, size, mix, want, coeff_type, monkeypatch):
    """JXLHIP_MFMA=1 sends DCT32X32 and DCT16X16 varblocks through the matrix-core kernels (kernels_mfma.hip: two
    v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 products per channel); everything else keeps its kernel.
    Same bar as the butterfly path -- and the two must agree with each other to rounding."""
    xs, ys = size
    kw = dict(coeff_type=1, amp=200000.0, decay=3.0) if coeff_type else {}
    params, t, fr = frames.make_case(xs, ys, mix=mix or synth.MIX_ALL, gab=False, epf_iters=0, seed=5 + xs, **kw)
    acs = t["ac_strategy"].numpy()
    assert set(want) <= set((acs[(acs & 1) == 1] >> 1).tolist())
    ref = fr.decode_groups()
    outs = {}
    for mfma in ("1", "0"):
        monkeypatch.setenv("JXLHIP_MFMA", mfma)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        d.decode_blocks()
        d.sync()
        outs[mfma] = d.export_xyb()
        d.close()
    for c in range(3):
        assert rel_err(outs["0"][c], ref[c]) <= TIGHT
        assert rel_err(outs["1"][c], ref[c]) <= TIGHT, (c, np.argwhere(np.abs(outs["1"][c] - ref[c]) > 1e-3)[:5])
    # the MFMA kernel really ran: a dense product rounds differently from the butterflies
    assert any(not np.array_equal(outs["0"][c], outs["1"][c]) for c in range(3))


def test_all_dct16_frame_of_16_mpx_takes_the_matrix_cores_by_default(dq, oracle, monkeypatch):
    """The context's own rule (LaunchBlocksBand): DCT16X16 alone in the row-per-lane families, 16 Mpx and more ->
    k_transform_mfma16.  Checked against the oracle, and against the butterflies to see that the rule engaged."""
    params, t, fr = frames.make_case(4096, 4096, mix={4: 1.0}, gab=False, epf_iters=0, seed=77)
    ref = fr.decode(threads=8)
    outs = {}
    for mfma in (None, "0"):
        if mfma is None:
            monkeypatch.delenv("JXLHIP_MFMA", raising=False)
        else:
            monkeypatch.setenv("JXLHIP_MFMA", mfma)
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        outs[mfma] = d.decode_frame().cpu().numpy()
        d.sync()
        d.close()
    assert rel_err(outs[None], ref) <= TIGHT
    assert rel_err(outs["0"], ref) <= TIGHT
    assert not np.array_equal(outs[None], outs["0"])


@pytest.mark.parametrize("gab", [1, 0])
@pytest.mark.parametrize("size,okind", [((1000, 520), 1), ((2048, 1029), 1), ((129, 16), 1), ((777, 300), 0)])
def test_three_epf_iterations_take_the_epf0_march(dec, dq, oracle, gab, size, okind):
    """epf_iters = 3: k_epf0 ([Gaborish] + EPF0 into the second plane set, kernels_epf0.hip) + the EPF1 + EPF2
    march, at sizes with several strips / row bands per wave, odd widths and the minimum height."""
    params, t, fr = frames.make_case(*size, mix=synth.MIX_D1, gab=bool(gab), epf_iters=3, seed=3 + size[0],
                                     output_kind=okind)
    dec.begin_frame(params)
    dec.set_inputs(to_dev(t), dq)
    out = dec.decode_frame()
    dec.sync()
    ref = fr.decode(threads=4)
    prof = dec.profile() if hasattr(dec, "profile") else None
    assert rel_err(out.cpu().numpy(), ref) <= TIGHT


@pytest.mark.parametrize("size,gab,epf", [((520, 300), True, 1), ((777, 300), False, 0), ((1000, 520), True, 2), ((520, 264), True, 3),
                                          ((4096, 3200), True, 1)])
@pytest.mark.parametrize("st,bits,nc_tf", [(1, 8, 1), (2, 16, 1), (2, 12, 1), (3, 0, 1), (0, 0, 1), (0, 0, 0), (2, 16, 2)])
def test_alpha_plane_in_every_packed_path(dq, size, gab, epf, st, bits, nc_tf, monkeypatch):
    """jxlhip_set_alpha (FilterParams::alpha): the fourth sample of a 4-channel packed output is the plane's value
    through MakeUnsigned / the float conversions -- like a colour sample, without the transfer function
    (stage_write.cc:350-366) -- whichever kernel writes the frame: the fixed-format and general row marches, the generic
    LDS kernel, k_epf0 + march, and (13 Mpx) the fused kernels; the colour samples do not change."""
    xs, ys = size
    tf = nc_tf  # 0 linear, 1 sRGB, 2 PQ
    if xs > 2000 and (st, tf) not in ((2, 2), (1, 1), (0, 0)):
        pytest.skip("the 13 Mpx frame: one format per kernel family")
    fmt = dict(transfer=tf, sample_type=st, num_channels=4, bits_per_sample=bits, tf_param=1000.0 if tf == 2 else 0.0)
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=gab, epf_iters=epf, seed=7 + xs, output_kind=2,
                                     out_format=fmt, intensity_target=1000.0 if tf == 2 else 80.0)
    rng = np.random.default_rng(xs + st)
    levels = 255 if st == 1 else 65535
    alpha = rng.integers(0, levels + 1, size=(ys, xs)).astype(np.float32) * np.float32(1.0 / levels)
    alpha[: ys // 4] = 1.0
    alpha[ys // 4: ys // 2, : xs // 3] = 0.0
    outs = {}
    for generic in ("", "generic"):
        if generic:
            monkeypatch.setenv("JXLHIP_FILTERS", generic)
        else:
            monkeypatch.delenv("JXLHIP_FILTERS", raising=False)
        if generic and xs > 2000:
            continue
        d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(to_dev(t), dq)
        opaque = d.decode_frame().cpu().numpy().copy()
        d.set_alpha(alpha)
        got = d.decode_frame().cpu().numpy().copy()
        d.begin_frame(params)           # a new frame: opaque again
        d.set_inputs(to_dev(t), dq)
        again = d.decode_frame().cpu().numpy().copy()
        d.sync()
        d.close()
        assert np.array_equal(again, opaque)
        assert np.array_equal(got[..., :3], opaque[..., :3])
        a = got[..., 3]
        if st == 1:
            assert (opaque[..., 3] == 255).all()
            assert np.array_equal(a, np.rint(alpha * 255.0).astype(np.uint8))      # the dither never crosses .5
        elif st == 2:
            full = (1 << bits) - 1
            au = a.view(np.uint16)
            assert (opaque[..., 3].view(np.uint16) == full).all()
            assert np.abs(au.astype(np.int32) - np.rint(alpha.astype(np.float64) * full)).max() <= (0 if bits == 16 else 1)
        elif st == 3:
            assert np.array_equal(a.view(np.uint16).view(np.float16), alpha.astype(np.float16))
        else:
            assert np.array_equal(a, alpha)
        outs[generic] = got
    if "generic" in outs:
        assert np.array_equal(outs[""][..., 3], outs["generic"][..., 3])
