"""Committed golden vectors (tests/golden/*.npz, made by make_golden.py).
CPU: the oracle must still reproduce them bit for bit (they pin the checker
against accidental edits).  GPU: the HIP path must match them within the
tolerances of tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

CASES = sorted(make_golden.CASES)


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    g = load(name)
    params, out = make_golden.build(name)
    for k in ("coeffs", "ac_strategy", "raw_quant", "epf_sharpness", "ytox_map", "ytob_map", "dc"):
        assert np.array_equal(out[k], g[k]), f"generator drifted: {k}"
    for k in ("xyb", "sigma", "rgb"):
        assert np.array_equal(out[k].view(np.uint32), g[k].view(np.uint32)), k


@pytest.mark.parametrize("name", CASES)
def test_oracle_packed_output_reproduces_golden(oracle, name):
    import frames
    g = load(name)
    xs, ys, kw = make_golden.CASES[name]
    _, _, fr = frames.make_case(xs, ys, **dict(kw, output_kind=2, out_format=make_golden.SRGB8))
    assert np.array_equal(fr.decode(threads=1), g["srgb8"])


@pytest.mark.parametrize("name", CASES)
def test_entropy_decoder_on_golden_streams(name):
    """f1 without the reference library: the committed AC streams (written by the
    reference encoder when the fixtures were made) decode to the committed coefficients."""
    import ctypes as C
    from libjxl_amd import abi
    g = load(name)
    xs, ys, _ = make_golden.CASES[name]
    L = abi.load_library()
    glob = np.ascontiguousarray(g["ac_global"])
    pos, h = C.c_size_t(0), C.c_void_p()
    assert L.jxlhip_ac_pass_decode(glob.ctypes.data, len(glob), C.byref(pos), int(g["ac_used_acs"][0]), 1, None,
                                   C.byref(h)) == 0
    acs, rq = np.ascontiguousarray(g["ac_strategy"]), np.ascontiguousarray(g["raw_quant"])
    xsb, ysb, xsg = (xs + 7) // 8, (ys + 7) // 8, (xs + 255) // 256
    offs, data = g["ac_offsets"], np.ascontiguousarray(g["ac_groups"])
    out = [np.zeros_like(g["coeffs"][c]) for c in range(3)]
    try:
        for gi in range(len(offs) - 1):
            d = np.ascontiguousarray(data[offs[gi]:offs[gi + 1]])
            gp = C.c_size_t(0)
            ptrs = (C.c_void_p * 3)(*[o[gi * 65536:].ctypes.data for o in out])
            assert L.jxlhip_ac_group_decode(h, xsb, ysb, gi % xsg, gi // xsg, acs.ctypes.data, rq.ctypes.data, None,
                                            d.ctypes.data, len(d), C.byref(gp), 0, 0, ptrs, None) == 0
    finally:
        L.jxlhip_ac_pass_destroy(h)
    for c in range(3):
        assert np.array_equal(out[c], g["coeffs"][c]), c


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_packed_output_matches_golden(name):
    import torch
    from libjxl_amd import VarDctDecoder, synth
    g = load(name)
    xs, ys, kw = make_golden.CASES[name]
    params, _ = synth.synth_frame(8, 8, mix=synth.MIX_DCT8, output_kind=2, out_format=make_golden.SRGB8,
                                  **{k: v for k, v in kw.items() if k != "mix"})
    params["xsize"], params["ysize"] = xs, ys
    params["used_acs"] = 0  # the strategies of the stored frame are not those of the 8x8 stand-in
    dec = VarDctDecoder(0)
    dec.begin_frame(params)
    dq = dec.default_dequant_tables()
    t = dict(coeffs=[torch.from_numpy(g["coeffs"][c]).cuda() for c in range(3)],
             ac_strategy=torch.from_numpy(g["ac_strategy"]).cuda(),
             raw_quant=torch.from_numpy(g["raw_quant"]).cuda(),
             epf_sharpness=torch.from_numpy(g["epf_sharpness"]).cuda(),
             ytox_map=torch.from_numpy(g["ytox_map"]).cuda(),
             ytob_map=torch.from_numpy(g["ytob_map"]).cuda(),
             dc=[torch.from_numpy(g["dc"][c]).cuda() for c in range(3)])
    dec.set_inputs(t, dq)
    out = dec.decode_frame().cpu().numpy()
    dec.sync()
    d = np.abs(out.astype(np.int32) - g["srgb8"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 5e-3
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_golden(name):
    import torch
    from libjxl_amd import VarDctDecoder
    g = load(name)
    xs, ys, kw = make_golden.CASES[name]
    import frames
    from libjxl_amd import synth
    params, _ = synth.synth_frame(8, 8, mix=synth.MIX_DCT8, **{k: v for k, v in kw.items() if k != "mix"})
    params["xsize"], params["ysize"] = xs, ys
    params["used_acs"] = 0  # the strategies of the stored frame are not those of the 8x8 stand-in
    dec = VarDctDecoder(0)
    dec.begin_frame(params)
    dq = dec.default_dequant_tables()
    t = dict(coeffs=[torch.from_numpy(g["coeffs"][c]).cuda() for c in range(3)],
             ac_strategy=torch.from_numpy(g["ac_strategy"]).cuda(),
             raw_quant=torch.from_numpy(g["raw_quant"]).cuda(),
             epf_sharpness=torch.from_numpy(g["epf_sharpness"]).cuda(),
             ytox_map=torch.from_numpy(g["ytox_map"]).cuda(),
             ytob_map=torch.from_numpy(g["ytob_map"]).cuda(),
             dc=[torch.from_numpy(g["dc"][c]).cuda() for c in range(3)])
    dec.set_inputs(t, dq)
    out = dec.decode_frame()
    dec.sync()
    ref = g["rgb"]
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(out.cpu().numpy() - ref).max()) / scale <= 2e-5
    # the phase-1 tap needs the split call: jxlhip_decode_frame may run fused (DCT8 never reaches the planes)
    dec.decode_blocks()
    got = np.stack(dec.export_xyb())
    assert float(np.abs(got - g["xyb"]).max()) <= 2e-5 * max(1.0, float(np.abs(g["xyb"]).max()))
    dec.close()
