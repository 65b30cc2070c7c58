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
    xsb, ysb = (xs + 7) // 8, (ys + 7) // 8
    got = np.stack(dec.export_xyb())
    assert float(np.abs(got - g["xyb"]).max()) <= 2e-5 * max(1.0, float(np.abs(g["xyb"]).max()))
    dec.close()
