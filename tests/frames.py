"""Shared test helpers: synthetic frames -> (product inputs, oracle frame)."""
import ctypes as C

import numpy as np

from libjxl_amd import abi, synth


def default_params(xsize, ysize, **kw):
    """abi.FrameParams with format defaults, viewed as the oracle's struct."""
    import oracle
    p, _ = synth.synth_frame(8, 8, mix=synth.MIX_DCT8, **kw)
    p["xsize"], p["ysize"] = xsize, ysize
    return to_oracle_params(abi.make_params(p))


def to_oracle_params(p):
    import oracle
    assert C.sizeof(abi.FrameParams) == C.sizeof(oracle.FrameParams)
    q = oracle.FrameParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
    return q


def make_case(xsize, ysize, dequant=None, **kw):
    """Returns (params_dict, torch tensors on CPU, oracle.Frame)."""
    import oracle
    params, t = synth.synth_frame(xsize, ysize, device="cpu", **kw)
    if dequant is None:
        dequant = oracle.default_dequant_tables()
    npy = dict(
        coeffs=[c.numpy() for c in t["coeffs"]],
        ac_strategy=t["ac_strategy"].numpy(), raw_quant=t["raw_quant"].numpy(),
        epf_sharpness=t["epf_sharpness"].numpy(), ytox_map=t["ytox_map"].numpy(),
        ytob_map=t["ytob_map"].numpy(), dc=[d.numpy() for d in t["dc"]])
    fr = oracle.Frame(to_oracle_params(abi.make_params(params)), npy["coeffs"],
                      npy["ac_strategy"], npy["raw_quant"], npy["epf_sharpness"],
                      npy["ytox_map"], npy["ytob_map"], npy["dc"], dequant)
    return params, t, fr

