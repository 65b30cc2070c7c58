#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded VarDCT frames (inputs in the
jxlhip_frame_inputs layout) together with the outputs of the libjxl REFERENCE
run in this container (oracle/_ref: lib/jxl's decoder sources compiled in place,
oracle/build_ref.py + oracle/ref_driver.cc): `rgb` = final linear RGB of the
full path, `xyb` = the planes after dequant + inverse transforms (the reference
run with the filters off and XYB output), `srgb8` = the same frame through the
reference's FromLinearStage (sRGB) + WriteToOutputStage as RGBA8, and
`ac_global` / `ac_groups` / `ac_offsets` / `ac_used_acs` = the frame's AC
coefficients as entropy-coded by the reference's own encoder (f1 test vectors).  `sigma` comes from the oracle
(ComputeSigma has no output tap in the reference; it is covered through `rgb`).
main() also asserts that the CPU oracle reproduces the reference bit for bit.
Re-run only on purpose:  python tests/golden/make_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import frames  # noqa: E402
from libjxl_amd import synth  # noqa: E402

CASES = {
    # name: (xsize, ysize, kwargs)
    "small_d1_gab_epf1": (96, 72, dict(mix={0: 30, 6: 10, 7: 10, 4: 10, 5: 10, 1: 2, 2: 2, 3: 2,
                                             12: 2, 13: 2, 14: 1, 15: 1, 16: 1, 17: 1},
                                       gab=True, epf_iters=1, seed=101)),
    "small_all_filters": (72, 56, dict(mix=synth.MIX_D1, gab=True, epf_iters=3, seed=102,
                                       custom_lf=True)),
    "tiny_3x8": (3, 8, dict(mix=synth.MIX_DCT8, gab=True, epf_iters=2, seed=103)),
}


def build(name):
    xs, ys, kw = CASES[name]
    params, t, fr = frames.make_case(xs, ys, **kw)
    planes = fr.decode_groups()
    out = dict(
        coeffs=np.stack([c.numpy() for c in t["coeffs"]]),
        ac_strategy=t["ac_strategy"].numpy(), raw_quant=t["raw_quant"].numpy(),
        epf_sharpness=t["epf_sharpness"].numpy(), ytox_map=t["ytox_map"].numpy(),
        ytob_map=t["ytob_map"].numpy(), dc=np.stack([d.numpy() for d in t["dc"]]),
        xyb=np.stack(planes), sigma=fr.compute_sigma(), rgb=fr.decode(threads=1))
    return params, out


SRGB8 = dict(transfer=1, sample_type=1, num_channels=4, bits_per_sample=8)


def reference_outputs(name):
    """rgb, (cropped) xyb planes, sRGB RGBA8 and AC entropy streams by the reference itself."""
    xs, ys, kw = CASES[name]
    _, _, fr = frames.make_case(xs, ys, **kw)
    rgb = fr.decode_ref(threads=1)
    kw1 = dict(kw, gab=False, epf_iters=0, output_kind=0)
    _, _, fr1 = frames.make_case(xs, ys, **kw1)
    _, _, fr2 = frames.make_case(xs, ys, **dict(kw, output_kind=2, out_format=SRGB8))
    srgb8 = fr2.decode_ref(threads=1)
    assert np.array_equal(srgb8, fr2.decode(threads=1)), "oracle != reference (srgb8)"
    glob, groups, used_acs, _ = fr.encode_ac_ref()
    ac = dict(ac_global=np.frombuffer(glob, np.uint8), ac_groups=np.frombuffer(b"".join(groups), np.uint8),
              ac_offsets=np.cumsum([0] + [len(g) for g in groups]).astype(np.int64),
              ac_used_acs=np.array([used_acs], np.uint32))
    return rgb, fr1.decode_ref(threads=1), srgb8, ac


def main():
    for name in CASES:
        params, out = build(name)
        xs, ys, _ = CASES[name]
        rgb, xyb, srgb8, ac = reference_outputs(name)
        assert np.array_equal(rgb.view(np.uint32), out["rgb"].view(np.uint32)), "oracle != reference (rgb)"
        assert np.array_equal(xyb.view(np.uint32), out["xyb"][:, :ys, :xs].view(np.uint32)), \
            "oracle != reference (xyb)"
        out["rgb"] = rgb  # the committed vector is the reference's own output
        out["srgb8"] = srgb8
        out.update(ac)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        h = hashlib.sha256(out["rgb"].tobytes()).hexdigest()[:16]
        print(name, {k: v.shape for k, v in out.items() if k in ("coeffs", "rgb")}, "rgb sha", h)


if __name__ == "__main__":
    main()
