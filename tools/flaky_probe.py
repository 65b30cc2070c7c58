"""Determinism probe: decode the same frame repeatedly and compare with the first
result (bitwise).  python tools/flaky_probe.py [epf_iters] [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth  # noqa: E402

epf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for name, kw in [("rgb_f32", dict(output_kind=1)),
                 ("srgb_u8_rgba", dict(output_kind=2, out_format=dict(transfer=1, sample_type=1, num_channels=4,
                                                                      bits_per_sample=8))),
                 ("srgb_u8_rgb", dict(output_kind=2, out_format=dict(transfer=1, sample_type=1, num_channels=3,
                                                                     bits_per_sample=8)))]:
    bad = 0
    first = None
    for r in range(reps):
        d = VarDctDecoder(0)
        params, t = synth.synth_frame(533, 401, device="cuda", mix=synth.MIX_ALL, gab=True, epf_iters=epf,
                                      intensity_target=80.0, **kw)
        d.begin_frame(params)
        dq = d.default_dequant_tables()
        d.set_inputs(t, dq)
        out = d.decode_frame()
        d.sync()
        o = out.clone()
        if first is None:
            first = o
        elif not torch.equal(first, o):
            bad += 1
            diff = (first.to(torch.float32) - o.to(torch.float32)).abs()
            nz = diff.nonzero()
            print(name, "run", r, "differs at", int((diff > 0).sum()), "samples; first", nz[0].tolist(), "last",
                  nz[-1].tolist(), "max", float(diff.max()))
        d.close()
    print(name, "epf", epf, ":", bad, "of", reps - 1, "repeat runs differ")
