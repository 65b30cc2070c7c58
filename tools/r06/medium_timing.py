#!/usr/bin/env python3
"""GPU box, experiment build -DJXLHIP_MEDIUM_TIMING (tools/r06/_variants/libjxl_hip_mtiming.so through JXLHIP_SO): where does a
64x64 varblock's time go inside MediumUnit?  Every wave's lane 0 adds its shader-clock ticks per phase to the first words
of the frame's inv_sigma table; read back through jxlhip_get_sigma."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from libjxl_amd import VarDctDecoder, synth  # noqa: E402

for mix, used in (("18:1", True), ("real4k", False)):
    # (no EPF: k_prepare then leaves the inv_sigma table alone, and the counters can be zeroed in place)
    params, t = synth.synth_frame(7680, 4320, mix=bench.resolve_mix(mix), gab=False, epf_iters=0, device="cuda:0")
    if not used:
        params["used_acs"] = 0
    os.environ["JXLHIP_FUSE"] = "0"
    dec = VarDctDecoder(0)
    dec.begin_frame(params)
    dq = dec.default_dequant_tables()
    dec.set_inputs(t, dq)
    import ctypes as C
    from libjxl_amd.decoder import _as_tensor
    for _ in range(3):
        dec.decode_blocks()
    dec.sync()
    ptr, stride = C.c_void_p(), C.c_size_t()
    assert dec.L.jxlhip_get_sigma(dec.ctx, C.byref(ptr), C.byref(stride)) == 0
    view = _as_tensor(ptr.value, 16, torch.float32, 0)  # the table's first 64 bytes, in place
    view.zero_()
    torch.cuda.synchronize()
    dec.decode_blocks()
    dec.sync()
    raw = view.view(torch.int64).cpu().tolist()
    n = raw[7]
    names = ["loads + dequant", "barrier wait", "LLF", "pass 1", "pass 2", "stores issued"]
    print(f"mix {mix} ({'class kernel' if used else 'merged launch'}): {n} wave-blocks in the last decode")
    tot = sum(raw[:6])
    for k, name in enumerate(names):
        print(f"  {name:18s} {raw[k] / max(n, 1):9.0f} ticks per wave and varblock  {100.0 * raw[k] / max(tot, 1):5.1f} %")
    print(f"  {'sum':18s} {tot / max(n, 1):9.0f} ticks = {tot / max(n, 1) / 2.4e3:.2f} us at 2.4 GHz")
    dec.close()
