"""GPU box: ms/frame of the 8K d1.0 workload for the packed output formats (f3)."""
import sys, time, torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
for st, bits, nc in ((1, 8, 4), (1, 8, 3), (2, 16, 3), (2, 16, 4), (3, 0, 4), (0, 0, 3)):
    params, t = synth.synth_frame(7680, 4320, device="cuda", output_kind=2, intensity_target=80.0,
                                  out_format=dict(transfer=1, sample_type=st, num_channels=nc, bits_per_sample=bits))
    dec = VarDctDecoder(0); dec.begin_frame(params); dq = dec.default_dequant_tables(); dec.set_inputs(t, dq); out = dec.alloc_output()
    for _ in range(3): dec.decode_frame(out)
    dec.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dec.decode_frame(out)
    dec.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("8K d1.0 -> sRGB %s x%d: %.3f ms/frame = %.1f Gpx/s" % (["f32", "u8", "u16", "f16"][st], nc, dt * 1e3, 7680 * 4320 / dt / 1e9))
    dec.close()
