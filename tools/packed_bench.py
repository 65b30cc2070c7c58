"""GPU box: ms/frame of the 8K d1.0 workload (gab + EPF1) for the packed output formats (f3)."""
import sys, time, torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
TF = {0: "linear", 1: "sRGB", 2: "PQ"}
ST = ["f32", "u8", "u16", "f16"]
CASES = [(1, 1, 8, 4, 0), (1, 1, 8, 3, 0), (1, 2, 16, 3, 0), (1, 2, 16, 4, 0), (1, 2, 16, 3, 1), (1, 2, 16, 4, 1), (1, 0, 0, 3, 0),
         (1, 0, 0, 4, 0), (0, 0, 0, 4, 0), (1, 3, 0, 4, 0), (0, 3, 0, 4, 0), (2, 2, 16, 3, 1), (2, 2, 16, 4, 1),
         (2, 2, 16, 3, 0), (0, 2, 16, 3, 0), (1, 3, 0, 3, 0)]   # the last three: no kernel of their own (general format)
for tf, st, bits, nc, sw in CASES:
    params, t = synth.synth_frame(7680, 4320, device="cuda", output_kind=2, intensity_target=80.0 if tf != 2 else 1000.0,
                                  out_format=dict(transfer=tf, sample_type=st, num_channels=nc, bits_per_sample=bits,
                                                  swap_endianness=sw, tf_param=1000.0 if tf == 2 else 0.0))
    dec = VarDctDecoder(0); dec.begin_frame(params); dq = dec.default_dequant_tables(); dec.set_inputs(t, dq); out = dec.alloc_output()
    for _ in range(3): dec.decode_frame(out)
    dec.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dec.decode_frame(out)
    dec.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("8K d1.0 -> %-6s %s x%d%s: %.3f ms/frame = %.1f Gpx/s" % (TF[tf], ST[st], nc, " big-endian" if sw else "", dt * 1e3, 7680 * 4320 / dt / 1e9), flush=True)
    dec.close()
