"""GPU box: ms/frame of the 8K d1.0 workload -> 8-bit sRGB RGBA (experiment helper, see tools/packed_bench.py)."""
import sys, time, torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
params, t = synth.synth_frame(7680, 4320, device="cuda", output_kind=2, intensity_target=80.0,
                              out_format=dict(transfer=1, sample_type=1, num_channels=4, bits_per_sample=8))
dec = VarDctDecoder(0); dec.begin_frame(params); dq = dec.default_dequant_tables(); dec.set_inputs(t, dq); out = dec.alloc_output()
for _ in range(3): dec.decode_frame(out)
dec.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): dec.decode_frame(out)
dec.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("8K d1.0 -> sRGB u8 x4: %.3f ms/frame = %.1f Gpx/s" % (dt * 1e3, 7680 * 4320 / dt / 1e9))
