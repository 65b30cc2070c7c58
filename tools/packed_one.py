"""GPU box: one packed format, 8K d1.0, a few frames (for rocprofv3). usage: packed_one.py st bits nc"""
import sys, torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
st, bits, nc = map(int, sys.argv[1:4])
params, t = synth.synth_frame(7680, 4320, device="cuda", output_kind=2, intensity_target=80.0,
                              out_format=dict(transfer=1, sample_type=st, num_channels=nc, bits_per_sample=bits))
dec = VarDctDecoder(0); dec.begin_frame(params); dq = dec.default_dequant_tables(); dec.set_inputs(t, dq); out = dec.alloc_output()
for _ in range(6): dec.decode_frame(out)
dec.sync(); torch.cuda.synchronize()
