import os, subprocess, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle as O
from test_djxl import read_pam
for kw, out in ((dict(xsize=200, ysize=120, seed=12, distance=1.0, speed_tier=3, alpha_bits=8, original="srgb8"), "pam"),
                (dict(xsize=200, ysize=120, seed=12, distance=1.0, speed_tier=3, original="srgb8"), "ppm")):
    rs = O.RealStream(**kw)
    open("/tmp/rgba.jxl", "wb").write(rs.codestream.tobytes())
    r = subprocess.run(["oracle/_ref/djxl_hip", "/tmp/rgba.jxl", "/tmp/hip." + out], capture_output=True, text=True, env=dict(os.environ, JXLHIP_SEAM_VERBOSE="1"))
    print(r.stderr)
