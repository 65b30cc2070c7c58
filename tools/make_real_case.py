"""Build container only (needs oracle/_ref with /root/reference): writes a genuine libjxl VarDCT
stream + the side info of the product's boundary to an .npz for tools/e2e_real.py.
usage: python tools/make_real_case.py out.npz [xsize ysize distance speed_tier]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402

out = sys.argv[1]
xs, ys = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
dist = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
tier = int(sys.argv[5]) if len(sys.argv) > 5 else 5
oracle.ref_lib()
rs = oracle.RealStream(xs, ys, seed=7, distance=dist, speed_tier=tier)
sub = rs.rgb[::8, ::8].astype(np.float16)
np.savez_compressed(
    out, xsize=xs, ysize=ys, codestream=rs.codestream, section_offset=rs.section_offset,
    section_size=rs.section_size, num_groups=rs.num_groups, num_dc_groups=rs.num_dc_groups,
    num_passes=rs.num_passes, shift=np.array(rs.shift, np.uint32), used_acs=rs.used_acs,
    num_histograms=rs.num_histograms, params=rs.params, ac_strategy=rs.ac_strategy, raw_quant=rs.raw_quant,
    epf_sharpness=rs.epf_sharpness, ytox_map=rs.ytox_map, ytob_map=rs.ytob_map, dc_x=rs.dc_x, dc_y=rs.dc_y,
    dc_b=rs.dc_b, quant_dc=rs.quant_dc, block_ctx_bytes=rs.block_ctx_bytes, rgb_sub8=sub)
print("wrote", out, "codestream bytes", len(rs.codestream), "groups", rs.num_groups, "epf", rs.frame_params.lf.epf_iters,
      "gab", rs.frame_params.lf.gab, "strategies", np.bincount(rs.ac_strategy.ravel() >> 1, minlength=27).tolist())
