"""Synthetic VarDCT frames in the coefficient domain (no bitstream).

Produces exactly the device-side structures the hot path consumes
(include/jxl_hip.h: jxlhip_frame_params + jxlhip_frame_inputs), with the
statistics SURVEY.md 8(d) asks for: a "d1.0 e7-like" strategy mix, a smooth
raw_quant field, ~88% zero coefficients that fit int16, CfL factors in
[-20, 20], a smooth DC field.

The generator is geometry-driven: a small library of legal 256x256-group
layouts (varblocks placed greedily in raster order, never crossing the group or
the frame, dec_modular.cc:539-549) is built per distinct clipped group size,
then groups pick layouts at random; everything else is vectorised torch and
runs on the CPU or directly in HBM (device="cuda").
"""
import math

import numpy as np
import torch

COVERED_X = [1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1,
             8, 4, 8, 16, 8, 16, 32, 16, 32]
COVERED_Y = [1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1,
             8, 8, 4, 16, 16, 8, 32, 32, 16]

# area shares: 45% DCT8, 20% 16x8/8x16, 12% 16x16, 8% 32x16/16x32/
# 32x32, 5% small/special kinds, 10% 32x8/8x32/64x*
MIX_D1 = {0: 45, 6: 10, 7: 10, 4: 12, 10: 2.5, 11: 2.5, 5: 3, 1: 0.5, 2: 0.5,
          3: 0.7, 12: 0.7, 13: 0.7, 14: 0.5, 15: 0.5, 16: 0.5, 17: 0.4,
          8: 2.5, 9: 2.5, 18: 2, 19: 1.5, 20: 1.5}
MIX_ALL = {s: 1.0 for s in range(27)}          # every strategy, for parity tests
MIX_DCT8 = {0: 1.0}
MIX_DCT32 = {5: 1.0}                            # config 5: 32x32 forced


def _make_layout(gw, gh, mix, rng):
    """Greedy raster placement inside a gw x gh (in blocks) group.
    Returns acs bytes [gh, gw] ((raw<<1)|first) and the visit-ordered list of
    (by, bx, strategy)."""
    kinds = np.array(list(mix.keys()))
    # convert area share -> per-placement probability (share / area)
    w = np.array([mix[k] / (COVERED_X[k] * COVERED_Y[k]) for k in kinds], np.float64)
    w /= w.sum()
    acs = np.zeros((gh, gw), np.uint8)
    used = np.zeros((gh, gw), bool)
    blocks = []
    for by in range(gh):
        for bx in range(gw):
            if used[by, bx]:
                continue
            s = 0
            for _ in range(4):  # a few tries, then fall back to DCT8/any fit
                cand = int(kinds[rng.choice(len(kinds), p=w)])
                cx, cy = COVERED_X[cand], COVERED_Y[cand]
                if bx + cx <= gw and by + cy <= gh and not used[by:by + cy, bx:bx + cx].any():
                    s = cand
                    break
            else:
                fits = [int(k) for k in kinds
                        if COVERED_X[k] == 1 and COVERED_Y[k] == 1]
                s = fits[0] if fits else 0
            cx, cy = COVERED_X[s], COVERED_Y[s]
            if not (bx + cx <= gw and by + cy <= gh and not used[by:by + cy, bx:bx + cx].any()):
                s, cx, cy = 0, 1, 1
            used[by:by + cy, bx:bx + cx] = True
            acs[by:by + cy, bx:bx + cx] = s << 1
            acs[by, bx] = (s << 1) | 1
            blocks.append((by, bx, s))
    return acs, blocks


def _layout_streams(blocks, decay, amp):
    """Per-coefficient Laplace scale (0 in LLF slots / unused tail) of the
    group's 65536-entry coefficient stream, in visit order."""
    scale = np.zeros(65536, np.float32)
    off = 0
    for (_, _, s) in blocks:
        cx, cy = COVERED_X[s], COVERED_Y[s]
        lo, hi = min(cx, cy), max(cx, cy)
        rows, cols = 8 * lo, 8 * hi
        fy = np.arange(rows, dtype=np.float32)[:, None] / rows
        fx = np.arange(cols, dtype=np.float32)[None, :] / cols
        f = np.sqrt(fx * fx + fy * fy)
        sc = (amp * np.exp(-decay * f)).astype(np.float32)
        sc[:lo, :hi] = 0.0  # LLF slots hold 0 in the stream (dec_group.cc:469)
        n = rows * cols
        scale[off:off + n] = sc.reshape(-1)
        off += n
    assert off <= 65536
    return scale


def synth_frame(xsize, ysize, *, mix=None, gab=True, epf_iters=1, seed=0x4A584C,
                device="cpu", coeff_type=0, num_layouts=6, intensity_target=255.0,
                output_kind=1, quant_mul=1.0, decay=6.0, amp=6.0,
                custom_lf=False, out_format=None, undo_orientation=0):
    """Returns (params: dict of python scalars, tensors: dict of torch tensors).
    params follows jxlhip_frame_params; tensors follows jxlhip_frame_inputs."""
    mix = MIX_D1 if mix is None else mix
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) & 0x7FFFFFFF)
    xsb, ysb = (xsize + 7) // 8, (ysize + 7) // 8
    xsg, ysg = (xsize + 255) // 256, (ysize + 255) // 256
    xt, yt = (xsb + 7) // 8, (ysb + 7) // 8
    ngroups = xsg * ysg

    # ---- layout library per clipped group size
    def gdim(i, n, nb):
        return min(32, nb - 32 * i)
    sizes = {}
    for gy in range(ysg):
        for gx in range(xsg):
            sizes.setdefault((gdim(gx, xsg, xsb), gdim(gy, ysg, ysb)), []).append(gy * xsg + gx)
    acs = np.zeros((ysb, xsb), np.uint8)
    lay_scale = []
    group_layout = np.zeros(ngroups, np.int64)
    for (gw, gh), groups in sizes.items():
        base = len(lay_scale)
        lays = []
        for _ in range(min(num_layouts, len(groups))):
            a, blocks = _make_layout(gw, gh, mix, rng)
            lays.append(a)
            lay_scale.append(_layout_streams(blocks, decay, amp))
        pick = rng.integers(0, len(lays), size=len(groups))
        for gi, li in zip(groups, pick):
            gy, gx = divmod(gi, xsg)
            acs[gy * 32:gy * 32 + gh, gx * 32:gx * 32 + gw] = lays[li]
            group_layout[gi] = base + li
    lay_scale = torch.from_numpy(np.stack(lay_scale))          # [nlay, 65536]

    dev = torch.device(device)
    gl = torch.from_numpy(group_layout).to(dev)
    scale = lay_scale.to(dev)[gl]                               # [ngroups, 65536]
    gdev = torch.Generator(device=dev)
    gdev.manual_seed(int(seed) & 0x7FFFFFFF)
    dtype = torch.int16 if coeff_type == 0 else torch.int32
    coeffs = []
    for c in range(3):
        u = torch.rand(scale.shape, generator=gdev, device=dev) - 0.5
        # Laplace(0, b): -b*sign(u)*ln(1-2|u|); chroma channels are sparser
        b = scale * (1.0 if c == 1 else 0.45)
        lap = -b * torch.sign(u) * torch.log1p(-2.0 * u.abs().clamp(max=0.4999999))
        lim = 30000 if coeff_type == 0 else (1 << 20)
        q = torch.round(lap).clamp(-lim, lim)
        coeffs.append(q.to(dtype).reshape(-1).contiguous())
        del u, lap, q

    # ---- side info
    yy = torch.arange(ysb, dtype=torch.float32)[:, None]
    xx = torch.arange(xsb, dtype=torch.float32)[None, :]
    qf = 18.0 + 10.0 * torch.sin(xx * 0.05) * torch.cos(yy * 0.07) + \
        3.0 * torch.randn((ysb, xsb), generator=g)
    raw_quant = (qf * quant_mul).clamp(1, 256).to(torch.int32)
    sharp_p = torch.tensor([0.02, 0.02, 0.03, 0.05, 0.18, 0.25, 0.25, 0.2])
    sharp = torch.multinomial(sharp_p, ysb * xsb, replacement=True, generator=g).to(torch.uint8).reshape(ysb, xsb)
    ytox = torch.randint(-20, 21, (yt, xt), generator=g, dtype=torch.int32).to(torch.int8)
    ytob = torch.randint(-20, 21, (yt, xt), generator=g, dtype=torch.int32).to(torch.int8)
    # smooth DC (XYB-ish ranges) + a little texture
    dcy = 0.35 + 0.25 * torch.sin(xx * 0.021 + 0.3) * torch.sin(yy * 0.017) + \
        0.01 * torch.randn((ysb, xsb), generator=g)
    dcx = 0.01 * torch.cos(xx * 0.013) * torch.sin(yy * 0.029) + \
        0.001 * torch.randn((ysb, xsb), generator=g)
    dcb = 0.9 * dcy + 0.05 * torch.cos(xx * 0.011 + yy * 0.009) + \
        0.005 * torch.randn((ysb, xsb), generator=g)
    dc = [dcx.contiguous(), dcy.contiguous(), dcb.contiguous()]

    inv = [11.031566901960783, -9.866943921568629, -0.16462299647058826,
           -3.254147380392157, 4.418770392156863, -0.16462299647058826,
           -3.6588512862745097, 2.7129230470588235, 1.9459282392156863]
    f32 = np.float32
    mul = f32(255.0) / f32(intensity_target)
    lf_w1 = float(f32(1.1 * f32(0.104699568)))
    lf_w2 = float(f32(1.1 * f32(0.055680538)))
    gabw = [lf_w1, lf_w2] * 3
    sharp_lut = [i / 7.0 for i in range(8)]
    ch_scale = [40.0, 5.0, 3.5]
    qmul, p0, p2, bsm = 0.46, 0.9, 6.5, 0.6666666666666666
    if custom_lf:  # exercise non-default LoopFilter fields
        gabw = [0.12, 0.05, 0.10, 0.07, 0.09, 0.04]
        sharp_lut = [0.0, 0.1, 0.3, 0.45, 0.6, 0.7, 0.9, 1.0]
        ch_scale = [35.0, 6.0, 3.0]
        qmul, p0, p2, bsm = 0.5, 1.0, 6.0, 0.7
    params = dict(
        xsize=xsize, ysize=ysize, coeff_type=coeff_type, output_kind=output_kind,
        global_scale=5243, quant_dc=16,
        x_dm_multiplier=float(f32(math.pow(1 / 1.25, 3 - 2.0))),  # x_qm_scale 3
        b_dm_multiplier=float(f32(math.pow(1 / 1.25, 2 - 2.0))),  # b_qm_scale 2
        quant_biases=[1.0 - 0.05465007330715401, 1.0 - 0.07005449891748593,
                      1.0 - 0.049935103337343655, 0.145],
        cfl_base_x=0.0, cfl_base_b=1.0, cfl_color_factor=84,
        gab=int(gab), gab_weights=gabw, epf_iters=int(epf_iters),
        epf_sharp_lut=sharp_lut, epf_channel_scale=ch_scale,
        epf_quant_mul=qmul, epf_pass0_sigma_scale=p0, epf_pass2_sigma_scale=p2,
        epf_border_sad_mul=bsm,
        opsin_biases=[-0.0037930732552754493] * 3,
        inverse_opsin_matrix=[float(f32(f32(v) * mul)) for v in inv],
        stripe_group_y0=0, stripe_group_rows=0, out_format=out_format, undo_orientation=undo_orientation,
        used_acs=int(np.bitwise_or.reduce(1 << (np.unique(acs[(acs & 1) == 1]) >> 1).astype(np.int64))))
    tensors = dict(
        coeffs=coeffs,
        ac_strategy=torch.from_numpy(acs).to(dev),
        raw_quant=raw_quant.contiguous().to(dev),
        epf_sharpness=sharp.contiguous().to(dev),
        ytox_map=ytox.contiguous().to(dev), ytob_map=ytob.contiguous().to(dev),
        dc=[d.to(dev) for d in dc])
    return params, tensors
