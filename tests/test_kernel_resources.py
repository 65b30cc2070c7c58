"""Register / scratch budget of the shipped gfx950 kernels, read from the code-object metadata inside libjxl_hip.so
(the AMDGPU msgpack notes: .symbol, .vgpr_count, .vgpr_spill_count, .private_segment_fixed_size).

k_fused_pc's producing wave prefetches through INLINE-ASM loads whose only wait is the barrier's vmcnt(0)
(kernels_fused.hip): the compiler believes those registers hold their values from the asm statement on, so a spill
(or a scratch copy) of one of them between the load and the barrier would store a value that has not arrived yet --
an ablation build capped at 128 VGPRs did exactly that and faulted.  The shipped kernels must therefore have no
scratch at all."""
import os
import re

import pytest

from libjxl_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _uint(b, i):
    t = b[i]
    if t <= 0x7F:
        return t
    if t == 0xCC:
        return b[i + 1]
    if t == 0xCD:
        return int.from_bytes(b[i + 1:i + 3], "big")
    if t == 0xCE:
        return int.from_bytes(b[i + 1:i + 5], "big")
    raise ValueError(hex(t))


def kernels():
    so = os.path.join(ROOT, "libjxl_amd", "csrc", "libjxl_hip.so")
    b = open(so, "rb").read()
    out = {}
    # the keys of a kernel record are sorted: .private_segment_fixed_size ... .symbol ... .vgpr_count .vgpr_spill_count
    for m in re.finditer(rb"\xbb\.private_segment_fixed_size", b):
        scratch = _uint(b, m.end())
        s = b.find(b"\xa7.symbol", m.end(), m.end() + 400)
        if s < 0:
            continue
        t = b[s + 8]
        if t == 0xD9:
            n, at = b[s + 9], s + 10
        elif t == 0xDA:
            n, at = int.from_bytes(b[s + 9:s + 11], "big"), s + 11
        else:
            n, at = t & 0x1F, s + 9
        name = b[at:at + n].decode()
        v = b.find(b"\xab.vgpr_count", at, at + 600)
        sp = b.find(b"\xb1.vgpr_spill_count", at, at + 700)
        out[name] = dict(scratch=scratch, vgprs=_uint(b, v + 12), spills=_uint(b, sp + 18))
    return out


def test_fused_pc_kernels_have_no_scratch():
    abi.load_library()
    ks = kernels()
    pc = {k: v for k, v in ks.items() if "k_fused_pc" in k}
    assert len(pc) >= 12, sorted(ks)[:5]  # 6 stage lists x 2 outputs x 2 coefficient types
    bad = {k: v for k, v in pc.items() if v["scratch"] or v["spills"]}
    assert not bad, bad
    assert max(v["vgprs"] for v in pc.values()) <= 168  # three waves per SIMD


def test_metadata_reader_sees_the_known_kernels():
    ks = kernels()
    names = " ".join(ks)
    for k in ("k_prepare", "k_transform_r", "k_filters_fast", "k_fused", "k_epf0", "k_transform_mfma32", "k_transform_mfma16"):
        assert k in names
    assert all(1 <= v["vgprs"] <= 512 for v in ks.values())
