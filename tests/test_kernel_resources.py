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


def kernels():
    from libjxl_amd import build
    return build.kernel_resources(os.path.join(ROOT, "libjxl_amd", "csrc", "libjxl_hip.so"))


def test_fused_pc_kernels_have_no_scratch():
    abi.load_library()
    ks = kernels()
    pc = {k: v for k, v in ks.items() if "k_fused_pc" in k}
    assert len(pc) >= 12, sorted(ks)[:5]  # 6 stage lists x 2 outputs x 2 coefficient types
    bad = {k: v for k, v in pc.items() if v["scratch"] or v["spills"]}
    assert not bad, bad
    # three waves per SIMD -- except k_fused_pc0 with Gaborish (epf_iters = 3: the EPF0 window), built for two
    assert max(v["vgprs"] for k, v in pc.items() if "k_fused_pc0" not in k) <= 168
    assert max(v["vgprs"] for k, v in pc.items() if "k_fused_pc0" in k) <= 256
    assert any("k_fused_pc0" in k for k in pc)


def test_a_build_with_scratch_in_the_fused_kernels_is_refused(tmp_path):
    """libjxl_amd/build.py checks the same metadata at build time (check_no_scratch) and deletes a library that fails:
    here against a copy of the shipped library and a pattern that does have scratch (k_transform_r<short> keeps its
    cold-path spills on purpose)."""
    import shutil
    from libjxl_amd import build
    so = str(tmp_path / "copy.so")
    shutil.copy(os.path.join(ROOT, "libjxl_amd", "csrc", "libjxl_hip.so"), so)
    build.check_no_scratch(so)  # the shipped fused kernels: clean
    assert os.path.exists(so)
    if any(v["scratch"] for v in build.kernel_resources(so).values()):
        spilling = next(k for k, v in build.kernel_resources(so).items() if v["scratch"])
        with pytest.raises(RuntimeError, match="refusing to ship"):
            build.check_no_scratch(so, pattern=spilling)
        assert not os.path.exists(so)


def test_metadata_reader_sees_the_known_kernels():
    ks = kernels()
    names = " ".join(ks)
    for k in ("k_prepare", "k_transform_r", "k_filters_fast", "k_fused", "k_epf0", "k_transform_mfma32", "k_transform_mfma16"):
        assert k in names
    assert all(1 <= v["vgprs"] <= 512 for v in ks.values())
