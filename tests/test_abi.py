"""CPU-side checks of the drop-in boundary: libjxl_hip.so loads without a GPU,
exports every symbol include/jxl_hip.h declares, its static geometry helpers
agree with the oracle's restatement of lib/jxl/ac_strategy.h:148-173 and
lib/jxl/quant_weights.h:337-367, and device entry points fail loudly (no CPU
fallback) when there is no device."""
import ctypes as C
import os
import re

import pytest

from libjxl_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from libjxl_amd import build
    build.build()
    return abi.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jxl_hip.h")).read()
    text += open(os.path.join(ROOT, "include", "jxl_hip_entropy.h")).read()
    text += open(os.path.join(ROOT, "include", "jxl_hip_frame.h")).read()
    text += open(os.path.join(ROOT, "include", "jxl_hip_codestream.h")).read()
    return sorted(set(re.findall(r"JXLHIP_EXPORT[^;{]*?\b(jxlhip_\w+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 26
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/jxl_hip.h but not exported"
    assert sorted(abi.EXPORTS) == syms


def test_geometry_helpers_match_oracle(lib, oracle):
    L = oracle.lib()
    for s in range(27):
        assert lib.jxlhip_covered_blocks_x(s) == L.jxo_covered_blocks_x(s)
        assert lib.jxlhip_covered_blocks_y(s) == L.jxo_covered_blocks_y(s)
        assert lib.jxlhip_log2_covered_blocks(s) == L.jxo_log2_covered_blocks(s)
        assert lib.jxlhip_quant_table_of_strategy(s) == L.jxo_quant_table_of_strategy(s)
        for c in range(3):
            assert lib.jxlhip_dequant_table_offset(s, c) == L.jxo_dequant_table_offset(s, c)
    assert lib.jxlhip_covered_blocks_x(27) == 0
    assert lib.jxlhip_quant_table_of_strategy(-1) == -1


def test_struct_layouts_agree_with_oracle(oracle):
    assert C.sizeof(abi.FrameParams) == C.sizeof(oracle.FrameParams)
    for (n1, t1), (n2, t2) in zip(abi.FrameParams._fields_, oracle.FrameParams._fields_):
        assert n1 == n2
        assert getattr(abi.FrameParams, n1).offset == getattr(oracle.FrameParams, n2).offset


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = lib.jxlhip_create(0, C.byref(ctx))
    assert rc == -2 and not ctx.value  # JXLHIP_ERR_NO_DEVICE
    assert b"no HIP device" in lib.jxlhip_status_string(rc)
    from libjxl_amd import VarDctDecoder
    with pytest.raises(abi.JxlHipError):
        VarDctDecoder(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under libjxl_amd/ may use it."""
    pkg = os.path.join(ROOT, "libjxl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert "jxl_oracle.h" not in text and "libjxl_oracle" not in text, f
