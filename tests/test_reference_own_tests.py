"""The libjxl reference's OWN unit tests of the VarDCT hot path -- lib/jxl/dct_test.cc, ac_strategy_test.cc,
quant_weights_test.cc, opsin_inverse_test.cc -- compiled in place from /root/reference and run against the Highway
stand-ins the checker (one lane) and the CPU baseline (eight lanes) rest on (oracle/build_ref_tests.py; googletest
replaced by oracle/gtest_shim).  The checker is checked by the reference's known-answer tests, not only by restatements
of them (tests/test_oracle_kat.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

EXPECTED = {"dct_test": 133, "ac_strategy_test": 84, "quant_weights_test": 16, "opsin_inverse_test": 2}


@pytest.fixture(scope="module")
def bins():
    import build_ref_tests
    try:
        return build_ref_tests.build()
    except RuntimeError as e:
        pytest.skip(str(e)[:200])


@pytest.mark.parametrize("lanes", [1, 8])
@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_reference_unit_tests_pass_on_the_highway_stand_in(bins, name, lanes):
    import build_ref_tests
    if lanes == 8:
        flags = open("/proc/cpuinfo").read()
        if " avx2" not in flags or " fma" not in flags:
            pytest.skip("host CPU without AVX2 / FMA")
    rc, out, err = build_ref_tests.run(bins[(name, lanes)])
    last = out.strip().splitlines()[-1] if out.strip() else ""
    assert rc == 0, (last, err[-1500:])
    assert last == "[==========] %d tests ran, 0 failed, 0 skipped" % EXPECTED[name], last
