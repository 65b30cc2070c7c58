"""Pins the CPU oracle against the reference's own fixture-free known-answer
tests.  Each test names the reference test it re-states; the
tolerances are the reference's.
"""
import ctypes as C

import numpy as np
import pytest

STRATS = list(range(27))
# the reference instantiates these for all 27; the 128/256 kinds are slow in a
# scalar oracle, so impulse counts are reduced for them (not the tolerances).


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --- lib/jxl/dct_test.cc:191-214,445-454 TestIdctAccuracyShard --------------
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256])
def test_idct_accuracy_vs_f64_matrix(oracle, n):
    L = oracle.lib()
    worst = 0.0
    for i in range(n):
        x = np.zeros(n, np.float32)
        x[i] = 1.0
        fast = np.zeros(n, np.float32)
        L.jxo_idct1d(n, _p(x), 1, _p(fast), 1)
        slow = np.zeros(n, np.float64)
        L.jxo_idct1d_slow(n, _p(x.astype(np.float64)), _p(slow))
        worst = max(worst, np.abs(fast - slow).max())
    assert worst <= 1e-7 * n + (0 if n <= 32 else 2e-7 * n), worst


# --- dct_test.cc:165-189 TestDctAccuracyShard (1.1e-7/N after 1/N scaling) --
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_dct_accuracy_vs_f64_matrix(oracle, n):
    L = oracle.lib()
    for i in range(n):
        x = np.zeros(n, np.float32)
        x[i] = 1.0
        fast = x.copy()
        L.jxo_dct1d(n, _p(fast), 1)
        fast /= n
        slow = np.zeros(n, np.float64)
        L.jxo_dct1d_slow(n, _p(x.astype(np.float64)), _p(slow))
        assert np.abs(fast - slow).max() <= 1.1e-7 * max(1, 8 / n) + 1e-7


# --- dct_test.cc:251-300 TestSlowInverse / DCT(IDCT) = I, 1e-5 --------------
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32])
def test_dct_idct_identity(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    y = np.zeros(n, np.float32)
    L.jxo_idct1d(n, _p(x), 1, _p(y), 1)
    L.jxo_dct1d(n, _p(y), 1)
    assert np.abs(y / n - x).max() < 1e-5


# --- dct_test.cc:314-432 TransposeTest.TestRectInverse/TestRectTranspose ----
RECT = [(1, 2), (2, 1), (2, 4), (4, 2), (4, 4), (4, 8), (8, 4), (8, 8),
        (8, 16), (16, 8), (16, 16), (8, 32), (32, 8), (16, 32), (32, 16),
        (32, 32)]


@pytest.mark.parametrize("r,c", RECT)
def test_rect_scaled_dct_inverse_and_transpose(oracle, r, c):
    L = oracle.lib()
    rng = np.random.default_rng(r * 100 + c)
    px = rng.random((r, c)).astype(np.float32)
    co = np.zeros(r * c, np.float32)
    L.jxo_scaled_dct(r, c, _p(px), c, _p(co))
    # DC is the mean (forward carries 1/N per axis, dct-inl.h:148-155)
    assert abs(co[0] - px.mean()) < 1e-6
    back = np.zeros((r, c), np.float32)
    L.jxo_scaled_idct(r, c, _p(co.copy()), _p(back), c)
    assert np.abs(back - px).max() < 1e-6 * 4
    # transposed block <-> swapped dims give the same coefficient matrix
    co_t = np.zeros(r * c, np.float32)
    pxt = np.ascontiguousarray(px.T)
    L.jxo_scaled_dct(c, r, _p(pxt), r, _p(co_t))
    if r != c:
        # both are stored short x long; R<C stores F[u][v], R>C stores F'[v][u]
        assert np.abs(co - co_t).max() < 3e-6
    else:
        assert np.abs(co.reshape(r, c) - co_t.reshape(r, c).T).max() < 3e-6
    # against the separable f64 matrix definition
    def M(n):
        k = np.arange(n)
        m = np.sqrt(2.0) * np.cos((k[:, None] + 0.5) * k[None, :] * np.pi / n)
        m[:, 0] = 1.0
        return m  # m[y,u]
    F = M(r).T @ px.astype(np.float64) @ M(c) / (r * c)  # F[u][v]
    got = co.reshape(min(r, c), max(r, c))
    want = F if r < c else F.T
    assert np.abs(got - want).max() < 1e-6


# --- ac_strategy_test.cc:28-92 AcStrategyRoundtrip --------------------------
@pytest.mark.parametrize("s", STRATS)
def test_ac_strategy_roundtrip(oracle, s):
    cx, cy = oracle.covered_blocks(s)
    n = 64 * cx * cy
    rng = np.random.default_rng(s * 65537 + 13)
    reps = 64 if n <= 4096 else 4
    for j in range(reps):
        i = int(rng.integers(0, n)) if n > 64 else j
        px = np.zeros(n, np.float32)
        px[i] = 0.2
        co = oracle.transform_from_pixels(s, px.reshape(8 * cy, 8 * cx))
        assert abs(co[0] - 0.2 / n) < 1e-6
        back = oracle.transform_to_pixels(s, co).reshape(-1)
        assert np.abs(back - px).max() < 2e-6, (s, i)
    # DC <-> LLF round trip, 1e-6
    positions = [(y, x) for y in range(cy) for x in range(cx)]
    if len(positions) > 64:
        positions = [positions[k] for k in rng.choice(len(positions), 64, replace=False)]
    for (y, x) in positions:
        dc = np.zeros((cy, cx), np.float32)
        dc[y, x] = 0.2
        llf = oracle.llf_from_dc(s, dc)
        back = oracle.dc_from_llf(s, llf)
        assert np.abs(back - dc).max() < 1e-6


# --- ac_strategy_test.cc:99-153 AcStrategyRoundtripDownsample ----------------
@pytest.mark.parametrize("s", STRATS)
def test_llf_idct_downsample_is_dc(oracle, s):
    cx, cy = oracle.covered_blocks(s)
    rng = np.random.default_rng(s * 65537 + 13)
    for y in range(cy):
        for x in range(cx):
            if (x > 4 or y > 4) and rng.random() < (0.9 if cx * cy <= 64 else 0.99):
                continue
            dc = np.zeros((cy, cx), np.float32)
            dc[y, x] = 0.2
            llf = oracle.llf_from_dc(s, dc)
            px = oracle.transform_to_pixels(s, llf)
            down = px.reshape(cy, 8, cx, 8).astype(np.float64).mean(axis=(1, 3))
            assert np.abs(down - dc).max() < 1e-6, s


# --- ac_strategy_test.cc:158-226 AcStrategyDownsample ------------------------
@pytest.mark.parametrize("s", STRATS)
def test_lowfreq_idct_downsample_matches_dc_from_llf(oracle, s):
    cx, cy = oracle.covered_blocks(s)
    lo, hi = min(cx, cy), max(cx, cy)  # CoefficientLayout
    rng = np.random.default_rng(s * 65537 + 13)
    for y in range(lo):
        for x in range(hi):
            if (x > 4 or y > 4) and rng.random() < (0.9 if cx * cy <= 64 else 0.99):
                continue
            co = np.zeros(64 * cx * cy, np.float32)
            co[y * hi * 8 + x] = 0.2
            px = oracle.transform_to_pixels(s, co)
            want = oracle.dc_from_llf(s, co)
            down = px.reshape(cy, 8, cx, 8).astype(np.float64).mean(axis=(1, 3))
            assert np.abs(down - want).max() < 2e-6, s


# --- ac_strategy_test.cc:231-245 RoundtripAFVDCT ------------------------------
def test_afv_basis_orthonormal(oracle):
    B = np.ctypeslib.as_array(oracle.lib().jxo_afv_basis(), (256,)).reshape(16, 16)
    assert np.abs(B @ B.T - np.eye(16)).max() < 1e-6
    assert np.abs(B.T @ B - np.eye(16)).max() < 1e-6


# --- quant_weights_test.cc:185-271 (structure + sanity of default tables) ----
def test_default_dequant_tables(oracle):
    t = oracle.default_dequant_tables()
    assert np.isfinite(t).all() and (t > 0).all()
    L = oracle.lib()
    # layout: 17 kinds, each 3 channels (quant_weights.cc:1190-1209)
    assert L.jxo_dequant_table_offset(26, 2) + 64 * 16 * 32 == oracle.DEQUANT_TABLE_FLOATS
    # DCT8: first band values X 3150, Y 560, B 512 -> dequant(0,0) = 1/w
    o = [L.jxo_dequant_table_offset(0, c) for c in range(3)]
    assert abs(t[o[0]] - 1 / 3150.0) < 1e-9
    assert abs(t[o[1]] - 1 / 560.0) < 1e-9
    assert abs(t[o[2]] - 1 / 512.0) < 1e-9
    # weights decrease (dequant steps grow) with frequency along the diagonal
    d8 = t[o[1]:o[1] + 64].reshape(8, 8)
    assert d8[7, 7] > d8[3, 3] > d8[0, 0]
    # R x C and C x R share a table (quant_weights.h:337-348)
    assert L.jxo_dequant_table_offset(6, 0) == L.jxo_dequant_table_offset(7, 0)
    # IDENTITY (mode ID): w = {280,3160,3160} at [0],[1],[8]/[9] for X
    oi = L.jxo_dequant_table_offset(1, 0)
    assert abs(t[oi + 1] - 1 / 3160.0) < 1e-9 and abs(t[oi + 2] - 1 / 280.0) < 1e-9


def test_fast_powf_accuracy(oracle):
    # fast_math-inl.h:88 "max relative error ~3e-5"
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        b = float(rng.uniform(0.05, 20.0))
        e = float(rng.uniform(0.0, 1.0))
        got = L.jxo_fast_powf(b, e)
        assert abs(got - b ** e) / (b ** e) < 5e-5


# --- quantizer-inl.h:34-67 -----------------------------------------------------
def test_adjust_quant_bias(oracle):
    L = oracle.lib()
    b = (C.c_float * 4)(1 - 0.05465007330715401, 1 - 0.07005449891748593,
                        1 - 0.049935103337343655, 0.145)
    for c in range(3):
        assert L.jxo_adjust_quant_bias(c, 0, b) == 0.0
        assert L.jxo_adjust_quant_bias(c, 1, b) == pytest.approx(b[c], abs=0)
        assert L.jxo_adjust_quant_bias(c, -1, b) == pytest.approx(-b[c], abs=0)
        for q in (2, -2, 3, 17, -300, 32767):
            want = q - 0.145 / q
            assert abs(L.jxo_adjust_quant_bias(c, q, b) - want) < 1e-6 * abs(q)


# --- opsin_inverse_test.cc:27-49 LinearInverseInverts --------------------------
def test_opsin_inverse_inverts(oracle, small_frame_factory=None):
    from tests.frames import default_params
    p = default_params(128, 128)
    L = oracle.lib()
    rng = np.random.default_rng(1)
    rgb = rng.random((128, 128, 3)).astype(np.float32)
    xyb = np.zeros((3, 128, 128), np.float32)
    tmp = np.zeros(3, np.float32)
    for y in range(128):
        for x in range(128):
            L.jxo_linear_rgb_to_xyb(float(rgb[y, x, 0]), float(rgb[y, x, 1]),
                                    float(rgb[y, x, 2]), _p(tmp))
            xyb[:, y, x] = tmp
    f = oracle.OracleFrame()
    f.p = p
    out = np.zeros((128, 128, 3), np.float32)
    planes = [np.ascontiguousarray(xyb[c]) for c in range(3)]
    L.jxo_xyb_to_linear_rgb(C.byref(f), (C.c_void_p * 3)(*[a.ctypes.data for a in planes]),
                            128, _p(out), 128 * 3, 0, 128)
    # intensity_target 255 in default_params => matrix scale 1
    assert np.abs(out - rgb).mean() < 3e-3 and np.abs(out - rgb).max() < 2e-4 * 8
