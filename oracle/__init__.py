"""ctypes binding of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from libjxl_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjxl_oracle.so")

NUM_STRATEGIES = 27
DEQUANT_TABLE_FLOATS = 2056 * 64 * 3


def build(force=False):
    """Compile the oracle with gcc (make); a no-op when up to date."""
    inc = os.path.join(_HERE, "..", "include")  # the restatement shares the C ABI's POD structs (jxl_oracle.h)
    deps = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")] if os.path.isdir(inc) else []
    if force or not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class LoopFilter(C.Structure):
    _fields_ = [("gab", C.c_uint32), ("gab_weights", C.c_float * 6),
                ("epf_iters", C.c_uint32), ("epf_sharp_lut", C.c_float * 8),
                ("epf_channel_scale", C.c_float * 3),
                ("epf_quant_mul", C.c_float),
                ("epf_pass0_sigma_scale", C.c_float),
                ("epf_pass2_sigma_scale", C.c_float),
                ("epf_border_sad_mul", C.c_float)]


class OutputFormat(C.Structure):
    _fields_ = [("transfer", C.c_uint32), ("sample_type", C.c_uint32),
                ("num_channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("swap_endianness", C.c_uint32), ("tf_param", C.c_float),
                ("luminances", C.c_float * 3)]


class FrameParams(C.Structure):
    """Mirror of jxlhip_frame_params (include/jxl_hip.h)."""
    _fields_ = [("xsize", C.c_uint32), ("ysize", C.c_uint32),
                ("coeff_type", C.c_uint32), ("output_kind", C.c_uint32),
                ("global_scale", C.c_int32), ("quant_dc", C.c_int32),
                ("x_dm_multiplier", C.c_float), ("b_dm_multiplier", C.c_float),
                ("quant_biases", C.c_float * 4),
                ("cfl_base_x", C.c_float), ("cfl_base_b", C.c_float),
                ("cfl_color_factor", C.c_uint32),
                ("lf", LoopFilter),
                ("opsin_biases", C.c_float * 3),
                ("inverse_opsin_matrix", C.c_float * 9),
                ("stripe_group_y0", C.c_uint32),
                ("stripe_group_rows", C.c_uint32),
                ("out_format", OutputFormat),
                ("used_acs", C.c_uint32), ("undo_orientation", C.c_uint32)]


class OracleFrame(C.Structure):
    _fields_ = [("p", FrameParams),
                ("coeffs", C.c_void_p * 3),
                ("ac_strategy", C.c_void_p), ("raw_quant", C.c_void_p),
                ("epf_sharpness", C.c_void_p),
                ("ytox_map", C.c_void_p), ("ytob_map", C.c_void_p),
                ("dc", C.c_void_p * 3),
                ("dequant_table", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p = C.POINTER(C.c_float)
        L.jxo_dequant_table_offset.restype = C.c_size_t
        L.jxo_afv_basis.restype = f32p
        L.jxo_fast_powf.restype = C.c_float
        L.jxo_fast_powf.argtypes = [C.c_float, C.c_float]
        L.jxo_adjust_quant_bias.restype = C.c_float
        L.jxo_adjust_quant_bias.argtypes = [C.c_int, C.c_int32, f32p]
        L.jxo_idct1d.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.jxo_dct1d.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.jxo_scaled_idct.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.jxo_scaled_dct.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.jxo_idct1d_slow.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.jxo_dct1d_slow.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.jxo_transform_to_pixels.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.jxo_transform_from_pixels.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.jxo_llf_from_dc.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.jxo_dc_from_llf.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.jxo_default_dequant_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.jxo_decode_groups.argtypes = [C.POINTER(OracleFrame), C.c_void_p * 3, C.c_size_t, C.c_uint32, C.c_uint32]
        L.jxo_compute_sigma.argtypes = [C.POINTER(OracleFrame), C.c_void_p]
        L.jxo_gaborish.argtypes = [C.POINTER(OracleFrame), C.c_void_p * 3, C.c_void_p * 3, C.c_size_t, C.c_uint32, C.c_uint32]
        L.jxo_epf.argtypes = [C.POINTER(OracleFrame), C.c_int, C.c_void_p, C.c_void_p * 3, C.c_void_p * 3, C.c_size_t, C.c_uint32, C.c_uint32]
        L.jxo_xyb_to_linear_rgb.argtypes = [C.POINTER(OracleFrame), C.c_void_p * 3, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.jxo_decode_frame.argtypes = [C.POINTER(OracleFrame), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.jxo_dequant_dc.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p * 3, C.c_void_p * 3, C.c_void_p, C.c_float, C.c_float]
        L.jxo_adaptive_dc_smoothing.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p * 3]
        L.jxo_linear_rgb_to_xyb.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _p3(arrs):
    return (C.c_void_p * 3)(*[a.ctypes.data for a in arrs])


# ---- small numpy conveniences used by the tests ---------------------------
def covered_blocks(s):
    L = lib()
    return L.jxo_covered_blocks_x(s), L.jxo_covered_blocks_y(s)


def default_dequant_tables():
    t = np.zeros(DEQUANT_TABLE_FLOATS, np.float32)
    rc = lib().jxo_default_dequant_tables(_p(t), None)
    assert rc == 0
    return t


def transform_to_pixels(strategy, coeffs):
    cx, cy = covered_blocks(strategy)
    co = np.ascontiguousarray(coeffs, np.float32).copy()
    px = np.zeros((8 * cy, 8 * cx), np.float32)
    lib().jxo_transform_to_pixels(strategy, _p(co), _p(px), 8 * cx)
    return px


def transform_from_pixels(strategy, pixels):
    cx, cy = covered_blocks(strategy)
    px = np.ascontiguousarray(pixels, np.float32)
    co = np.zeros(64 * cx * cy, np.float32)
    lib().jxo_transform_from_pixels(strategy, _p(px), 8 * cx, _p(co))
    return co


def llf_from_dc(strategy, dc):
    cx, cy = covered_blocks(strategy)
    d = np.ascontiguousarray(dc, np.float32)
    llf = np.zeros(64 * cx * cy, np.float32)
    lib().jxo_llf_from_dc(strategy, _p(d), cx, _p(llf))
    return llf


def dc_from_llf(strategy, block):
    cx, cy = covered_blocks(strategy)
    b = np.ascontiguousarray(block, np.float32)
    dc = np.zeros((cy, cx), np.float32)
    lib().jxo_dc_from_llf(strategy, _p(b), _p(dc), cx)
    return dc


class Frame:
    """Holds numpy inputs alive and exposes the jxo_frame struct."""

    def __init__(self, params, coeffs, ac_strategy, raw_quant, epf_sharpness,
                 ytox_map, ytob_map, dc, dequant_table):
        self.keep = (coeffs, ac_strategy, raw_quant, epf_sharpness, ytox_map,
                     ytob_map, dc, dequant_table)
        f = OracleFrame()
        f.p = params
        for c in range(3):
            f.coeffs[c] = coeffs[c].ctypes.data
            f.dc[c] = dc[c].ctypes.data
        f.ac_strategy = ac_strategy.ctypes.data
        f.raw_quant = raw_quant.ctypes.data
        f.epf_sharpness = epf_sharpness.ctypes.data
        f.ytox_map = ytox_map.ctypes.data
        f.ytob_map = ytob_map.ctypes.data
        f.dequant_table = dequant_table.ctypes.data
        self.c = f
        self.params = params

    @property
    def dims(self):
        p = self.params
        return (p.xsize + 7) // 8, (p.ysize + 7) // 8

    def decode_groups(self):
        xsb, ysb = self.dims
        planes = [np.zeros((ysb * 8, xsb * 8), np.float32) for _ in range(3)]
        ng = ((self.params.xsize + 255) // 256) * ((self.params.ysize + 255) // 256)
        rc = lib().jxo_decode_groups(C.byref(self.c), _p3(planes), xsb * 8, 0, ng)
        if rc != 0:
            raise ValueError("malformed strategy map")
        return planes

    def compute_sigma(self):
        xsb, ysb = self.dims
        s = np.zeros((ysb, xsb), np.float32)
        lib().jxo_compute_sigma(C.byref(self.c), _p(s))
        return s

    def gaborish(self, planes):
        out = [np.zeros_like(p) for p in planes]
        lib().jxo_gaborish(C.byref(self.c), _p3(planes), _p3(out),
                           planes[0].shape[1], 0, self.params.ysize)
        return out

    def epf(self, which, sigma, planes):
        out = [np.zeros_like(p) for p in planes]
        lib().jxo_epf(C.byref(self.c), which, _p(sigma), _p3(planes), _p3(out),
                      planes[0].shape[1], 0, self.params.ysize)
        return out

    def xyb_to_rgb(self, planes):
        p = self.params
        rgb = np.zeros((p.ysize, p.xsize, 3), np.float32)
        lib().jxo_xyb_to_linear_rgb(C.byref(self.c), _p3(planes),
                                    planes[0].shape[1], _p(rgb), p.xsize * 3,
                                    0, p.ysize)
        return rgb

    def decode(self, threads=1):
        p = self.params
        if p.output_kind == 2:  # packed: stride in bytes; F16 as raw uint16 bits
            of = p.out_format
            dt = {0: np.float32, 1: np.uint8, 2: np.uint16, 3: np.uint16}[of.sample_type]
            out = np.zeros((p.ysize, p.xsize, of.num_channels), dt)
            rc = lib().jxo_decode_frame(C.byref(self.c), _p(out), out.strides[0], 0, threads)
        elif p.output_kind == 1:
            out = np.zeros((p.ysize, p.xsize, 3), np.float32)
            rc = lib().jxo_decode_frame(C.byref(self.c), _p(out), p.xsize * 3, 0, threads)
        else:
            out = np.zeros((3, p.ysize, p.xsize), np.float32)
            rc = lib().jxo_decode_frame(C.byref(self.c), _p(out), p.xsize,
                                        p.xsize * p.ysize, threads)
        if rc != 0:
            raise ValueError("oracle decode failed")
        return out


# ---- oracle/_ref: the libjxl reference itself (compiled in place) ----------
_REF_SO = os.path.join(_HERE, "_ref", "libjxl_ref.so")
_ref = None


def ref_available():
    """True when oracle/_ref/libjxl_ref.so exists or can be built here."""
    from . import build_ref
    return os.path.exists(_REF_SO) or build_ref.available()


def build_reference():
    """Compile the reference decoder sources in place (needs /root/reference;
    on the GPU box the prebuilt .so is used)."""
    from . import build_ref
    return build_ref.build()


def ref_lib():
    global _ref
    if _ref is None:
        lib()  # the restatement exports the shared POD helpers
        build_reference()
        L = C.CDLL(_REF_SO)
        L.jxr_decode_frame.argtypes = [C.POINTER(OracleFrame), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.jxr_default_dequant_tables.argtypes = [C.c_void_p]
        L.jxr_dequant_dc.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p * 3, C.c_void_p * 3, C.c_void_p,
                                     C.c_float, C.c_float, C.c_int]
        L.jxr_describe.restype = C.c_char_p
        _ref = L
    return _ref


_REF_FMA_SO = os.path.join(_HERE, "_ref", "libjxl_ref_fma.so")
_ref_fma = None


def ref_lib_fma():
    """The same reference sources compiled -O3 -mavx2 -mfma (build_ref.py variant "fma"): bench.py's cpu_baseline
    only.  Falls back to the checker build when the host CPU has no AVX2 / FMA or the library is absent."""
    global _ref_fma
    if _ref_fma is None:
        from . import build_ref
        try:
            flags = open("/proc/cpuinfo").read()
            if " avx2" not in flags or " fma" not in flags:
                raise RuntimeError("host CPU without AVX2 / FMA")
            build_ref.build(variant="fma")
            L = C.CDLL(_REF_FMA_SO)
            L.jxr_decode_frame.argtypes = [C.POINTER(OracleFrame), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
            _ref_fma = L
        except (RuntimeError, OSError):
            _ref_fma = False
    return _ref_fma or None


_ref_v8 = None


def ref_lib_v8():
    """The reference's decode hot path on 8 float lanes (build_ref.py variant "v8": dec_group.cc and the Gaborish / EPF /
    XYB / write stages compiled against oracle/hwy_shim_v, 256-bit vectors; everything else = the "fma" objects):
    bench.py's cpu_baseline -- libjxl's SIMD code path, not its code on one lane.  None without AVX2 / FMA."""
    global _ref_v8
    if _ref_v8 is None:
        from . import build_ref
        try:
            flags = open("/proc/cpuinfo").read()
            if " avx2" not in flags or " fma" not in flags:
                raise RuntimeError("host CPU without AVX2 / FMA")
            L = C.CDLL(build_ref.build(variant="v8"))
            L.jxr_decode_frame.argtypes = [C.POINTER(OracleFrame), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
            _ref_v8 = L
        except (RuntimeError, OSError):
            _ref_v8 = False
    return _ref_v8 or None


def ref_default_dequant_tables():
    t = np.zeros(DEQUANT_TABLE_FLOATS, np.float32)
    assert ref_lib().jxr_default_dequant_tables(_p(t)) == 0
    return t


def dequant_tables(encodings):
    """The C restatement's tables for 17 jxlhip_quant_encoding (a ctypes array, e.g.
    libjxl_amd.abi.QuantEncodings); None when the reference would reject them."""
    t = np.zeros(DEQUANT_TABLE_FLOATS, np.float32)
    L = lib()
    L.jxo_dequant_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.jxo_dequant_tables(C.cast(C.byref(encodings), C.c_void_p), _p(t), None)
    return t if rc == 0 else None


def ref_dequant_encode(encodings):
    """The encodings as the reference's DequantMatricesEncode writes them (bytes)."""
    L = ref_lib()
    L.jxr_dequant_encode.restype = C.c_int64
    L.jxr_dequant_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = np.zeros(1 << 16, np.uint8)
    n = L.jxr_dequant_encode(C.cast(C.byref(encodings), C.c_void_p), _p(buf), len(buf))
    if n < 0:
        raise ValueError("reference could not encode the quant encodings")
    return bytes(buf[:n])


def ref_dequant_decode(data):
    """DequantMatrices::Decode + EnsureComputed by the reference on raw bytes:
    (status, table, bits_consumed); status 0 ok, 1 Decode failed, 2 compute failed."""
    L = ref_lib()
    L.jxr_dequant_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    d = np.frombuffer(data, np.uint8)
    t = np.zeros(DEQUANT_TABLE_FLOATS, np.float32)
    bits = C.c_size_t(0)
    rc = L.jxr_dequant_decode(d.ctypes.data, len(d), _p(t), C.byref(bits))
    return rc, t, bits.value


def ref_dequant_dc(quant_dc, mul_dc, cfl_x_dc, cfl_b_dc, smooth, mul=1.0):
    """DequantDC (+AdaptiveDCSmoothing) by the reference; quant_dc: 3 int32 planes;
    mul = 1 / (1 << extra_precision) of the DC group."""
    ysb, xsb = quant_dc[0].shape
    q = [np.ascontiguousarray(a, np.int32) for a in quant_dc]
    out = [np.zeros((ysb, xsb), np.float32) for _ in range(3)]
    m = np.ascontiguousarray(mul_dc, np.float32)
    fn = ref_lib().jxr_dequant_dc_mul
    fn.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int]
    rc = fn(xsb, ysb, _p3(q), _p3(out), _p(m), mul, cfl_x_dc, cfl_b_dc, int(smooth))
    assert rc == 0
    return out


def ref_threads(xsize, ysize, max_threads):
    """Threads for Frame.decode_ref when its result is the expected value of a test.

    The reference's LowMemoryRenderPipeline is not independent of the order in which groups finish
    (root-caused in round 2, tools/probes/ref_mirror_fix_probe.py; pinned by
    tests/test_reference_parity.py::test_reference_group_order_dependence_is_the_mirroring_test).  RenderRect lets a
    stage run xextra_right columns past its rect (low_memory_render_pipeline.cc:751-757) but ApplyXMirroring
    (:486-517) mirrors the stage's input at the right image edge only when rect.x1 + border_x >= image_xsize --
    xextra_right is not part of that test.  A rect that ends within xextra_right + border_x of the edge, but not
    within border_x of it, makes the stage read columns past the image edge that nobody wrote: stale floats of the
    thread's stage buffer (small errors, up to 6.5e-4 seen; NaN when the memory was never used).  Such a rect only
    exists when the LAST group column is narrower than 16 + the stage list's total border (at most 16 + 7) AND
    finishes before its left neighbour, whose border strip [x1 - 16, x1 + 16) is then rendered on its own: never
    with one thread (groups finish in index order), now and then with several.  The in-order result equals
    SimpleRenderPipeline's and the C restatement's bit for bit, and so does the threaded result once the
    mirroring test includes the extra columns -- so frames with such a last column are decoded with one
    thread; every other frame (4K / 8K: widths that are multiples of 256) keeps the threads, where 1 and N threads
    are bit-identical (test_reference_threads_bit_identical_when_last_column_is_wide)."""
    narrow = 0 < (xsize % 256) < 16 + 8
    small = ((xsize + 255) // 256) * ((ysize + 255) // 256) <= 16
    return 1 if (narrow or small) else max_threads


def _decode_ref(self, threads=1, simple_pipeline=False, quant_encodings=None, fma_build=False, v8_build=False):
    """The same frame through the REFERENCE's DecodeGroupForRoundtrip + render
    pipeline (LowMemory executor by default, as djxl; simple_pipeline=True for
    SimpleRenderPipeline).  The reference computes its own dequant tables: the
    default library, or quant_encodings (17 jxlhip_quant_encoding) when given."""
    if quant_encodings is not None:
        L = ref_lib()
        L.jxr_set_quant_encodings.argtypes = [C.c_void_p]
        assert L.jxr_set_quant_encodings(C.cast(C.byref(quant_encodings), C.c_void_p)) == 0
        try:
            return _decode_ref(self, threads, simple_pipeline)
        finally:
            L.jxr_set_quant_encodings(None)
    p = self.params
    threads = ref_threads(p.xsize, p.ysize, threads)
    ref_lib()
    R = (ref_lib_v8() if v8_build else None) or (ref_lib_fma() if fma_build else None) or ref_lib()
    self.last_ref_lib = R  # (last_decode_seconds reads this library's clock)
    # undo_orientation 5..8: the reference writes an xsize-high, ysize-wide frame (stage_write.cc:664-680)
    oh, ow = (p.xsize, p.ysize) if p.undo_orientation >= 5 else (p.ysize, p.xsize)
    if p.output_kind == 2:
        # packed RGB(A) through the reference's FromLinearStage + WriteToOutputStage;
        # out_stride is in BYTES for this kind; F16 comes back as raw uint16 bits
        of = p.out_format
        dt = {0: np.float32, 1: np.uint8, 2: np.uint16, 3: np.uint16}[of.sample_type]
        out = np.zeros((oh, ow, of.num_channels), dt)
        rc = R.jxr_decode_frame(C.byref(self.c), _p(out), out.strides[0], 0, threads,
                                        int(simple_pipeline))
    elif p.output_kind == 1:
        out = np.zeros((oh, ow, 3), np.float32)
        rc = R.jxr_decode_frame(C.byref(self.c), _p(out), ow * 3, 0, threads, int(simple_pipeline))
    else:
        out = np.zeros((3, p.ysize, p.xsize), np.float32)
        rc = R.jxr_decode_frame(C.byref(self.c), _p(out), p.xsize, p.xsize * p.ysize, threads,
                                        int(simple_pipeline))
    if rc != 0:
        raise ValueError("reference decode failed")
    return out


Frame.decode_ref = _decode_ref


def _last_decode_seconds(self):
    """Seconds the last decode_ref spent in the reference's threaded group decode + render pipeline (jxr_decode_frame
    without the driver's own serial set-up: oracle/ref_driver.cc)."""
    R = getattr(self, "last_ref_lib", None) or ref_lib()
    R.jxr_last_decode_seconds.restype = C.c_double
    return float(R.jxr_last_decode_seconds())


Frame.last_decode_seconds = _last_decode_seconds


def _encode_ac_ref(self, force_huffman=False, lz77_method=0, custom_orders=True, histo_sets=1,
                   custom_block_ctx=False, quant_dc=None, want_block_ctx=False):
    """The frame's quantized coefficients as AC entropy streams written by the
    REFERENCE's own encoder (oracle/ref_driver.cc EncodeAc).  Returns
    (global_bytes, [group_bytes...], used_acs, used_orders) and, with
    want_block_ctx, the EncodeBlockCtxMap bytes as a fifth element."""
    p = self.params
    ng = ((p.xsize + 255) // 256) * ((p.ysize + 255) // 256)
    gcap, cap = 1 << 22, max(1 << 20, ng * 65536 * 3 * 4)
    gbuf = np.zeros(gcap, np.uint8)
    bbuf = np.zeros(1 << 16, np.uint8)
    buf = np.zeros(cap, np.uint8)
    offs = np.zeros(ng + 1, np.uint64)
    gsize, bsize = C.c_size_t(0), C.c_size_t(0)
    used_acs, used_orders = C.c_uint32(0), C.c_uint32(0)
    L = ref_lib()
    L.jxr_encode_ac.argtypes = [C.POINTER(OracleFrame), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_void_p,
                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    qdc = None if quant_dc is None else np.ascontiguousarray(quant_dc, np.uint8)
    rc = L.jxr_encode_ac(C.byref(self.c), int(force_huffman), int(lz77_method), int(custom_orders), int(histo_sets),
                         int(custom_block_ctx), None if qdc is None else qdc.ctypes.data, _p(bbuf), len(bbuf),
                         C.byref(bsize), _p(gbuf), gcap, C.byref(gsize), _p(buf), cap, _p(offs),
                         C.byref(used_acs), C.byref(used_orders))
    if rc != 0:
        raise ValueError("reference AC encode failed (%d)" % rc)
    groups = [bytes(buf[int(offs[g]):int(offs[g + 1])]) for g in range(ng)]
    res = (bytes(gbuf[:gsize.value]), groups, used_acs.value, used_orders.value)
    return res + (bytes(bbuf[:bsize.value]),) if want_block_ctx else res


def ref_quant_dc_contexts(quant_dc):
    """quant_dc context indices by the reference's DequantDC under the test block
    context map; quant_dc: 3 int32 planes in X, Y, B order."""
    ysb, xsb = quant_dc[0].shape
    q = [np.ascontiguousarray(a, np.int32) for a in quant_dc]
    out = np.zeros((ysb, xsb), np.uint8)
    L = ref_lib()
    L.jxr_quant_dc_contexts.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p * 3, C.c_void_p]
    assert L.jxr_quant_dc_contexts(xsb, ysb, _p3(q), _p(out)) == 0
    return out


Frame.encode_ac_ref = _encode_ac_ref


def feature_stream(feature, xsize=600, ysize=400, seed=5, distance=1.0):
    """A small codestream written by the REFERENCE ENCODER (oracle/ref_real_stream.cc: FeatureStream) that uses one
    feature the product's seam declines or handles specially: "noise" | "splines" | "patches" | "modular" | "animation" |
    "progressive" | "plain" (the control).  Describes an 8-bit sRGB original.  Test infrastructure."""
    L = ref_lib()
    L.jxr_feature_stream.restype = C.c_size_t
    L.jxr_feature_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_char_p, C.c_void_p, C.c_size_t]
    buf = C.create_string_buffer(max(1 << 20, 4 * xsize * ysize))
    n = L.jxr_feature_stream(xsize, ysize, seed, distance, feature.encode(), buf, len(buf))
    if not n:
        raise ValueError("reference encoder failed for feature %r" % feature)
    return buf.raw[:n]


class RealStream:
    """A genuine VarDCT codestream written by the REFERENCE's own encoder
    (jxl::EncodeFrame on a procedural image, oracle/ref_real_stream.cc), the
    reference FrameDecoder's pixels for it, and the inputs of the product's
    boundary lifted from that decoder's PassesSharedState.  Test infrastructure."""

    _WHAT = dict(codestream=(0, np.uint8), ac_strategy=(1, np.uint8), raw_quant=(2, np.int32),
                 epf_sharpness=(3, np.uint8), ytox_map=(4, np.int8), ytob_map=(5, np.int8),
                 dc_x=(6, np.float32), dc_y=(7, np.float32), dc_b=(8, np.float32), quant_dc=(9, np.uint8),
                 block_ctx_bytes=(10, np.uint8), rgb=(11, np.float32), section_offset=(12, np.uint64),
                 section_size=(13, np.uint64), params=(14, np.uint8), dequant_table=(15, np.float32),
                 alpha=(16, np.float32), extra=(17, np.float32))

    def __init__(self, xsize, ysize, seed=1, distance=1.0, speed_tier=3, epf=-1, progressive=0, alpha_bits=0,
                 alpha_levels=0, original=None, icc=None, extra=0):
        L = ref_lib()
        L.jxr_real_case_create.restype = C.c_void_p
        L.jxr_real_case_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int]
        L.jxr_real_case_destroy.argtypes = [C.c_void_p]
        L.jxr_real_case_data.restype = C.c_void_p
        L.jxr_real_case_data.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.jxr_real_case_info.restype = C.c_uint64
        L.jxr_real_case_info.argtypes = [C.c_void_p, C.c_int]
        # original = "srgb8" | "srgb16": the stream describes an integer sRGB original, like a file cjxl made from a PNG
        # icc = bytes: the original carries this ICC profile (coded behind the image header; pixels: linear sRGB)
        icc_file = None
        if icc:
            import tempfile
            icc_file = tempfile.NamedTemporaryFile(suffix=".icc", delete=False)
            icc_file.write(bytes(icc))
            icc_file.close()
        # extra = n: n more extra channels (depth 16 bit, thermal 8 bit, optional 12 bit) behind the alpha channel
        knobs = {"JXR_ALPHA": alpha_bits, "JXR_ALPHA_LEVELS": alpha_levels, "JXR_ORIGINAL": original, "JXR_EXTRA": extra,
                 "JXR_ICC_FILE": icc_file.name if icc_file else None}
        old = {k: os.environ.get(k) for k in knobs}
        for k, v in knobs.items():
            if v:
                os.environ[k] = str(v)
        try:
            h = L.jxr_real_case_create(xsize, ysize, seed, distance, speed_tier, epf, progressive)
        finally:
            for k, v in knobs.items():
                if v:
                    if old[k] is None:
                        del os.environ[k]
                    else:
                        os.environ[k] = old[k]
            if icc_file:
                os.unlink(icc_file.name)
        if not h:
            raise ValueError("reference encode/decode failed")
        try:
            for name, (what, dt) in self._WHAT.items():
                n = C.c_size_t(0)
                p = L.jxr_real_case_data(h, what, C.byref(n))
                a = np.frombuffer(C.string_at(p, n.value), dt).copy() if n.value else np.zeros(0, dt)
                setattr(self, name, a)
            (self.num_groups, self.num_dc_groups, self.num_histograms, self.used_acs, self.frame_offset,
             self.sections_offset) = [int(L.jxr_real_case_info(h, i)) for i in range(6)]
            self.num_passes = int(L.jxr_real_case_info(h, 6))
            self.toc_bit_offset = int(L.jxr_real_case_info(h, 7))
            self.shift = [int(L.jxr_real_case_info(h, 16 + i)) for i in range(self.num_passes)]
        finally:
            L.jxr_real_case_destroy(h)
        self.xsize, self.ysize = xsize, ysize
        xsb, ysb = (xsize + 7) // 8, (ysize + 7) // 8
        for name in ("ac_strategy", "raw_quant", "epf_sharpness", "dc_x", "dc_y", "dc_b", "quant_dc"):
            setattr(self, name, getattr(self, name).reshape(ysb, xsb))
        self.rgb = self.rgb.reshape(ysize, xsize, 3)
        self.alpha = self.alpha.reshape(ysize, xsize) if alpha_bits else None
        self.frame_params = FrameParams.from_buffer_copy(self.params.tobytes())

    def section(self, logical_id):
        """Bytes of TOC section `logical_id` (0 = DC global, 1.. = DC groups, then AC
        global, then one per AC group: frame_header.h / dec_frame.cc:655-705)."""
        o, n = int(self.section_offset[logical_id]), int(self.section_size[logical_id])
        return self.codestream[o:o + n].tobytes()

    def ac_global(self):
        return self.section(1 + self.num_dc_groups)

    def ac_group(self, g, pass_idx=0):
        # AcGroupIndex (frame_dimensions / toc.h): passes are the outer dimension
        return self.section(2 + self.num_dc_groups + pass_idx * self.num_groups + g)

    def frame(self, coeffs):
        """An oracle Frame over this stream's side info and the given coefficient buffers."""
        return Frame(self.frame_params, coeffs, self.ac_strategy, self.raw_quant, self.epf_sharpness,
                     self.ytox_map, self.ytob_map, [self.dc_x, self.dc_y, self.dc_b], self.dequant_table)
