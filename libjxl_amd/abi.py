"""ctypes binding of the C ABI in include/jxl_hip.h (libjxl_hip.so)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JXLHIP_SO: an experiment build of the library (tools/build_variant.py); default = the product
_SO = os.environ.get("JXLHIP_SO") or os.path.join(_HERE, "csrc", "libjxl_hip.so")
_RUNNER_SO = os.path.join(_HERE, "csrc", "libjxl_threads_hip.so")

KERNEL_COUNT = 8
KERNEL_NAMES = ["prepare", "blocks", "filters", "fused", "epf0", "k5", "k6", "k7"]


class JxlHipError(RuntimeError):
    pass


class LoopFilter(C.Structure):
    _fields_ = [("gab", C.c_uint32), ("gab_weights", C.c_float * 6),
                ("epf_iters", C.c_uint32), ("epf_sharp_lut", C.c_float * 8),
                ("epf_channel_scale", C.c_float * 3),
                ("epf_quant_mul", C.c_float),
                ("epf_pass0_sigma_scale", C.c_float),
                ("epf_pass2_sigma_scale", C.c_float),
                ("epf_border_sad_mul", C.c_float)]


class OutputFormat(C.Structure):
    """jxlhip_output_format: FromLinearStage + WriteToOutputStage parameters."""
    _fields_ = [("transfer", C.c_uint32), ("sample_type", C.c_uint32),
                ("num_channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("swap_endianness", C.c_uint32), ("tf_param", C.c_float),
                ("luminances", C.c_float * 3)]


OUT_XYB_PLANAR, OUT_LINEAR_RGB_F32, OUT_PACKED = 0, 1, 2
TF_LINEAR, TF_SRGB, TF_PQ, TF_709, TF_GAMMA, TF_HLG = 0, 1, 2, 3, 4, 5
SAMPLE_F32, SAMPLE_U8, SAMPLE_U16, SAMPLE_F16 = 0, 1, 2, 3


class BlockCtxMap(C.Structure):
    """jxlhip_block_ctx_map (include/jxl_hip_entropy.h)."""
    _fields_ = [("num_dc_thresholds", C.c_uint32 * 3), ("dc_thresholds", (C.c_int32 * 15) * 3),
                ("num_dc_ctxs", C.c_uint32), ("num_qf_thresholds", C.c_uint32),
                ("qf_thresholds", C.c_uint32 * 15), ("num_ctxs", C.c_uint32), ("ctx_map_size", C.c_uint32),
                ("ctx_map", C.c_uint8 * (3 * 13 * 64))]


QUANT_LIBRARY, QUANT_ID, QUANT_DCT2, QUANT_DCT4, QUANT_DCT4X8, QUANT_AFV, QUANT_DCT, QUANT_RAW = range(8)
NUM_QUANT_TABLES, MAX_DISTANCE_BANDS = 17, 17


class QuantEncoding(C.Structure):
    """jxlhip_quant_encoding: one dequant table's parameters (QuantEncodingInternal)."""
    _fields_ = [("mode", C.c_uint32), ("num_bands", C.c_uint32), ("num_bands_afv_4x4", C.c_uint32),
                ("reserved", C.c_uint32), ("bands", (C.c_float * 17) * 3), ("bands_afv_4x4", (C.c_float * 17) * 3),
                ("weights", (C.c_float * 9) * 3)]


QuantEncodings = QuantEncoding * NUM_QUANT_TABLES


class DcGlobal(C.Structure):
    """jxlhip_dc_global (include/jxl_hip_frame.h)."""
    _fields_ = [("dc_quant", C.c_float * 3), ("global_scale", C.c_int32), ("quant_dc", C.c_int32),
                ("cfl_color_factor", C.c_uint32), ("cfl_base_x", C.c_float), ("cfl_base_b", C.c_float),
                ("ytox_dc", C.c_int32), ("ytob_dc", C.c_int32), ("block_ctx_map", BlockCtxMap)]


class ImageInfo(C.Structure):
    """jxlhip_image_info (include/jxl_hip_frame.h)."""
    _fields_ = [("xsize", C.c_uint32), ("ysize", C.c_uint32), ("xyb_encoded", C.c_uint32),
                ("num_extra_channels", C.c_uint32), ("ec_dim_shift", C.c_void_p), ("have_animation", C.c_uint32),
                ("have_timecodes", C.c_uint32), ("is_preview", C.c_uint32), ("bits_per_sample", C.c_uint32)]


class BitDepth(C.Structure):
    """jxlhip_bit_depth (include/jxl_hip_frame.h)."""
    _fields_ = [("floating_point_sample", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("exponent_bits_per_sample", C.c_uint32)]


class ExtraChannel(C.Structure):
    """jxlhip_extra_channel (include/jxl_hip_frame.h)."""
    _fields_ = [("all_default", C.c_uint32), ("type", C.c_uint32), ("bit_depth", BitDepth), ("dim_shift", C.c_uint32),
                ("name_length", C.c_uint32), ("alpha_associated", C.c_uint32), ("spot_color", C.c_float * 4),
                ("cfa_channel", C.c_uint32)]


class ColorEncoding(C.Structure):
    """jxlhip_color_encoding (include/jxl_hip_frame.h)."""
    _fields_ = [("all_default", C.c_uint32), ("want_icc", C.c_uint32), ("color_space", C.c_uint32),
                ("white_point", C.c_uint32), ("primaries", C.c_uint32), ("have_gamma", C.c_uint32),
                ("gamma", C.c_uint32), ("transfer_function", C.c_uint32), ("rendering_intent", C.c_uint32),
                ("white_xy", C.c_int32 * 2), ("primaries_xy", C.c_int32 * 6)]


class ImageHeader(C.Structure):
    """jxlhip_image_header (include/jxl_hip_frame.h)."""
    _fields_ = [("xsize", C.c_uint32), ("ysize", C.c_uint32), ("all_default", C.c_uint32), ("orientation", C.c_uint32),
                ("have_intrinsic_size", C.c_uint32), ("intrinsic_xsize", C.c_uint32), ("intrinsic_ysize", C.c_uint32),
                ("have_preview", C.c_uint32), ("preview_xsize", C.c_uint32), ("preview_ysize", C.c_uint32),
                ("have_animation", C.c_uint32), ("tps_numerator", C.c_uint32), ("tps_denominator", C.c_uint32),
                ("num_loops", C.c_uint32), ("have_timecodes", C.c_uint32), ("bit_depth", BitDepth),
                ("modular_16_bit_buffer_sufficient", C.c_uint32), ("num_extra_channels", C.c_uint32),
                ("xyb_encoded", C.c_uint32), ("color_encoding", ColorEncoding),
                ("tone_mapping_all_default", C.c_uint32), ("intensity_target", C.c_float), ("min_nits", C.c_float),
                ("relative_to_max_display", C.c_uint32), ("linear_below", C.c_float), ("extensions", C.c_uint64),
                ("transform_all_default", C.c_uint32), ("opsin_all_default", C.c_uint32),
                ("inverse_opsin_matrix", C.c_float * 9), ("opsin_biases", C.c_float * 3),
                ("quant_biases", C.c_float * 4), ("custom_weights_mask", C.c_uint32),
                ("upsampling2_weights", C.c_float * 15), ("upsampling4_weights", C.c_float * 55),
                ("upsampling8_weights", C.c_float * 210)]


class FrameHeader(C.Structure):
    """jxlhip_frame_header (include/jxl_hip_frame.h)."""
    _fields_ = [("all_default", C.c_uint32), ("frame_type", C.c_uint32), ("is_modular", C.c_uint32),
                ("color_transform", C.c_uint32), ("flags", C.c_uint64), ("chroma_mode", C.c_uint32 * 3),
                ("upsampling", C.c_uint32), ("group_size_shift", C.c_uint32), ("x_qm_scale", C.c_uint32),
                ("b_qm_scale", C.c_uint32), ("num_passes", C.c_uint32), ("num_downsample", C.c_uint32),
                ("shift", C.c_uint32 * 11), ("downsample", C.c_uint32 * 4), ("last_pass", C.c_uint32 * 4),
                ("dc_level", C.c_uint32), ("custom_size_or_origin", C.c_uint32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("coded_xsize", C.c_uint32), ("coded_ysize", C.c_uint32), ("blend_mode", C.c_uint32),
                ("blend_alpha_channel", C.c_uint32), ("blend_clamp", C.c_uint32), ("blend_source", C.c_uint32),
                ("duration", C.c_uint32), ("timecode", C.c_uint32), ("is_last", C.c_uint32),
                ("save_as_reference", C.c_uint32), ("save_before_color_transform", C.c_uint32),
                ("name_length", C.c_uint32), ("extensions", C.c_uint64), ("lf_all_default", C.c_uint32),
                ("gab_custom", C.c_uint32), ("epf_sharp_custom", C.c_uint32), ("epf_weight_custom", C.c_uint32),
                ("epf_sigma_custom", C.c_uint32), ("lf", LoopFilter), ("epf_pass1_zeroflush", C.c_float),
                ("epf_pass2_zeroflush", C.c_float), ("epf_sigma_for_modular", C.c_float), ("lf_extensions", C.c_uint64),
                ("xsize", C.c_uint32), ("ysize", C.c_uint32), ("xsize_blocks", C.c_uint32), ("ysize_blocks", C.c_uint32),
                ("group_dim", C.c_uint32), ("xsize_groups", C.c_uint32), ("ysize_groups", C.c_uint32),
                ("num_groups", C.c_uint64), ("num_dc_groups", C.c_uint64), ("num_toc_entries", C.c_uint64),
                ("x_dm_multiplier", C.c_float), ("b_dm_multiplier", C.c_float),
                ("num_extra_channels", C.c_uint32), ("ec_upsampling", C.c_uint32 * 4), ("image_bits", C.c_uint32)]


class CodestreamInfo(C.Structure):
    """jxlhip_codestream_info (include/jxl_hip_codestream.h)."""
    _fields_ = [("xsize", C.c_uint32), ("ysize", C.c_uint32), ("container", C.c_uint32), ("orientation", C.c_uint32),
                ("intensity_target", C.c_float), ("bits_per_sample", C.c_uint32), ("transfer_function", C.c_uint32),
                ("primaries", C.c_uint32), ("white_point", C.c_uint32), ("num_passes", C.c_uint32),
                ("num_groups", C.c_uint32), ("num_dc_groups", C.c_uint32), ("epf_iters", C.c_uint32),
                ("gab", C.c_uint32), ("used_acs", C.c_uint32), ("coeff_type", C.c_uint32), ("fused", C.c_uint32),
                ("num_extra_channels", C.c_uint32), ("alpha_bits", C.c_uint32), ("alpha_premultiplied", C.c_uint32),
                ("luminances", C.c_float * 3), ("gamma", C.c_float), ("icc_size", C.c_uint32), ("grey", C.c_uint32)]


class FrameParams(C.Structure):
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


class FrameInputs(C.Structure):
    _fields_ = [("coeffs", C.c_void_p * 3),
                ("ac_strategy", C.c_void_p), ("raw_quant", C.c_void_p),
                ("epf_sharpness", C.c_void_p),
                ("ytox_map", C.c_void_p), ("ytob_map", C.c_void_p),
                ("dc", C.c_void_p * 3),
                ("dequant_table", C.c_void_p)]


def make_params(d):
    """dict (libjxl_amd.synth.synth_frame) -> FrameParams."""
    p = FrameParams()
    for k in ("xsize", "ysize", "coeff_type", "output_kind", "global_scale",
              "quant_dc", "x_dm_multiplier", "b_dm_multiplier", "cfl_base_x",
              "cfl_base_b", "cfl_color_factor", "stripe_group_y0",
              "stripe_group_rows"):
        setattr(p, k, d[k])
    p.used_acs = d.get("used_acs", 0)
    p.undo_orientation = d.get("undo_orientation", 0)
    p.quant_biases[:] = d["quant_biases"]
    p.opsin_biases[:] = d["opsin_biases"]
    p.inverse_opsin_matrix[:] = d["inverse_opsin_matrix"]
    p.lf.gab = d["gab"]
    p.lf.gab_weights[:] = d["gab_weights"]
    p.lf.epf_iters = d["epf_iters"]
    p.lf.epf_sharp_lut[:] = d["epf_sharp_lut"]
    p.lf.epf_channel_scale[:] = d["epf_channel_scale"]
    p.lf.epf_quant_mul = d["epf_quant_mul"]
    p.lf.epf_pass0_sigma_scale = d["epf_pass0_sigma_scale"]
    p.lf.epf_pass2_sigma_scale = d["epf_pass2_sigma_scale"]
    p.lf.epf_border_sad_mul = d["epf_border_sad_mul"]
    of = d.get("out_format")
    if of:
        p.out_format.transfer = of.get("transfer", TF_LINEAR)
        p.out_format.sample_type = of.get("sample_type", SAMPLE_F32)
        p.out_format.num_channels = of.get("num_channels", 3)
        p.out_format.bits_per_sample = of.get("bits_per_sample", 0)
        p.out_format.swap_endianness = of.get("swap_endianness", 0)
        p.out_format.tf_param = of.get("tf_param", 0.0)
        p.out_format.luminances[:] = of.get("luminances", (0.2126, 0.7152, 0.0722))
    return p


def library_path():
    return _SO


def runner_library_path():
    return _RUNNER_SO


_lib = None

# every symbol include/jxl_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "jxlhip_covered_blocks_x", "jxlhip_covered_blocks_y",
    "jxlhip_log2_covered_blocks", "jxlhip_quant_table_of_strategy",
    "jxlhip_dequant_table_offset", "jxlhip_status_string", "jxlhip_create", "jxlhip_create_ex", "jxlhip_create_multi",
    "jxlhip_destroy", "jxlhip_last_error", "jxlhip_debug_reload_env", "jxlhip_set_stream",
    "jxlhip_frame_begin", "jxlhip_frame_set_inputs", "jxlhip_upload_side_info",
    "jxlhip_submit_group", "jxlhip_set_alpha", "jxlhip_alpha_staging", "jxlhip_decode_blocks", "jxlhip_halo_rows",
    "jxlhip_halo_export", "jxlhip_halo_import", "jxlhip_decode_filters", "jxlhip_decode_filters_rows", "jxlhip_stripe_begin",
    "jxlhip_stripe_finish", "jxlhip_decode_frame",
    "jxlhip_decode_frame_host", "jxlhip_decode_frame_pinned",
    "jxlhip_sync", "jxlhip_export_xyb", "jxlhip_get_sigma",
    "jxlhip_set_concurrency_hint", "jxlhip_profile_enable", "jxlhip_profile_read",
    "jxlhip_dequant_tables", "jxlhip_default_dequant_tables", "jxlhip_dequant_dc",
    "jxlhip_dequant_dc_groups",
    # include/jxl_hip_entropy.h
    "jxlhip_ac_pass_decode", "jxlhip_ac_pass_destroy", "jxlhip_ac_pass_max_num_bits",
    "jxlhip_ac_pass_used_orders", "jxlhip_ac_pass_order", "jxlhip_ac_group_decode", "jxlhip_ac_group_decode_sparse",
    "jxlhip_ac_group_decode_submit", "jxlhip_block_ctx_map_decode", "jxlhip_quant_dc_contexts",
    "jxlhip_dequant_encodings_decode", "jxlhip_ac_global_decode", "jxlhip_ac_group_decode_submit_passes",
    "jxlhip_ac_groups_decode_submit", "jxlhip_ac_groups_decode_submit_ex", "jxlhip_num_toc_entries", "jxlhip_toc_decode", "jxlhip_ac_global_decode_at",
    # include/jxl_hip_frame.h
    "jxlhip_frame_header_decode", "jxlhip_dc_global_decode", "jxlhip_image_header_decode", "jxlhip_icc_decode", "jxlhip_output_opsin_matrix",
    "jxlhip_modular_global_decode", "jxlhip_modular_tree_destroy", "jxlhip_dc_group_decode", "jxlhip_dc_group_decode_staged",
    "jxlhip_modular_ac_group_decode", "jxlhip_modular_ac_group_decode_f32", "jxlhip_modular_extra_channel_f32",
    "jxlhip_modular_groups_are_final", "jxlhip_modular_uses_dc_groups", "jxlhip_modular_finalize",
    "jxlhip_modular_extra_channel_rows_f32", "jxlhip_modular_ac_group_decode_f32_strided",
    # include/jxl_hip_codestream.h
    "jxlhip_codestream_basic_info", "jxlhip_decode_codestream", "jxlhip_decode_codestream_extra",
    "jxlhip_codestream_icc_profile", "jxlhip_codestream_phase_ms",
]


def load_library():
    """Loads libjxl_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise JxlHipError(
            f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(_SO)
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    L.jxlhip_dequant_table_offset.restype = sz
    L.jxlhip_dequant_table_offset.argtypes = [i32, i32]
    L.jxlhip_status_string.restype = C.c_char_p
    L.jxlhip_last_error.restype = C.c_char_p
    L.jxlhip_last_error.argtypes = [vp]
    L.jxlhip_debug_reload_env.restype = None
    L.jxlhip_debug_reload_env.argtypes = []
    L.jxlhip_create.argtypes = [i32, C.POINTER(vp)]
    L.jxlhip_create_ex.argtypes = [i32, vp, C.POINTER(vp)]
    L.jxlhip_create_multi.argtypes = [vp, i32, vp, C.POINTER(vp)]
    L.jxlhip_destroy.argtypes = [vp]
    L.jxlhip_destroy.restype = None
    L.jxlhip_set_stream.argtypes = [vp, vp, i32]
    L.jxlhip_ac_pass_decode.argtypes = [vp, sz, C.POINTER(sz), u32, u32, vp, C.POINTER(vp)]
    L.jxlhip_block_ctx_map_decode.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(BlockCtxMap)]
    L.jxlhip_quant_dc_contexts.argtypes = [C.POINTER(BlockCtxMap), sz, vp * 3, vp]
    L.jxlhip_ac_pass_destroy.argtypes = [vp]
    L.jxlhip_ac_pass_destroy.restype = None
    L.jxlhip_ac_pass_max_num_bits.argtypes = [vp]
    L.jxlhip_ac_pass_max_num_bits.restype = u32
    L.jxlhip_ac_pass_used_orders.argtypes = [vp]
    L.jxlhip_ac_pass_used_orders.restype = u32
    L.jxlhip_ac_group_decode.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, vp, sz, C.POINTER(sz), u32, u32,
                                         vp * 3, C.POINTER(sz)]
    L.jxlhip_ac_group_decode_sparse.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, vp, sz, C.POINTER(sz), u32,
                                                vp * 3, u32 * 3, u32 * 3, C.POINTER(sz)]
    L.jxlhip_ac_group_decode_submit.argtypes = [vp, vp, u32, vp, vp, vp, vp, sz, C.POINTER(sz)]
    L.jxlhip_ac_group_decode_submit_passes.argtypes = [vp, u32, vp, vp, u32, vp, vp, vp, vp, vp, vp]
    L.jxlhip_ac_groups_decode_submit.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.jxlhip_ac_groups_decode_submit_ex.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.jxlhip_num_toc_entries.argtypes = [u32, u32, u32]
    L.jxlhip_num_toc_entries.restype = u32
    L.jxlhip_toc_decode.argtypes = [vp, sz, C.POINTER(sz), u32, vp, vp, vp]
    L.jxlhip_frame_header_decode.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(ImageInfo), C.POINTER(FrameHeader)]
    L.jxlhip_dc_global_decode.argtypes = [vp, sz, C.POINTER(sz), C.c_uint64, C.POINTER(DcGlobal)]
    L.jxlhip_modular_global_decode.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(FrameHeader), C.POINTER(vp)]
    L.jxlhip_modular_tree_destroy.argtypes = [vp]
    L.jxlhip_modular_tree_destroy.restype = None
    L.jxlhip_modular_ac_group_decode.argtypes = [vp, C.POINTER(FrameHeader), u32, u32, vp, sz, C.POINTER(sz)]
    L.jxlhip_modular_extra_channel_f32.argtypes = [vp, u32, u32, u32, vp, sz]
    L.jxlhip_modular_groups_are_final.argtypes = [vp]
    L.jxlhip_modular_uses_dc_groups.argtypes = [vp]
    L.jxlhip_modular_finalize.argtypes = [vp, vp, vp]
    L.jxlhip_modular_extra_channel_rows_f32.argtypes = [vp, u32, u32, u32, u32, u32, vp, sz]
    L.jxlhip_modular_ac_group_decode_f32.argtypes = [vp, C.POINTER(FrameHeader), u32, u32, vp, sz, C.POINTER(sz), vp, u32, vp, sz]
    L.jxlhip_dc_group_decode.argtypes = [vp, vp, sz, C.POINTER(sz), C.POINTER(FrameHeader), C.c_uint32,
                                         C.POINTER(vp), C.POINTER(C.c_uint32), vp, vp, vp, vp, vp,
                                         C.POINTER(C.c_uint32)]
    L.jxlhip_dc_group_decode_staged.argtypes = [vp, vp, sz, C.POINTER(sz), C.POINTER(FrameHeader), C.c_uint32,
                                                C.POINTER(vp), C.POINTER(C.c_uint32), vp, vp, vp, vp, vp,
                                                C.POINTER(C.c_uint32), vp, vp]
    L.jxlhip_image_header_decode.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(ExtraChannel), sz,
                                             C.POINTER(ImageHeader)]
    L.jxlhip_output_opsin_matrix.argtypes = [C.POINTER(ImageHeader), C.c_float * 9, C.c_float * 3]
    L.jxlhip_frame_begin.argtypes = [vp, C.POINTER(FrameParams)]
    L.jxlhip_frame_set_inputs.argtypes = [vp, C.POINTER(FrameInputs)]
    L.jxlhip_upload_side_info.argtypes = [vp, vp, vp, vp, vp, vp, vp * 3, vp]
    L.jxlhip_submit_group.argtypes = [vp, u32, vp * 3, sz]
    L.jxlhip_set_alpha.argtypes = [vp, vp, sz]
    L.jxlhip_alpha_staging.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlhip_decode_blocks.argtypes = [vp]
    L.jxlhip_halo_rows.argtypes = [vp]
    L.jxlhip_halo_export.argtypes = [vp, i32, vp]
    L.jxlhip_halo_import.argtypes = [vp, i32, vp]
    L.jxlhip_decode_filters.argtypes = [vp, vp, sz, sz]
    L.jxlhip_decode_filters_rows.argtypes = [vp, vp, sz, sz, u32, u32]
    L.jxlhip_stripe_begin.argtypes = [vp, vp, vp]
    L.jxlhip_stripe_finish.argtypes = [vp, vp, vp, vp, sz, sz, u32, u32]
    L.jxlhip_decode_frame.argtypes = [vp, vp, sz, sz]
    L.jxlhip_decode_frame_host.argtypes = [vp, vp, sz, sz]
    L.jxlhip_decode_frame_pinned.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlhip_sync.argtypes = [vp]
    L.jxlhip_export_xyb.argtypes = [vp, vp * 3, sz]
    L.jxlhip_get_sigma.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlhip_profile_enable.argtypes = [vp, i32]
    L.jxlhip_set_concurrency_hint.argtypes = [vp, i32]
    L.jxlhip_profile_read.argtypes = [vp, C.c_float * KERNEL_COUNT, u32 * KERNEL_COUNT]
    L.jxlhip_default_dequant_tables.argtypes = [vp, vp]
    L.jxlhip_dequant_tables.argtypes = [vp, vp, vp]
    L.jxlhip_dequant_encodings_decode.argtypes = [vp, sz, C.POINTER(sz), vp]
    L.jxlhip_ac_global_decode.argtypes = [vp, sz, u32, u32, u32, vp, vp, C.POINTER(u32), C.POINTER(vp),
                                          C.POINTER(sz)]
    L.jxlhip_dequant_dc.argtypes = [vp, vp * 3, vp * 3, vp, C.c_float, C.c_float, i32]
    L.jxlhip_dequant_dc_groups.argtypes = [vp, vp * 3, vp * 3, vp, C.c_float, C.c_float, i32, vp]
    L.jxlhip_codestream_basic_info.argtypes = [vp, sz, C.POINTER(CodestreamInfo)]
    L.jxlhip_codestream_icc_profile.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]
    L.jxlhip_icc_decode.argtypes = [vp, sz, C.POINTER(sz), vp, sz, C.POINTER(sz)]
    L.jxlhip_decode_codestream.argtypes = [vp, vp, vp, vp, sz, u32, vp, vp, sz, sz, C.POINTER(CodestreamInfo)]
    L.jxlhip_codestream_phase_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.jxlhip_decode_codestream_extra.argtypes = [vp, vp, vp, vp, sz, u32, vp, vp, sz, sz, C.POINTER(vp), u32, sz,
                                                 C.POINTER(CodestreamInfo)]
    L.jxlhip_modular_ac_group_decode_f32_strided.argtypes = [vp, vp, u32, u32, vp, sz, C.POINTER(sz), vp, u32, vp, vp]
    L.jxlhip_ac_global_decode_at.argtypes = [vp, sz, C.POINTER(sz), u32, u32, u32, vp, vp, C.POINTER(u32), C.POINTER(vp)]
    _lib = L
    return L
