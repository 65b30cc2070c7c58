/* jxl_hip_frame.h -- host front-end pieces in front of the VarDCT back-end (SURVEY.md section
 * 8, row f4): the image header, the frame header (everything FrameDecoder::InitFrame reads before
 * the table of contents, jxlhip_toc_decode in jxl_hip_entropy.h), the DC-global section and the
 * Modular-coded DC groups of a VarDCT frame.  Plain C ABI, host code only.
 *
 * Replaces, behaviour for behaviour (libjxl tree, lib/jxl/):
 *   FrameHeader::VisitFields, Passes, BlendingInfo, AnimationFrame   frame_header.cc:63-439
 *   YCbCrChromaSubsampling, VisitNameString                          frame_header.h:35-50,76-168
 *   LoopFilter::VisitFields                                          loop_filter.cc:18-106
 *   the field coders U32 / U64 / Bool / Bits / F16 / extensions     fields.cc:179-262,494-574
 *   FrameHeader::ToFrameDimensions, FrameDimensions::Set             frame_header.h:466-484,
 *                                                                    frame_dimensions.h:34-60
 */
#ifndef JXL_HIP_FRAME_H_
#define JXL_HIP_FRAME_H_

#include <stddef.h>
#include <stdint.h>

#include "jxl_hip.h"
#include "jxl_hip_entropy.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- The codestream's image header: signature, SizeHeader, ImageMetadata, CustomTransformData ----
 * Replaces (libjxl tree, lib/jxl/):
 *   the 0xFF 0x0A signature check and the read order            decode.cc:1049-1133
 *   SizeHeader / PreviewHeader / AnimationHeader                 headers.cc:59-66,117-125,127-198
 *   BitDepth, ExtraChannelInfo, ImageMetadata, ToneMapping,
 *   OpsinInverseMatrix, CustomTransformData                      image_metadata.cc:25-61,72-197,199-245,
 *                                                                270-392
 *   ColorEncoding, Customxy, CustomTransferFunction, Enum()      color_encoding_internal.cc:106-216,
 *                                                                fields.h:205-216, field_encodings.h:119-131
 * What the back-end takes from it: the image size, xyb_encoded, the inverse opsin matrix / opsin biases /
 * quant biases (jxlhip_frame_params), intensity_target (the 255/intensity_target scale of the matrix,
 * dec_xyb.cc:175-186), the enumerated colour encoding and bit depth (jxlhip_output_format), and the inputs
 * of jxlhip_frame_header_decode (jxlhip_image_info).
 * The reference also refuses headers whose CUSTOM white point / primaries its ICC synthesiser cannot express
 * (ColorEncoding::CreateICC at the end of VisitFields; cms/jxl_cms_internal.h:43-126,235-244,354-372,403-409):
 * those conditions are restated (no profile is written), so the verdicts agree.
 * Not here (colour management, out of scope): the ICC stream that follows when color_encoding.want_icc is
 * set is left unread. */
typedef struct jxlhip_bit_depth {
  uint32_t floating_point_sample, bits_per_sample, exponent_bits_per_sample;
} jxlhip_bit_depth;

enum { JXLHIP_EC_ALPHA = 0, JXLHIP_EC_DEPTH = 1, JXLHIP_EC_SPOT_COLOR = 2, JXLHIP_EC_SELECTION_MASK = 3,
       JXLHIP_EC_BLACK = 4, JXLHIP_EC_CFA = 5, JXLHIP_EC_THERMAL = 6, JXLHIP_EC_UNKNOWN = 15, JXLHIP_EC_OPTIONAL = 16 };

typedef struct jxlhip_extra_channel {
  uint32_t all_default;
  uint32_t type;               /* JXLHIP_EC_* */
  jxlhip_bit_depth bit_depth;
  uint32_t dim_shift;          /* 0..3 */
  uint32_t name_length;
  uint32_t alpha_associated;   /* type ALPHA */
  float spot_color[4];         /* type SPOT_COLOR */
  uint32_t cfa_channel;        /* type CFA */
} jxlhip_extra_channel;

enum { JXLHIP_CS_RGB = 0, JXLHIP_CS_GRAY = 1, JXLHIP_CS_XYB = 2, JXLHIP_CS_UNKNOWN = 3 };
enum { JXLHIP_WP_D65 = 1, JXLHIP_WP_CUSTOM = 2, JXLHIP_WP_E = 10, JXLHIP_WP_DCI = 11 };
enum { JXLHIP_PRIM_SRGB = 1, JXLHIP_PRIM_CUSTOM = 2, JXLHIP_PRIM_2100 = 9, JXLHIP_PRIM_P3 = 11 };
/* transfer_function holds the CICP code as coded (1 = 709, 8 = linear, 13 = sRGB, 16 = PQ, 17 = DCI, 18 = HLG);
 * jxlhip_output_format::transfer_function uses the back-end's own JXLHIP_TF_* numbering. */

typedef struct jxlhip_color_encoding {
  uint32_t all_default;        /* sRGB, D65, relative intent */
  uint32_t want_icc;           /* an ICC stream follows the headers (not read here) */
  uint32_t color_space;        /* JXLHIP_CS_* */
  uint32_t white_point;        /* JXLHIP_WP_* */
  uint32_t primaries;          /* JXLHIP_PRIM_*; meaningful for RGB / UNKNOWN colour spaces */
  uint32_t have_gamma, gamma;  /* gamma in units of 1e-7 */
  uint32_t transfer_function;
  uint32_t rendering_intent;   /* 0 perceptual, 1 relative, 2 saturation, 3 absolute */
  int32_t white_xy[2];         /* custom chromaticities in units of 1e-6 */
  int32_t primaries_xy[6];     /* r.x r.y g.x g.y b.x b.y */
} jxlhip_color_encoding;

typedef struct jxlhip_image_header {
  uint32_t xsize, ysize;                         /* SizeHeader */
  uint32_t all_default;                          /* ImageMetadata::all_default */
  uint32_t orientation;                          /* 1..8 */
  uint32_t have_intrinsic_size, intrinsic_xsize, intrinsic_ysize;
  uint32_t have_preview, preview_xsize, preview_ysize;
  uint32_t have_animation, tps_numerator, tps_denominator, num_loops, have_timecodes;
  jxlhip_bit_depth bit_depth;
  uint32_t modular_16_bit_buffer_sufficient;
  uint32_t num_extra_channels;
  uint32_t xyb_encoded;
  jxlhip_color_encoding color_encoding;
  uint32_t tone_mapping_all_default;
  float intensity_target, min_nits;
  uint32_t relative_to_max_display;
  float linear_below;
  uint64_t extensions;
  /* CustomTransformData */
  uint32_t transform_all_default, opsin_all_default;
  float inverse_opsin_matrix[9];                 /* UNSCALED (multiply by 255/intensity_target for frame_params) */
  float opsin_biases[3];
  float quant_biases[4];
  uint32_t custom_weights_mask;                  /* bit 0/1/2: the 2x/4x/8x upsampling weights below are coded */
  float upsampling2_weights[15], upsampling4_weights[55], upsampling8_weights[210]; /* zero when not coded
                                                    (upsampling is outside the back-end; defaults not carried) */
} jxlhip_image_header;

/* Reads the image header of a bare codestream starting at byte 0 of data (0xFF 0x0A).  extra[] receives
 * the first min(num_extra_channels, extra_capacity) channel descriptions (may be NULL with capacity 0).
 * On success *bit_pos is the first bit after CustomTransformData when an ICC stream follows
 * (color_encoding.want_icc), otherwise the byte-aligned position of the first frame header (the
 * padding bits must be zero, BitReader::JumpToByteBoundary).  JXLHIP_ERR_BAD_STREAM on everything
 * the reference rejects and on truncation. */
JXLHIP_EXPORT int jxlhip_image_header_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                             jxlhip_extra_channel* extra, size_t extra_capacity,
                                             jxlhip_image_header* out);

/* The ICC profile of the original, coded behind the image header when color_encoding.want_icc (ICCReader::Init /
 * Process + UnpredictICC, icc_codec.cc:135-334,340-421; read by JxlDecoderReadAllHeaders, decode.cc:1101-1133).
 * *bit_pos: in = the position jxlhip_image_header_decode returned, out = the byte-aligned first bit of the first
 * frame header.  icc may be NULL (the stream is still decoded and checked: that is the only way to find its end);
 * *icc_size (may be NULL) = the profile's size; icc_capacity below it with a non-NULL icc is INVALID_ARGUMENT.
 * JXLHIP_ERR_BAD_STREAM on everything the reference rejects. */
JXLHIP_EXPORT int jxlhip_icc_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint8_t* icc,
                                    size_t icc_capacity, size_t* icc_size);

/* The inverse opsin matrix (unscaled: multiply by 255 / intensity_target for jxlhip_frame_params) that makes the
 * back-end's pixels come out in the image's ORIGINAL colour space -- what OutputEncodingInfo::SetFromMetadata /
 * SetColorEncoding derive (dec_xyb.cc:144-165,180-249): the coded matrix for sRGB / D65 originals, the coded matrix
 * followed by sRGB -> XYZ(D50) -> original primaries / white point otherwise (P3, Rec.2100, custom xy); for a grey
 * (D65) original every row becomes luminances x matrix, so that R = G = B = the grey sample -- and the luminance
 * weights of that space (jxlhip_output_format::luminances, the HLG OOTF).  An ICC original (want_icc): the reference
 * without a CMS falls back to LINEAR sRGB (grey for a grey profile, dec_xyb.cc:160-164), and so does this: the coded
 * matrix, transfer function linear.  JXLHIP_ERR_UNSUPPORTED: a transfer function outside the enumerated ones, an
 * image that is not XYB encoded. */
JXLHIP_EXPORT int jxlhip_output_opsin_matrix(const jxlhip_image_header* header, float inverse_matrix[9],
                                             float luminances[3]);

/* What the frame header's conditions read from the image header (CodecMetadata). */
typedef struct jxlhip_image_info {
  uint32_t xsize, ysize;          /* image size, or the preview size when is_preview */
  uint32_t xyb_encoded;           /* ImageMetadata::xyb_encoded */
  uint32_t num_extra_channels;    /* extra_channel_info.size(), <= 4096 */
  const uint8_t* ec_dim_shift;    /* ExtraChannelInfo::dim_shift per extra channel; NULL = all 0 */
  uint32_t have_animation;        /* ImageMetadata::have_animation */
  uint32_t have_timecodes;        /* AnimationHeader::have_timecodes */
  uint32_t is_preview;            /* FrameHeader::nonserialized_is_preview */
  uint32_t bits_per_sample;       /* ImageMetadata::bit_depth.bits_per_sample: lands in jxlhip_frame_header::image_bits */
} jxlhip_image_info;

enum { JXLHIP_FRAME_REGULAR = 0, JXLHIP_FRAME_DC = 1, JXLHIP_FRAME_REFERENCE_ONLY = 2, JXLHIP_FRAME_SKIP_PROGRESSIVE = 3 };
enum { JXLHIP_CT_XYB = 0, JXLHIP_CT_NONE = 1, JXLHIP_CT_YCBCR = 2 };
enum { JXLHIP_FLAG_NOISE = 1, JXLHIP_FLAG_PATCHES = 2, JXLHIP_FLAG_SPLINES = 16, JXLHIP_FLAG_USE_DC_FRAME = 32,
       JXLHIP_FLAG_SKIP_ADAPTIVE_DC_SMOOTHING = 128 };

typedef struct jxlhip_frame_header {
  uint32_t all_default;
  uint32_t frame_type;         /* JXLHIP_FRAME_* */
  uint32_t is_modular;         /* FrameEncoding::kModular (the back-end decodes VarDCT frames) */
  uint32_t color_transform;    /* JXLHIP_CT_* */
  uint64_t flags;              /* JXLHIP_FLAG_* */
  uint32_t chroma_mode[3];     /* YCbCrChromaSubsampling::channel_mode_ (0 = 1x1) */
  uint32_t upsampling;         /* 1, 2, 4, 8 */
  uint32_t group_size_shift;   /* modular frames */
  uint32_t x_qm_scale, b_qm_scale;
  uint32_t num_passes, num_downsample;
  uint32_t shift[11], downsample[4], last_pass[4];
  uint32_t dc_level;
  uint32_t custom_size_or_origin;
  int32_t x0, y0;              /* frame_origin */
  uint32_t coded_xsize, coded_ysize; /* frame_size as coded (0 = the image size) */
  uint32_t blend_mode, blend_alpha_channel, blend_clamp, blend_source;
  uint32_t duration, timecode;
  uint32_t is_last, save_as_reference, save_before_color_transform;
  uint32_t name_length;
  uint64_t extensions;
  /* LoopFilter: the part the back-end takes as jxlhip_frame_params::lf, and the rest */
  uint32_t lf_all_default, gab_custom, epf_sharp_custom, epf_weight_custom, epf_sigma_custom;
  jxlhip_loop_filter lf;
  float epf_pass1_zeroflush, epf_pass2_zeroflush, epf_sigma_for_modular;
  uint64_t lf_extensions;
  /* derived: FrameHeader::ToFrameDimensions and what PassesDecoderState::Init makes of the scales */
  uint32_t xsize, ysize;       /* frame size in pixels (after upsampling division, dc_level) */
  uint32_t xsize_blocks, ysize_blocks, group_dim;
  uint32_t xsize_groups, ysize_groups;
  uint64_t num_groups, num_dc_groups, num_toc_entries; /* 64-bit: a custom frame size may reach 2^30 squared */
  float x_dm_multiplier, b_dm_multiplier;
  /* copied from jxlhip_image_info: the Modular parts of the frame carry the extra channels (alpha, depth, ...;
     dec_modular.cc:230-262).  jxlhip_modular_global_decode / jxlhip_modular_ac_group_decode take up to four of them,
     each at the frame's own resolution (ec_upsampling == upsampling == 1); JXLHIP_ERR_UNSUPPORTED otherwise */
  uint32_t num_extra_channels;
  uint32_t ec_upsampling[4];   /* FrameHeader::extra_channel_upsampling of the first four (dim_shift applied) */
  uint32_t image_bits;         /* jxlhip_image_info::bits_per_sample (the Modular Image's bitdepth: the implicit entries
                                  of multi-channel palettes scale with it, palette.h:53-122); 0 = not given: such a
                                  palette is then JXLHIP_ERR_UNSUPPORTED */
} jxlhip_frame_header;

/* ReadFrameHeader (frame_header.cc:212-215): reads the header at bit *bit_pos of data (advanced to
 * the first bit of the TOC).  JXLHIP_ERR_BAD_STREAM on every condition the reference rejects
 * (and on truncation), JXLHIP_ERR_INVALID_ARGUMENT on bad arguments. */
JXLHIP_EXPORT int jxlhip_frame_header_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                             const jxlhip_image_info* image, jxlhip_frame_header* out);

/* The VarDCT part of the DC-global section in front of the modular global info
 * (FrameDecoder::ProcessDCGlobal, dec_frame.cc:268-302): DequantMatrices::DecodeDC
 * (quant_weights.cc:513-528), Quantizer::Decode (quantizer.cc:125-149), DecodeBlockCtxMap
 * (entropy_coder.cc:25-61), ColorCorrelation::DecodeDC (chroma_from_luma.cc:24-44) -- the values
 * jxlhip_frame_params, jxlhip_dequant_dc and the AC decoder take.  frame_flags: the frame header's
 * flags; JXLHIP_ERR_UNSUPPORTED when patches, splines or noise precede these fields in the section. */
typedef struct jxlhip_dc_global {
  float dc_quant[3];          /* DequantMatrices::DCQuant(c); default 1/4096, 1/512, 1/256 */
  int32_t global_scale;       /* Quantizer::global_scale_ */
  int32_t quant_dc;           /* Quantizer::quant_dc_ */
  uint32_t cfl_color_factor;  /* ColorCorrelation::color_factor_ (default 84) */
  float cfl_base_x, cfl_base_b;
  int32_t ytox_dc, ytob_dc;
  jxlhip_block_ctx_map block_ctx_map;
} jxlhip_dc_global;
JXLHIP_EXPORT int jxlhip_dc_global_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint64_t frame_flags,
                                          jxlhip_dc_global* out);

/* ---- The Modular-coded parts of a VarDCT frame: global MA tree and the DC groups ----
 * Replaces (libjxl tree, lib/jxl/): ModularFrameDecoder::DecodeGlobalInfo (dec_modular.cc:207-316),
 * FrameDecoder::ProcessDCGroup (dec_frame.cc:318-342) = DecodeVarDCTDC + DecodeAcMetadata
 * (dec_modular.cc:427-562), over DecodeTree (modular/encoding/dec_ma.cc), ModularDecode
 * (modular/encoding/encoding.cc:553-680) and the property / predictor definitions of
 * modular/encoding/context_predict.h, the self-correcting predictor included.
 * JXLHIP_ERR_UNSUPPORTED for what libjxl's encoder does not write into these streams: transforms
 * (RCT, delta palettes; palettes without deltas -- over one channel or several -- and the squeeze of the extra
 * channels' global image are taken, see jxlhip_modular_groups_are_final), LZ77 with 2-D distances; also for chroma
 * subsampling and DC frames. */
typedef struct jxlhip_modular_tree jxlhip_modular_tree;

/* The rest of the DC-global section behind jxlhip_dc_global_decode: *tree receives the global MA
 * tree and its entropy code (NULL when the section carries none: every stream then has its own). */
JXLHIP_EXPORT int jxlhip_modular_global_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                               const jxlhip_frame_header* frame, jxlhip_modular_tree** tree);
JXLHIP_EXPORT void jxlhip_modular_tree_destroy(jxlhip_modular_tree* tree);

/* Extra channels (alpha, depth, ...) of a VarDCT frame.  jxlhip_modular_global_decode has read the global part
 * (group header, transforms, channels that fit one group) into the handle; this call reads what follows the VarDCT
 * coefficients of pass `pass` in AC group `group` -- ModularFrameDecoder::DecodeGroup for
 * ModularStreamId::ModularAC(group, pass), dec_modular.cc:330-425, dec_frame.cc:497-530 -- from bit *bit_pos of the
 * section (advanced): the group's rectangle of every larger extra channel.  Thread-safe for different groups.  A
 * frame without extra channels (tree may be NULL) reads nothing. */
JXLHIP_EXPORT int jxlhip_modular_ac_group_decode(jxlhip_modular_tree* tree, const jxlhip_frame_header* frame,
                                                 uint32_t group, uint32_t pass, const uint8_t* data, size_t size,
                                                 size_t* bit_pos);
/* The same, with the samples converted on the group's thread and written straight into the caller's float planes --
 * planes[e] for extra channel e (NULL = not wanted), rows of stride_floats, ec_bits[e] / image_bits as for
 * jxlhip_modular_extra_channel_f32 -- so that the frame's int32 image is never allocated and nothing is left to
 * convert afterwards.  Channels that fit one group are coded in the global section and do not pass here: they are
 * read with jxlhip_modular_extra_channel_f32.  JXLHIP_ERR_UNSUPPORTED: the global image carries a squeeze or a
 * palette of a palette -- jxlhip_modular_groups_are_final says so beforehand (use the collecting form above). */
JXLHIP_EXPORT int jxlhip_modular_ac_group_decode_f32(jxlhip_modular_tree* tree, const jxlhip_frame_header* frame,
                                                     uint32_t group, uint32_t pass, const uint8_t* data, size_t size,
                                                     size_t* bit_pos, const uint32_t* ec_bits, uint32_t image_bits,
                                                     float* const* planes, size_t stride_floats);
/* (the same with a row stride per plane: strides_floats[e] for planes[e], e < 4) */
JXLHIP_EXPORT int jxlhip_modular_ac_group_decode_f32_strided(jxlhip_modular_tree* tree, const jxlhip_frame_header* frame,
                                                             uint32_t group, uint32_t pass, const uint8_t* data,
                                                             size_t size, size_t* bit_pos, const uint32_t* ec_bits,
                                                             uint32_t image_bits, float* const* planes,
                                                             const size_t* strides_floats);
/* 1: the groups' samples of the extra channels are final as they arrive (jxlhip_modular_ac_group_decode_f32 may write
 * them out); 0: the global image carries transforms that need every group first -- a squeeze (Haar-like pyramid with
 * a smoothness term, modular/transform/squeeze.cc; what `cjxl -p` applies to extra channels: their 1:8 and smaller
 * levels then sit in the DC groups, which jxlhip_dc_group_decode reads into the handle, the finer ones in the AC
 * groups pass by pass), one palette over several channels, or a palette of a palette: collect with jxlhip_modular_ac_group_decode, then
 * jxlhip_modular_extra_channel_f32 undoes them.  A NULL tree (no extra channels) is 1. */
JXLHIP_EXPORT int jxlhip_modular_groups_are_final(const jxlhip_modular_tree* tree);
/* Once every group is in (collecting form): undoes the global image's transforms with the caller's threads -- every
 * level of a squeeze pyramid spread over `runner` (a JxlParallelRunner, NULL = calling thread; 8-row / 64-column tasks
 * like InvHSqueeze / InvVSqueeze, squeeze.cc:171-228,262-303), then the palettes.  jxlhip_modular_extra_channel_f32
 * does the same serially on its first call; after this call it -- and the thread-safe row-range form below -- only
 * convert.  (the runner type: include/jxl_hip_entropy.h) */
JXLHIP_EXPORT int jxlhip_modular_finalize(jxlhip_modular_tree* tree,
                                          int (*runner)(void*, void*, int (*)(void*, size_t), void (*)(void*, uint32_t, size_t),
                                                        uint32_t, uint32_t),
                                          void* runner_opaque);
JXLHIP_EXPORT int jxlhip_modular_extra_channel_rows_f32(const jxlhip_modular_tree* tree, uint32_t ec, uint32_t ec_bits,
                                                        uint32_t image_bits, uint32_t y0, uint32_t y1, float* out,
                                                        size_t stride_floats);
/* 1: part of the extra channels is coded in the DC groups (a squeezed image): a caller that does not run
 * jxlhip_dc_group_decode with this handle cannot complete them. */
JXLHIP_EXPORT int jxlhip_modular_uses_dc_groups(const jxlhip_modular_tree* tree);
/* Once every group is in: extra channel `ec` as float samples, out[y * stride_floats + x] = v / (2^ec_bits - 1)
 * (FinalizeDecoding + ModularImageToDecodedRect, dec_modular.cc:686-737,739-760; image_bits = the IMAGE's
 * bits_per_sample, which picks the float or the double multiply, :726-731).  The first call undoes the global
 * image's transforms; not thread-safe, and no group may be decoded afterwards. */
JXLHIP_EXPORT int jxlhip_modular_extra_channel_f32(jxlhip_modular_tree* tree, uint32_t ec, uint32_t ec_bits,
                                                   uint32_t image_bits, float* out, size_t stride_floats);

/* One DC group section (section 1 + dc_group of the TOC).  All outputs are FRAME-level arrays in the
 * layouts jxlhip_upload_side_info / jxlhip_dequant_dc take, of which this call fills the group's
 * rectangle (256x256 blocks at the default group size):
 *   quant_dc[3]     X, Y, B: int32, xsize_blocks*ysize_blocks -- the input of jxlhip_dequant_dc, with
 *                   *extra_precision (0..3): DequantDC multiplies by 1 / (1 << extra_precision)
 *   ac_strategy     (type << 1) | first_block, one byte per block
 *   raw_quant       int32 in [1, 256], written at the first block of each varblock
 *   epf_sharpness   0..7 per block
 *   ytox_map, ytob_map   int8 per 8x8 blocks, ceil(xsize_blocks/8) per row
 *   *used_acs       |= bit per strategy type seen (jxlhip_frame_params::used_acs)
 * Between the DC image and the metadata sits the ModularDC stream (dec_frame.cc:330-337): the group's rectangle of
 * the extra channels' sub-channels at 1:8 and below, present when their image is squeezed
 * (jxlhip_modular_uses_dc_groups); it is read INTO the handle's image (the handle is const for everything else).
 * Safe to call from several threads for different groups. */
JXLHIP_EXPORT int jxlhip_dc_group_decode(const jxlhip_modular_tree* global_tree, const uint8_t* data, size_t size,
                                         size_t* bit_pos, const jxlhip_frame_header* frame, uint32_t dc_group,
                                         int32_t* const quant_dc[3], uint32_t* extra_precision,
                                         uint8_t* ac_strategy, int32_t* raw_quant, uint8_t* epf_sharpness,
                                         int8_t* ytox_map, int8_t* ytob_map, uint32_t* used_acs);
/* The same, announcing the moment the AC groups under this DC group can start: block_info_ready(opaque) is called (on
 * the calling thread, at most once, only when the call goes on to succeed that far) as soon as the group's rectangles
 * of quant_dc, ac_strategy and raw_quant are complete and *used_acs has this group's bits -- before the EPF sharpness
 * channel, which is the last and, at one sample per block, a tenth of the section's decode time.  The reference runs
 * every ProcessDCGroup to its end before the first ProcessACGroup (dec_frame.cc:318-360,596-703); nothing an AC group
 * reads (dec_group.cc:431-560: strategy, quant field, the DC-derived block contexts) comes from the sharpness map.
 * A failure reported AFTER the callback (a damaged sharpness channel) fails the frame all the same. */
JXLHIP_EXPORT int jxlhip_dc_group_decode_staged(const jxlhip_modular_tree* global_tree, const uint8_t* data, size_t size,
                                                size_t* bit_pos, const jxlhip_frame_header* frame, uint32_t dc_group,
                                                int32_t* const quant_dc[3], uint32_t* extra_precision,
                                                uint8_t* ac_strategy, int32_t* raw_quant, uint8_t* epf_sharpness,
                                                int8_t* ytox_map, int8_t* ytob_map, uint32_t* used_acs,
                                                void (*block_info_ready)(void* opaque), void* opaque);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_FRAME_H_ */
