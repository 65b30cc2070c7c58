/* jxl_hip_frame.h -- host front-end pieces in front of the VarDCT back-end (SURVEY.md section
 * 8, row f4): the frame header, i.e. everything FrameDecoder::InitFrame reads before the table
 * of contents (jxlhip_toc_decode, jxl_hip_entropy.h).  Plain C ABI, host code only.
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

/* What the frame header's conditions read from the image header (CodecMetadata). */
typedef struct jxlhip_image_info {
  uint32_t xsize, ysize;          /* image size, or the preview size when is_preview */
  uint32_t xyb_encoded;           /* ImageMetadata::xyb_encoded */
  uint32_t num_extra_channels;    /* extra_channel_info.size(), <= 4096 */
  const uint8_t* ec_dim_shift;    /* ExtraChannelInfo::dim_shift per extra channel; NULL = all 0 */
  uint32_t have_animation;        /* ImageMetadata::have_animation */
  uint32_t have_timecodes;        /* AnimationHeader::have_timecodes */
  uint32_t is_preview;            /* FrameHeader::nonserialized_is_preview */
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

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_FRAME_H_ */
