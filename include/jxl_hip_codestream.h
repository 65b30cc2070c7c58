/* jxl_hip_codestream.h -- one call from the bytes of a .jxl file to pixels in HBM (SURVEY.md section 8, row f4:
 * the host front-end in front of the VarDCT back-end, glued).
 *
 * What FrameDecoder does for a plain VarDCT still image, with the same order of operations (libjxl tree, lib/jxl/):
 *   container boxes (jxlc / jxlp)                       decode.cc:1639-1672 (ParseBoxHeader), 1674-2020 (HandleBoxes)
 *   signature, SizeHeader, ImageMetadata, transform data decode.cc:1049-1133  -> jxlhip_image_header_decode
 *   the original's ICC profile                          decode.cc:1101-1130, icc_codec.cc -> jxlhip_icc_decode
 *   FrameHeader, TOC                                    dec_frame.cc:96-189  -> jxlhip_frame_header_decode, jxlhip_toc_decode
 *   ProcessDCGlobal                                     dec_frame.cc:268-302 -> jxlhip_dc_global_decode, jxlhip_modular_global_decode
 *   ProcessDCGroup on the pool                          dec_frame.cc:318-342, 660-680 -> jxlhip_dc_group_decode on the runner
 *   FinalizeDC (DequantDC + AdaptiveDCSmoothing)        dec_frame.cc:344-360, compressed_dc.cc:128-250 -> jxlhip_dequant_dc_groups (device)
 *   ProcessACGlobal                                     dec_frame.cc:372-416 -> jxlhip_ac_global_decode, jxlhip_dequant_tables (device)
 *   ProcessACGroup on the pool                          dec_frame.cc:455-560, 700-760 -> jxlhip_ac_groups_decode_submit
 *   ... its Modular half (extra channels) + FinalizeDecoding  dec_frame.cc:497-530, dec_modular.cc:739-760
 *                                                       -> jxlhip_modular_ac_group_decode[_f32], jxlhip_modular_finalize
 *   the render pipeline                                 dec_cache.cc:117-371 -> jxlhip_decode_frame (device)
 * Taken: any enumerated colour encoding and ICC originals (pixels then linear sRGB, like JxlDecoder without a CMS),
 * grey images, up to four full-resolution integer extra channels (alpha into a 4-channel output, all of them as host
 * planes) incl. the squeeze `cjxl -p` puts on them and palettes without deltas, progressive passes, orientation.
 * Everything this front-end does not decode is refused with JXLHIP_ERR_UNSUPPORTED so that the caller can hand the
 * file to libjxl's CPU decoder: Modular-mode frames, animation / multiple frames, previews, patches / splines / noise,
 * chroma subsampling and YCbCr (JPEG recompression), upsampling, cropped frames, DC frames, RAW dequant tables, RCT /
 * delta palettes in the extra channels' Modular streams.
 */
#ifndef JXL_HIP_CODESTREAM_H_
#define JXL_HIP_CODESTREAM_H_

#include <stddef.h>
#include <stdint.h>

#include "jxl_hip.h"
#include "jxl_hip_entropy.h"
#include "jxl_hip_frame.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jxlhip_codestream_info {
  uint32_t xsize, ysize;       /* image = frame size */
  uint32_t container;          /* the bytes were an ISOBMFF container (jxlc / jxlp boxes) */
  uint32_t orientation;        /* 1..8 (ImageMetadata::orientation); see JXLHIP_OUT_UNDO_ORIENTATION */
  float intensity_target;      /* ImageMetadata::tone_mapping.intensity_target */
  uint32_t bits_per_sample;    /* of the ORIGINAL image (metadata; the decode itself is float) */
  uint32_t transfer_function;  /* enumerated colour encoding of the original: CICP code (13 = sRGB, 8 = linear, 16 = PQ, 18 = HLG ...) */
  uint32_t primaries, white_point;
  /* filled by jxlhip_decode_codestream only */
  uint32_t num_passes, num_groups, num_dc_groups;
  uint32_t epf_iters, gab;
  uint32_t used_acs;
  uint32_t coeff_type;         /* JXLHIP_COEFF_I16, or I32 after the JXLHIP_ERR_RANGE redo */
  uint32_t fused;              /* reserved */
  /* headers: the image's extra channels (this front-end takes up to four, full resolution, integer samples) and the
     first one of type alpha: its bit depth (0 = the image has no alpha channel), whether it is premultiplied */
  uint32_t num_extra_channels, alpha_bits, alpha_premultiplied;
  /* headers: the pixels are produced in the image's ORIGINAL colour space (primaries / white_point above; the inverse
     opsin matrix is adapted like OutputEncodingInfo::SetColorEncoding, dec_xyb.cc:180-249).  luminances: the
     luminance weights of that space, for jxlhip_output_format::luminances (HLG OOTF); gamma: the original's gamma
     exponent when its transfer function is a gamma curve (tf_param of JXLHIP_TF_GAMMA), else 0 */
  float luminances[3];
  float gamma;
  /* headers: the original carried an ICC profile of this many bytes (0 = an enumerated colour encoding; fetch it with
     jxlhip_codestream_icc_profile).  The pixels of such an image are LINEAR sRGB (transfer_function 8, primaries /
     white_point 1), grey for a grey profile -- JxlDecoder's output when no CMS is set (dec_xyb.cc:160-164).
     grey: the original is a grey image (R = G = B in every output) */
  uint32_t icc_size, grey;
} jxlhip_codestream_info;

/* Headers only (no device needed): size and colour metadata of the first frame's image.  JXLHIP_ERR_BAD_STREAM /
 * JXLHIP_ERR_UNSUPPORTED as the decode call would return them for the header part. */
JXLHIP_EXPORT int jxlhip_codestream_basic_info(const uint8_t* data, size_t size, jxlhip_codestream_info* info);

/* The original's ICC profile (JxlDecoderGetColorAsICCProfile(JXL_COLOR_PROFILE_TARGET_ORIGINAL), decode.cc:2411-2430)
 * of a .jxl file or bare codestream.  *icc_size = its size, 0 for an image with an enumerated colour encoding;
 * icc_capacity 0 only asks for the size, a capacity below the size is JXLHIP_ERR_INVALID_ARGUMENT. */
JXLHIP_EXPORT int jxlhip_codestream_icc_profile(const uint8_t* data, size_t size, uint8_t* icc, size_t icc_capacity,
                                                size_t* icc_size);

/* Decodes the (single, VarDCT) frame of a .jxl file or bare codestream into device memory.
 *   alpha                  : an alpha channel of the image is decoded (host, Modular) and written as the fourth
 *                            channel of a 4-channel JXLHIP_OUT_PACKED output; every other output ignores it (its bytes
 *                            are skipped), and a frame without one gives the opaque value.  Not un-premultiplied.
 *   runner / runner_opaque : a JxlParallelRunner (include/jxl/parallel_runner.h; e.g. JxlThreadParallelRunner of
 *                            libjxl_threads_hip.so) for the DC groups and the AC groups; NULL = calling thread
 *   output_kind, out_format: as jxlhip_frame_params (out_format only for JXLHIP_OUT_PACKED; NULL otherwise).
 *                            output_kind | JXLHIP_OUT_UNDO_ORIENTATION writes the pixels in DISPLAY orientation, as
 *                            JxlDecoder does by default (jxlhip_frame_params::undo_orientation = the image's
 *                            orientation: for orientations 5..8 `out` is ysize pixels wide and xsize rows high);
 *                            without the flag the pixels stay in coded orientation (JxlDecoderSetKeepOrientation)
 *   out, out_stride, out_plane_stride : as jxlhip_decode_frame (device pointer)
 * The call returns after the frame is complete (jxlhip_sync included).  The context is left with the frame's
 * inputs resident: jxlhip_decode_frame can re-render (another output format after a new jxlhip_frame_begin needs
 * the inputs again: call this function again). */
#define JXLHIP_OUT_UNDO_ORIENTATION 0x100u
JXLHIP_EXPORT int jxlhip_decode_codestream(jxlhip_ctx* ctx, jxlhip_parallel_runner runner, void* runner_opaque,
                                           const uint8_t* data, size_t size, uint32_t output_kind,
                                           const jxlhip_output_format* out_format, void* out, size_t out_stride,
                                           size_t out_plane_stride, jxlhip_codestream_info* info);

/* jxlhip_decode_codestream, and the image's extra channels (alpha, depth, spot colours ...) as float planes in HOST
 * memory -- what JxlDecoderSetExtraChannelBuffer with JXL_TYPE_FLOAT gives (decode.cc:2627; samples v / (2^bits - 1),
 * ModularImageToDecodedRect dec_modular.cc:686-737): extra_planes[e] for extra channel e < num_extra_planes (NULL = not
 * wanted; entries beyond the image's channels are ignored), rows of extra_stride floats (>= xsize), in CODED
 * orientation (JXLHIP_OUT_UNDO_ORIENTATION turns `out` only).  The planes are complete when the call returns.  An
 * alpha channel may be asked for here and ride in a 4-channel `out` at the same time. */
JXLHIP_EXPORT int jxlhip_decode_codestream_extra(jxlhip_ctx* ctx, jxlhip_parallel_runner runner, void* runner_opaque,
                                                 const uint8_t* data, size_t size, uint32_t output_kind,
                                                 const jxlhip_output_format* out_format, void* out, size_t out_stride,
                                                 size_t out_plane_stride, float* const* extra_planes,
                                                 uint32_t num_extra_planes, size_t extra_stride,
                                                 jxlhip_codestream_info* info);

/* Wall-clock milliseconds of the phases of the LAST jxlhip_decode_codestream[_extra] call on this context, ms[0 ..
 * JXLHIP_CODESTREAM_PHASES): what bench.py's `e2e` block prints beside the whole-file rate (the project's own measure is
 * the whole call: tools/djxl_main.cc:392-426, tools/speed_stats.cc:102-121). */
enum {
  JXLHIP_PHASE_HEADERS = 0,        /* container, image / frame header, TOC, DC global, the global Modular tree */
  /* With a runner, the DC groups, AC global and the AC groups are ONE runner call over num_dc_groups + 1 + num_groups
   * work units (the AC groups under a DC group start when its strategy map / quant field are in: DESIGN.md section 4;
   * JXLHIP_NO_PIPELINE=1 restores the three barriers, JXLHIP_CODESTREAM_VERBOSE=1 prints the call's timeline to stderr):
   * DC_GROUPS is then the time until the LAST DC group ended, AC_GROUPS what of the call came after it, AC_GLOBAL ~0. */
  JXLHIP_PHASE_DC_GROUPS = 1,      /* DecodeVarDCTDC + DecodeAcMetadata of every DC group, on the runner */
  JXLHIP_PHASE_AC_GLOBAL = 2,      /* block contexts, dequant encodings, histograms and coefficient orders of every pass */
  JXLHIP_PHASE_SIDE_INFO = 3,      /* frame_begin, side-info uploads, DC dequant + smoothing and dequant tables queued */
  JXLHIP_PHASE_AC_GROUPS = 4,      /* entropy decode of every AC group on the runner + the coefficient uploads queued */
  JXLHIP_PHASE_EXTRA_CHANNELS = 5, /* alpha / extra channels (Modular), 0 when none is asked for */
  JXLHIP_PHASE_KERNELS = 6,        /* jxlhip_decode_frame + jxlhip_sync: what is left of uploads and kernels */
  JXLHIP_CODESTREAM_PHASES = 7
};
JXLHIP_EXPORT int jxlhip_codestream_phase_ms(const jxlhip_ctx* ctx, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_CODESTREAM_H_ */
