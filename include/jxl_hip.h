/*
 * jxl_hip.h -- C ABI of the MI355X (gfx950) VarDCT decode back-end.
 *
 * This is the drop-in boundary for ONE hot path of libjxl: everything between
 * "quantized AC coefficients + per-block side info are in memory" and "float
 * pixels are in the output buffer":
 *
 *   dequant + chroma-from-luma + LLF-from-DC + variable-size inverse transform
 *   -> [Gaborish] -> [EPF0] [EPF1] [EPF2] -> XYB -> linear RGB
 *
 * Each entry point cites the reference interface (path relative to the libjxl
 * tree) that it replaces.  Plain C: pointers + sizes only, no C++/torch types.
 * All functions return JXLHIP_OK (0) or a negative jxlhip_status; nothing
 * throws; there is NO CPU fallback (a missing device is an error).
 *
 * Coordinate / layout conventions are the reference's own:
 *   - block  = 8x8 px; group = 256x256 px = 32x32 blocks; colour tile = 64x64 px
 *     (lib/jxl/frame_dimensions.h:21-27, lib/jxl/chroma_from_luma.h:28-32)
 *   - channel order c = 0:X 1:Y 2:B
 *   - AC coefficient stream per group and channel: varblocks in raster visit
 *     order of their top-left block, 64*covered_blocks coefficients each, in the
 *     layout of the dequant matrix (rows = short side, lib/jxl/dec_group.cc:221,
 *     334-359; lib/jxl/coeff_order_fwd.h:27-43); group g starts at element
 *     g*65536 (lib/jxl/dec_frame.cc:423, lib/jxl/dct_util.h:23-96)
 */
#ifndef JXL_HIP_H_
#define JXL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define JXLHIP_EXPORT __attribute__((visibility("default")))
#else
#define JXLHIP_EXPORT
#endif

#define JXLHIP_BLOCK_DIM 8
#define JXLHIP_GROUP_DIM 256
#define JXLHIP_GROUP_DIM_IN_BLOCKS 32
#define JXLHIP_GROUP_COEFFS 65536 /* per channel */
#define JXLHIP_COLOR_TILE_DIM_IN_BLOCKS 8
#define JXLHIP_NUM_STRATEGIES 27
#define JXLHIP_NUM_QUANT_TABLES 17
/* lib/jxl/quant_weights.h:412-417: sum(required_size_x*required_size_y)=2056 */
#define JXLHIP_DEQUANT_TABLE_FLOATS (2056 * 64 * 3)

typedef enum {
  JXLHIP_OK = 0,
  JXLHIP_ERR_INVALID_ARGUMENT = -1,
  JXLHIP_ERR_NO_DEVICE = -2,
  JXLHIP_ERR_OUT_OF_MEMORY = -3,
  JXLHIP_ERR_HIP = -4,         /* a HIP runtime call failed; see last_error */
  JXLHIP_ERR_BAD_STREAM = -5,  /* side info violates a format constraint */
  JXLHIP_ERR_STATE = -6,       /* call sequence error */
  JXLHIP_ERR_UNSUPPORTED = -7, /* valid stream feature outside this back-end */
  JXLHIP_ERR_RANGE = -8        /* a coefficient does not fit the 16-bit buffers: use JXLHIP_COEFF_I32 */
} jxlhip_status;

/* ACType, lib/jxl/dct_util.h:23; chosen per frame at lib/jxl/dec_frame.cc:421-431 */
typedef enum { JXLHIP_COEFF_I16 = 0, JXLHIP_COEFF_I32 = 1 } jxlhip_coeff_type;

typedef enum {
  /* 3 planes of xsize*ysize floats (plane stride = out_plane_stride floats):
     the frame after the loop filters, still in XYB (ColorSpace::kXYB output) */
  JXLHIP_OUT_XYB_PLANAR = 0,
  /* interleaved linear RGB float, 3 floats per pixel, row stride in bytes
     given at decode time: what XYBStage + WriteToOutputStage(float) produce
     (lib/jxl/render_pipeline/stage_xyb.cc:42-98, stage_write.cc:334-368) */
  JXLHIP_OUT_LINEAR_RGB_F32 = 1,
  /* interleaved RGB / RGBA samples after the colour-encoding and packing
     stages, as described by jxlhip_frame_params::out_format: what
     FromLinearStage + WriteToOutputStage produce for a buffer output
     (lib/jxl/render_pipeline/stage_from_linear.cc:34-155,
     stage_write.cc:254-700, JxlPixelFormat in include/jxl/types.h).  Row
     stride in bytes given at decode time. */
  JXLHIP_OUT_PACKED = 2
} jxlhip_output_kind;

/* Transfer function FromLinearStage applies (the output ColorEncoding's tf;
 * lib/jxl/cms/transfer_functions-inl.h). */
typedef enum jxlhip_transfer {
  JXLHIP_TF_LINEAR = 0, /* OpLinear: samples stay linear */
  JXLHIP_TF_SRGB = 1,   /* OpRgb: TF_SRGB::EncodedFromDisplay (JXL_HIGH_PRECISION) */
  JXLHIP_TF_PQ = 2,     /* OpPq: TF_PQ(tf_param = intensity target in nits)::EncodedFromDisplay */
  JXLHIP_TF_709 = 3,    /* Op709: TF_709::EncodedFromDisplay */
  JXLHIP_TF_GAMMA = 4,  /* OpGamma: x <= 1e-5 ? 0 : FastPowf(x, tf_param = inverse gamma; DCI: 1/2.6) */
  JXLHIP_TF_HLG = 5     /* OpHlg: HlgOOTF::ToSceneLight(tf_param = intensity target, luminances) on the
                           pixel, then TF_HLG::EncodedFromDisplay per sample */
} jxlhip_transfer;

/* JxlDataType of the output buffer (include/jxl/types.h:40-60). */
typedef enum jxlhip_sample_type {
  JXLHIP_SAMPLE_F32 = 0, /* JXL_TYPE_FLOAT */
  JXLHIP_SAMPLE_U8 = 1,  /* JXL_TYPE_UINT8: x (2^bits-1), ordered dither, clamp, round */
  JXLHIP_SAMPLE_U16 = 2, /* JXL_TYPE_UINT16: x (2^bits-1), clamp, round */
  JXLHIP_SAMPLE_F16 = 3  /* JXL_TYPE_FLOAT16 */
} jxlhip_sample_type;

/* Mirrors the members of jxl::ImageOutput / JxlPixelFormat the write stage
 * reads (lib/jxl/dec_cache.h:70-82). */
typedef struct jxlhip_output_format {
  uint32_t transfer;        /* jxlhip_transfer */
  uint32_t sample_type;     /* jxlhip_sample_type */
  uint32_t num_channels;    /* 3 = RGB, 4 = RGBA (alpha = 1.0: the frame has no alpha channel) */
  uint32_t bits_per_sample; /* U8: 1..8, U16: 1..16; ignored for the float types */
  uint32_t swap_endianness; /* U16 / F16 / F32: byte-swap every sample (JXL_BIG_ENDIAN on this host) */
  float tf_param;           /* PQ, HLG: OutputEncodingInfo::desired_intensity_target; GAMMA: inverse_gamma */
  float luminances[3];      /* HLG: OutputEncodingInfo::luminances, the Y row of the output primaries'
                               RGB->XYZ matrix (dec_xyb.cc:187-211; sRGB: 0.2126, 0.7152, 0.0722) */
} jxlhip_output_format;

/* Mirrors jxl::LoopFilter (lib/jxl/loop_filter.h:20-70); values as decoded. */
typedef struct jxlhip_loop_filter {
  uint32_t gab;          /* LoopFilter::gab */
  float gab_weights[6];  /* gab_{x,y,b}_weight{1,2}: x1,x2,y1,y2,b1,b2 */
  uint32_t epf_iters;    /* 0..3 */
  float epf_sharp_lut[8];
  float epf_channel_scale[3];
  float epf_quant_mul;
  float epf_pass0_sigma_scale;
  float epf_pass2_sigma_scale;
  float epf_border_sad_mul;
} jxlhip_loop_filter;

/* The scalar members of PassesSharedState / PassesDecoderState the hot path
 * reads (lib/jxl/passes_state.h:48-95, lib/jxl/dec_cache.h:86-229). */
typedef struct jxlhip_frame_params {
  uint32_t xsize, ysize;       /* FrameDimensions::xsize/ysize (true size) */
  uint32_t coeff_type;         /* jxlhip_coeff_type */
  uint32_t output_kind;        /* jxlhip_output_kind */
  int32_t global_scale;        /* Quantizer::global_scale_ (quantizer.h:82-85) */
  int32_t quant_dc;            /* Quantizer::quant_dc_ */
  float x_dm_multiplier;       /* 0.8^(x_qm_scale-2), dec_cache.h:161 */
  float b_dm_multiplier;       /* 0.8^(b_qm_scale-2), dec_cache.h:162 */
  float quant_biases[4];       /* OpsinParams::quant_biases (dec_xyb.h:27-33) */
  float cfl_base_x;            /* ColorCorrelation::base_correlation_x_ */
  float cfl_base_b;            /* ColorCorrelation::base_correlation_b_ */
  uint32_t cfl_color_factor;   /* ColorCorrelation::color_factor_ (default 84) */
  jxlhip_loop_filter lf;
  float opsin_biases[3];       /* OpsinParams::opsin_biases (negated biases) */
  /* row-major inverse opsin absorbance matrix ALREADY multiplied by
     255/intensity_target (InitSIMDInverseMatrix, opsin_params.cc:35-45) */
  float inverse_opsin_matrix[9];
  /* Multi-GPU: this context decodes AC-group rows [stripe_group_y0,
     stripe_group_y0+stripe_group_rows) of the frame.  0,0 = whole frame. */
  uint32_t stripe_group_y0;
  uint32_t stripe_group_rows;
  /* output_kind == JXLHIP_OUT_PACKED only */
  jxlhip_output_format out_format;
  /* PassesDecoderState::used_acs (dec_cache.h:120): bit s set = raw strategy s
     occurs in the frame; known once the DC groups are decoded.  A hint: transform
     kernels of families without a set bit are not launched.  0 = unknown (everything
     is launched; a strategy missing from a non-zero mask is NOT decoded). */
  uint32_t used_acs;
  /* ImageMetadata::orientation to UNDO while writing (PassesDecoderState::undo_orientation, dec_cache.h:124;
     WriteToOutputStage's flip_x / flip_y / transpose, stage_write.cc:441-457,486,664-680 -- what JxlDecoder does
     unless JxlDecoderSetKeepOrientation).  0 or 1 = write in coded orientation.  2..8: jxlhip_decode_frame /
     jxlhip_decode_frame_host write DISPLAY orientation; for 5..8 the output is ysize pixels wide and xsize rows
     high (strides refer to that shape).  Interleaved outputs only (JXLHIP_OUT_LINEAR_RGB_F32, JXLHIP_OUT_PACKED;
     the 8-bit dither pattern follows the flipped coordinates like the reference's); not with stripes / several
     devices, not with the split calls (JXLHIP_ERR_UNSUPPORTED). */
  uint32_t undo_orientation;
} jxlhip_frame_params;

/* Device pointers of one frame's inputs.  Same content the reference keeps in
 * PassesSharedState (ac_strategy, raw_quant_field, epf_sharpness, cmap, dc) and
 * in PassesDecoderState::coefficients (ACImage).  All frame-sized planes are
 * dense row-major with row stride xsize_blocks (resp. xsize_tiles) elements
 * and cover the WHOLE frame, also when the context only decodes a stripe;
 * coeffs[c] likewise holds all groups (only the stripe's groups are read). */
typedef struct jxlhip_frame_inputs {
  const void* coeffs[3];        /* int16_t or int32_t [num_groups*65536] */
  const uint8_t* ac_strategy;   /* (raw_strategy<<1)|is_first, ac_strategy.h:187-198 */
  const int32_t* raw_quant;     /* 1..256, valid at first blocks (dec_modular.cc:552) */
  const uint8_t* epf_sharpness; /* 0..7 per block */
  const int8_t* ytox_map;       /* per 64x64 tile */
  const int8_t* ytob_map;
  const float* dc[3];           /* dequantized (+smoothed) DC, per block */
  const float* dequant_table;   /* DequantMatrices::table_ layout, see
                                   jxlhip_dequant_table_offset() */
} jxlhip_frame_inputs;

typedef struct jxlhip_ctx jxlhip_ctx;

/* ---- static geometry helpers (host, no device needed) ------------------- */
/* AcStrategy::covered_blocks_x/y, log2_covered_blocks (ac_strategy.h:148-173) */
JXLHIP_EXPORT int jxlhip_covered_blocks_x(int raw_strategy);
JXLHIP_EXPORT int jxlhip_covered_blocks_y(int raw_strategy);
JXLHIP_EXPORT int jxlhip_log2_covered_blocks(int raw_strategy);
/* kAcStrategyToQuantTableMap (quant_weights.h:337-348) */
JXLHIP_EXPORT int jxlhip_quant_table_of_strategy(int raw_strategy);
/* DequantMatrices::Matrix(kind,c) - table_ (quant_weights.h:364-367); floats */
JXLHIP_EXPORT size_t jxlhip_dequant_table_offset(int raw_strategy, int c);
JXLHIP_EXPORT const char* jxlhip_status_string(int status);

/* ---- context ------------------------------------------------------------ */
/* Replaces the per-decoder state setup of PassesDecoderState::Init
 * (dec_cache.h:153-229).  device = HIP device ordinal. */
JXLHIP_EXPORT int jxlhip_create(int device, jxlhip_ctx** out);
/* JxlMemoryManager (lib/include/jxl/memory_manager.h:45-63), restated so that this header stands alone; the
 * layout is the reference's.  Both callbacks or none (lib/threads/thread_parallel_runner.cc:37-53): exactly
 * one NULL is JXLHIP_ERR_INVALID_ARGUMENT. */
typedef struct JxlMemoryManagerHip {
  void* opaque;
  void* (*alloc)(void* opaque, size_t size);
  void (*free)(void* opaque, void* address);
} JxlMemoryManagerHip;
/* The same with the caller's memory manager: the context object and the pinned staging slots of
 * the entropy decoder (pinned in place with hipHostRegister) come from it; device memory comes from hipMalloc,
 * and the bookkeeping of the C++ containers inside the context from the C++ runtime. */
JXLHIP_EXPORT int jxlhip_create_ex(int device, const JxlMemoryManagerHip* memory_manager, jxlhip_ctx** out);
/* One context over SEVERAL devices of this process ("groups shard across the GPUs of one
 * node"): the frame is cut into contiguous stripes of AC-group rows (the first ysg % ndev stripes one row taller),
 * stripe i lives on devices[i].  A device may be listed more than once (several stripes on one GPU).
 * Supported on a multi context: jxlhip_frame_begin, jxlhip_upload_side_info, jxlhip_submit_group,
 * jxlhip_ac_group(s)_decode_submit[_passes] (a group goes to the device of its stripe),
 * jxlhip_decode_frame (out: device memory of devices[0]; the other stripes arrive by peer copies = the gather),
 * jxlhip_decode_frame_host (every device copies its own stripe to the host rows: no gather), jxlhip_halo_rows,
 * jxlhip_sync, jxlhip_last_error, jxlhip_destroy; everything else is JXLHIP_ERR_UNSUPPORTED.
 * Between the two phases each stripe's LoopFilter::Padding() boundary rows go to its neighbours with
 * hipMemcpyPeerAsync (xGMI when peer access is available), ordered by events only: the host never waits inside
 * a frame.  Replaces GroupBorderAssigner / SaveBorders / LoadBorders across devices
 * (lib/jxl/dec_group_border.cc:68-187) and the data-parallel contract of lib/jxl/base/data_parallel.h:50-78. */
JXLHIP_EXPORT int jxlhip_create_multi(const int* devices, int num_devices,
                                      const JxlMemoryManagerHip* memory_manager, jxlhip_ctx** out);
JXLHIP_EXPORT void jxlhip_destroy(jxlhip_ctx* ctx);
JXLHIP_EXPORT const char* jxlhip_last_error(const jxlhip_ctx* ctx);
/* The library samples its debug / test switches (JXLHIP_WP_GENERAL, JXLHIP_NO_PIPELINE, JXLHIP_TEST_RANGE_GROUP,
 * JXLHIP_CODESTREAM_VERBOSE, JXLHIP_MULTI_INTERIOR_FIRST, JXLHIP_DC_TREE, and the kernels' launch-geometry knobs
 * JXLHIP_FUSED_PC, JXLHIP_FUSED_PC_RH, JXLHIP_FUSED_PC_ROLE, JXLHIP_FUSED_PC0_ROLE, JXLHIP_FUSED_TILES, JXLHIP_BIG_WGS,
 * JXLHIP_DEBUG) from the environment when a context is created (jxlhip_create / _ex / _multi) and at their first use
 * before that; a test that changes one of them under a live context calls this to have them read again.  Decoding and
 * kernel launches never call getenv (no reference counterpart: libjxl has no run-time switches on this path). */
JXLHIP_EXPORT void jxlhip_debug_reload_env(void);
/* external != 0: all launches go to the caller's hipStream_t `hip_stream`
 * (NULL = the device's default stream, which is what torch's default stream
 * is); external == 0: back to the context's own non-blocking stream. */
JXLHIP_EXPORT int jxlhip_set_stream(jxlhip_ctx* ctx, void* hip_stream,
                                    int external);

/* Replaces PassesDecoderState::InitForAC + PreparePipeline
 * (dec_cache.cc:78-96,117-371): fixes the frame geometry and the stage list
 * and (re)allocates the device intermediates (XYB planes, sigma image, block
 * offset table). */
JXLHIP_EXPORT int jxlhip_frame_begin(jxlhip_ctx* ctx,
                                     const jxlhip_frame_params* params);

/* Zero-copy path: the caller already has the inputs in device memory. */
JXLHIP_EXPORT int jxlhip_frame_set_inputs(jxlhip_ctx* ctx,
                                          const jxlhip_frame_inputs* dev);

/* Host-upload path (what a libjxl FrameDecoder would call).
 * Replaces the reads of shared->ac_strategy/raw_quant_field/epf_sharpness/
 * cmap/dc in DecodeGroupImpl (dec_group.cc:275-316).  Host pointers, dense
 * frame-sized planes; copied asynchronously into context-owned HBM. */
JXLHIP_EXPORT int jxlhip_upload_side_info(
    jxlhip_ctx* ctx, const uint8_t* ac_strategy, const int32_t* raw_quant,
    const uint8_t* epf_sharpness, const int8_t* ytox_map,
    const int8_t* ytob_map, const float* const dc[3],
    const float* dequant_table);
/* The frame's alpha channel for 4-channel JXLHIP_OUT_PACKED outputs: a dense host plane of xsize x ysize floats
 * (stride_floats per row; 1.0 = opaque), copied into context-owned HBM.  It is written to the output like a
 * colour channel without the transfer function -- what WriteToOutputStage does with input channel alpha_c
 * (stage_write.cc:350-366); not un-premultiplied.  Without this call (jxlhip_frame_begin resets it) alpha is the
 * opaque 1.0 the reference substitutes (:355-360).  On a multi-device context every stripe takes its own rows. */
JXLHIP_EXPORT int jxlhip_set_alpha(jxlhip_ctx* ctx, const float* host_plane, size_t stride_floats);
/* A pinned host plane of the current frame's size, owned by the context, for the caller to fill and hand to
 * jxlhip_set_alpha (the copy is then a true asynchronous DMA).  Valid until the next jxlhip_frame_begin of a larger
 * frame or jxlhip_destroy; the previous frame must have been synchronised before it is written again. */
JXLHIP_EXPORT int jxlhip_alpha_staging(jxlhip_ctx* ctx, float** plane, size_t* stride_floats);
/* Replaces GetBlockFromEncoder/GetBlockFromBitstream -> DequantBlock hand-off
 * (dec_group.cc:334-359,662-706): the group's quantized coefficient stream, as
 * produced by DecodeACVarBlock, ncoeffs <= 65536 elements per channel.
 * Thread-safe w.r.t. other groups; copies on a pooled stream. */
JXLHIP_EXPORT int jxlhip_submit_group(jxlhip_ctx* ctx, uint32_t group_idx,
                                      const void* const coeffs[3],
                                      size_t ncoeffs);

/* Phase 1, replaces DecodeGroupImpl's DequantBlock + TransformToPixels for
 * every group of the stripe (dec_group.cc:431-450) and ComputeSigma
 * (epf.cc:39-133).  Result: XYB planes in context memory. */
JXLHIP_EXPORT int jxlhip_decode_blocks(jxlhip_ctx* ctx);

/* Multi-GPU halo hand-off between phase 1 and 2.  Number of rows a stripe
 * needs from each neighbour = LoopFilter::Padding() (loop_filter.h:26-29):
 * the GPU-side replacement of GroupBorderAssigner / SaveBorders / LoadBorders
 * (lib/jxl/dec_group_border.cc:68-187, low_memory_render_pipeline.cc:832-934). */
JXLHIP_EXPORT int jxlhip_halo_rows(const jxlhip_ctx* ctx);
/* The XYB planes are kept block-major (8x8 tiles) in context memory, so halo
 * rows travel through dense staging buffers: dev points to 3 * halo * xsize
 * floats (channel-major, then row, then x) in device memory.
 * export which: 0 = this stripe's first `halo` rows (to send UP),
 *               1 = its last `halo` rows (to send DOWN);
 * import which: 0 = rows received from the stripe ABOVE (placed just above
 *               this stripe), 1 = rows received from BELOW.
 * Both are asynchronous on the context's stream. */
JXLHIP_EXPORT int jxlhip_halo_export(jxlhip_ctx* ctx, int which, float* dev);
JXLHIP_EXPORT int jxlhip_halo_import(jxlhip_ctx* ctx, int which,
                                     const float* dev);

/* Phase 2, replaces the render pipeline stages Gaborish/EPF0/EPF1/EPF2/XYB
 * (+ float WriteToOutput) run by RenderPipeline::InputReady ->
 * ProcessBuffers (low_memory_render_pipeline.cc:832-934) with the
 * SimpleRenderPipeline border semantics (simple_render_pipeline.cc:129-164).
 * out: device pointer; for LINEAR_RGB_F32 out_stride = bytes per row
 * (>= xsize*12), for XYB_PLANAR out_stride = floats per row and the 3 planes
 * are out_plane_stride floats apart.  Rows written: the stripe's rows only
 * (row 0 of `out` = first row of the stripe). */
JXLHIP_EXPORT int jxlhip_decode_filters(jxlhip_ctx* ctx, void* out,
                                        size_t out_stride,
                                        size_t out_plane_stride);
/* Phase 2 for the frame rows [y_begin, y_end) of the stripe only (`out` still names the stripe's first row).  What a
 * stripe with neighbours does while its halo rows travel: the rows whose filter support stays inside the stripe -- all
 * but the first / last block row -- need no halo (the reference hands a group's border rows to its neighbours without
 * waiting for them either, dec_group_border.cc:68-187; stage borders: loop_filter.h:26-29), the two boundary block
 * rows follow once jxlhip_halo_import has run.  y_begin and y_end must be multiples of 8 or the stripe's own first /
 * last row; the pixels do not depend on how the rows are cut. */
JXLHIP_EXPORT int jxlhip_decode_filters_rows(jxlhip_ctx* ctx, void* out, size_t out_stride, size_t out_plane_stride,
                                             uint32_t y_begin, uint32_t y_end);
/* One stripe step in three calls (what libjxl_amd/stripes.py enqueues per rank and frame): jxlhip_stripe_begin = phase 1
 * (jxlhip_decode_blocks) + the stripe's boundary rows into send_up / send_down (jxlhip_halo_export; NULL = no neighbour
 * on that side); the caller then posts its sends / receives and filters the interior rows with
 * jxlhip_decode_filters_rows while they travel; jxlhip_stripe_finish = the neighbours' rows installed
 * (jxlhip_halo_import) + phase 2 of the rows outside [y_interior_begin, y_interior_end) (equal: of every row).
 * Reference: the neighbour hand-off of lib/jxl/dec_group_border.cc:68-187 around the render pipeline. */
JXLHIP_EXPORT int jxlhip_stripe_begin(jxlhip_ctx* ctx, float* send_up, float* send_down);
JXLHIP_EXPORT int jxlhip_stripe_finish(jxlhip_ctx* ctx, const float* recv_up, const float* recv_down, void* out,
                                       size_t out_stride, size_t out_plane_stride, uint32_t y_interior_begin,
                                       uint32_t y_interior_end);

/* Both phases (single GPU).  When the context holds the whole frame (no stripe)
 * and the stage list has at most two EPF passes this runs FUSED
 * (kernels_fused.hip): the filter kernel decodes the DCT8 varblocks of its own
 * window straight from the coefficient stream, those pixels never exist in
 * HBM, and the XYB planes afterwards hold the other strategies' pixels only --
 * jxlhip_export_xyb / jxlhip_halo_export need jxlhip_decode_blocks.
 * JXLHIP_FUSE=0 in the environment forces the two-phase path. */
JXLHIP_EXPORT int jxlhip_decode_frame(jxlhip_ctx* ctx, void* out,
                                      size_t out_stride,
                                      size_t out_plane_stride);

/* The same into HOST memory (the buffer of JxlDecoderSetImageOutBuffer,
 * lib/include/jxl/decode.h:1100-1140; ImageOutput::buffer / stride,
 * lib/jxl/dec_cache.h:70-82): decodes into a context-owned device frame, copies
 * it out with one strided device-to-host transfer and synchronises (returns
 * what jxlhip_sync would).  Strides as for jxlhip_decode_frame, in host terms. */
JXLHIP_EXPORT int jxlhip_decode_frame_host(jxlhip_ctx* ctx, void* host_out,
                                           size_t out_stride,
                                           size_t out_plane_stride);

/* The same into a context-owned PINNED host frame (grown on demand, reused by later frames, freed with the
 * context; taken from the caller's JxlMemoryManager when there is one): returns its address and row stride in
 * bytes once the pixels are there.  The source for a row consumer -- JxlDecoderSetImageOutCallback /
 * JxlDecoderSetMultithreadedImageOutCallback (lib/include/jxl/decode.h:1040-1100; how WriteToOutputStage
 * hands row runs to PixelCallback::run, lib/jxl/render_pipeline/stage_write.cc:662-700): the device-to-host
 * transfer is one DMA into pinned memory and the callbacks then read it in place.  Interleaved outputs
 * (JXLHIP_OUT_LINEAR_RGB_F32, JXLHIP_OUT_PACKED); valid until the next decode call on this context. */
JXLHIP_EXPORT int jxlhip_decode_frame_pinned(jxlhip_ctx* ctx, const void** host_frame, size_t* stride);

JXLHIP_EXPORT int jxlhip_sync(jxlhip_ctx* ctx);

/* Debug/test taps on context-owned intermediates (device pointers).
 * export_xyb: row-major copy of the phase-1 result, rows [first row of the
 * stripe, + its block-padded height) x xsize_blocks*8 columns, row stride
 * dst_stride floats. */
JXLHIP_EXPORT int jxlhip_export_xyb(jxlhip_ctx* ctx, float* const dst[3],
                                    size_t dst_stride);
JXLHIP_EXPORT int jxlhip_get_sigma(jxlhip_ctx* ctx, float** inv_sigma,
                                   size_t* row_stride);

/* Per-kernel timing with HIP events on the launch stream.  enable!=0 starts
 * recording around every launch of subsequent decode calls;
 * jxlhip_profile_read syncs and returns accumulated milliseconds and launch
 * counts per kernel slot (see JXLHIP_KERNEL_*), then resets. */
enum {
  JXLHIP_KERNEL_PREPARE = 0,  /* block-offset scan, work lists, sigma */
  JXLHIP_KERNEL_BLOCKS = 1,   /* dequant+CfL+LLF+inverse transforms: one launch
                                 per strategy class, overlapped on several
                                 streams; the span covers all of them */
  JXLHIP_KERNEL_FILTERS = 2,  /* Gaborish/EPF/XYB->RGB row march reading the planes (k_filters_fast / k_filters) */
  JXLHIP_KERNEL_FUSED = 3,    /* ... fed from the coefficient stream instead (k_fused: whole frames, see
                                 jxlhip_decode_frame); a frame has a FILTERS or a FUSED span, never both */
  JXLHIP_KERNEL_EPF0 = 4,     /* epf_iters == 3: [Gaborish] + EPF0 into the second plane set (k_epf0); the EPF1 +
                                 EPF2 + output march that follows is the FILTERS span */
  JXLHIP_KERNEL_COUNT = 8
};
/* A hint, not a contract: `frames_in_flight` = how many contexts the caller keeps busy on this device at the same time
 * (a pool of decoders over a queue of images; 1 = this context runs alone, the default).  It only moves the frame size
 * from which jxlhip_decode_frame takes the fused kernel: alone, fusing pays from 12 Mpx (a fused wave's head / tail rows
 * against the plane traffic it saves); with several frames in flight the device is bound by HBM
 * traffic, which the fused path has less of, and it pays from 6 Mpx (4K d1.0, three in flight: 83 -> 105 Gpx/s; 1080p
 * stays two-phase: 77 vs 67).  The pixels do not depend on it beyond the rounding difference between the two paths
 * (both within the parity bar).  Nothing like it in libjxl. */
JXLHIP_EXPORT int jxlhip_set_concurrency_hint(jxlhip_ctx* ctx, int frames_in_flight);
JXLHIP_EXPORT int jxlhip_profile_enable(jxlhip_ctx* ctx, int enable);
JXLHIP_EXPORT int jxlhip_profile_read(jxlhip_ctx* ctx,
                                      float ms[JXLHIP_KERNEL_COUNT],
                                      uint32_t launches[JXLHIP_KERNEL_COUNT]);

/* ---- device-side helpers for rows of SURVEY 8(a) outside the two phases -- */
/* a5: DequantMatrices (quant_weights.h:350-428).  One parameter set per
 * quant table (17 of them, QuantTable order: DCT, IDENTITY, DCT2X2, DCT4X4,
 * DCT16X16, DCT32X32, DCT8X16, DCT8X32, DCT16X32, DCT4X8, AFV0, DCT64X64,
 * DCT32X64, DCT128X128, DCT64X128, DCT256X256, DCT128X256) mirroring
 * QuantEncodingInternal (quant_weights.h:57-187) with the values as the
 * decoder holds them (i.e. after Decode's *64 scalings, quant_weights.cc:373-
 * 470).  jxlhip_dequant_encodings_decode (jxl_hip_entropy.h) fills it from the
 * AC-global section. */
typedef enum {
  JXLHIP_QUANT_LIBRARY = 0, /* the default parameters of the table's kind */
  JXLHIP_QUANT_ID = 1,      /* weights[c][0..2] */
  JXLHIP_QUANT_DCT2 = 2,    /* weights[c][0..5] */
  JXLHIP_QUANT_DCT4 = 3,    /* weights[c][0..1] multipliers + bands */
  JXLHIP_QUANT_DCT4X8 = 4,  /* weights[c][0] multiplier + bands */
  JXLHIP_QUANT_AFV = 5,     /* weights[c][0..8] + bands (4x8) + bands_afv_4x4 */
  JXLHIP_QUANT_DCT = 6,     /* bands */
  JXLHIP_QUANT_RAW = 7      /* modular-coded explicit table: NOT supported */
} jxlhip_quant_mode;
#define JXLHIP_NUM_QUANT_TABLES 17
#define JXLHIP_MAX_DISTANCE_BANDS 17
typedef struct jxlhip_quant_encoding {
  uint32_t mode;               /* jxlhip_quant_mode */
  uint32_t num_bands;          /* DctQuantWeightParams::num_distance_bands */
  uint32_t num_bands_afv_4x4;  /* dct_params_afv_4x4 (AFV only) */
  uint32_t reserved;
  float bands[3][JXLHIP_MAX_DISTANCE_BANDS];
  float bands_afv_4x4[3][JXLHIP_MAX_DISTANCE_BANDS];
  float weights[3][9];
} jxlhip_quant_encoding;

/* DequantMatrices::EnsureComputed (quant_weights.cc:163-358,1211-1271): fills a
 * JXLHIP_DEQUANT_TABLE_FLOATS device buffer with the dequant tables of the 17
 * encodings (NULL = all JXLHIP_QUANT_LIBRARY).  Asynchronous on the context's
 * stream like the decode calls; parameters that give a weight outside
 * [1e-8, 1e8) (the reference's "Invalid quantization table",
 * quant_weights.cc:329-339) surface as JXLHIP_ERR_BAD_STREAM from the next
 * jxlhip_sync().  JXLHIP_ERR_UNSUPPORTED for JXLHIP_QUANT_RAW. */
JXLHIP_EXPORT int jxlhip_dequant_tables(jxlhip_ctx* ctx,
                                        const jxlhip_quant_encoding* encodings,
                                        float* table_dev);
/* = jxlhip_dequant_tables(ctx, NULL, table_dev) */
JXLHIP_EXPORT int jxlhip_default_dequant_tables(jxlhip_ctx* ctx,
                                                float* table_dev);
/* a8: DequantDC + AdaptiveDCSmoothing (compressed_dc.cc:128-250), 4:4:4.
 * quant_dc[c]: int32 planes from the modular DC decode (device), dc_out[c]:
 * float planes (device), both xsize_blocks*ysize_blocks dense.
 * dc_quant = DequantMatrices::DCQuant(c) (quant_weights.h:289-299; NULL = the
 * defaults 1/4096, 1/512, 1/256); the step is inv_global_scale / quant_dc *
 * dc_quant[c] (quantizer.h:133-139) with the current frame's global_scale and
 * quant_dc.  cfl_*_dc = ColorCorrelation::YtoXRatio(ytox_dc) / YtoBRatio.
 * smooth != 0 runs AdaptiveDCSmoothing afterwards. */
JXLHIP_EXPORT int jxlhip_dequant_dc(jxlhip_ctx* ctx,
                                    const int32_t* const quant_dc[3],
                                    float* const dc_out[3],
                                    const float dc_quant[3], float cfl_x_dc,
                                    float cfl_b_dc, int smooth);
/* The same with the per-DC-group multiplier 1 / (1 << extra_precision) that
 * DecodeVarDCTDC reads in front of every DC group (dec_modular.cc:443-445) and
 * DequantDC applies to the three factors of that group's rectangle
 * (compressed_dc.cc:207-209) -- NOT to the AdaptiveDCSmoothing thresholds, which
 * use the plain factors (FinalizeDC, dec_frame.cc:344-357).  extra_precision: HOST array,
 * one byte (0..3, as jxlhip_dc_group_decode returns it) per DC group
 * (2048x2048 px) in raster order, ceil(xsize_blocks/256) per row; NULL = all 0. */
JXLHIP_EXPORT int jxlhip_dequant_dc_groups(jxlhip_ctx* ctx,
                                           const int32_t* const quant_dc[3],
                                           float* const dc_out[3],
                                           const float dc_quant[3],
                                           float cfl_x_dc, float cfl_b_dc,
                                           int smooth,
                                           const uint8_t* extra_precision);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_H_ */
