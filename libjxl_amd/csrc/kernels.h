// kernels.h -- host-callable launchers of the HIP kernels (internal).
#ifndef JXLHIP_KERNELS_H_
#define JXLHIP_KERNELS_H_

#include "dev_common.h"

namespace jxlhip {

struct WorkLists {
  WorkItem* list[kNumClasses];
  uint32_t* count;  // kCountStride counters (classes + unit tickets), zeroed before k_prepare
};

struct SharpLut {
  float v[8];
};

// Stage parameters of phase 2, precomputed on the host exactly as the
// reference stages do in their constructors / per-row prologues.
struct FilterParams {
  float gab_w[3][3];     // per channel: normalised w0, w1, w2 (stage_gaborish.cc:36-53)
  float ch_scale[3];     // epf_channel_scale
  float sm[3], bsm[3];   // per EPF stage: sigma multiplier inside / on 8x8 borders
  float opsin_bias[3];   // OpsinParams::opsin_biases
  float cbrt_bias[3];    // cbrt(opsin_biases)
  float minv[9];         // inverse opsin matrix * 255/intensity_target
  float xyb_bias[6];     // -cbrt_bias[0..2], opsin_bias[0..2] (kernels_filters_fast.hip)
  float mcol[3][4];      // its columns, wrapped: (m[j], m[3+j], m[6+j], m[j]) (kernels_filters_fast.hip)
  void* out;
  size_t out_stride;        // RGB / packed: bytes per row; XYB: floats per row
  size_t out_plane_stride;  // XYB only
  // JXLHIP_OUT_PACKED: FromLinearStage + WriteToOutputStage parameters
  jxlhip_output_format fmt;
  float sample_mul;         // 2^bits_per_sample - 1
  float tf_scale;           // PQ: intensity_target / 10000; GAMMA: inverse gamma
  float hlg_exponent;       // HLG: HlgOOTF exponent (gamma - 1), 0 = OOTF not applied
  const float* dither;      // 32x32 pattern (device)
  // dither coordinates = (dither_x0 + dither_xs * x, dither_y0 + dither_ys * y): the reference dithers AFTER
  // undo_orientation's flips (stage_write.cc:486-492): identity = (0, 1, 0, 1)
  int32_t dither_x0, dither_xs, dither_y0, dither_ys;
  // 4-channel packed output: the frame's alpha channel (floats, image coordinates, alpha_stride floats per row);
  // nullptr = the opaque 1.0 the reference substitutes (stage_write.cc:355-360)
  const float* alpha;
  uint32_t alpha_stride;
};

void LaunchPrepare(const DevFrame& f, const WorkLists& wl, int with_sigma, float epf_quant_mul,
                   const SharpLut& lut, hipStream_t st);
// Five launches (k_dct8, the row-per-lane families R16 / R32, family A, the large kinds) on
// streams[0] / streams[1 % nstreams]; `cells` = block cells of the band (bounds the unit count).
// emit: see LaunchMfma32 (nullptr: every class writes the XYB planes).
void LaunchBlocks(const DevFrame& f, const WorkLists& wl, uint32_t cells, const float* wc,
                  const float* resample, hipStream_t* streams, int nstreams, const FilterParams* emit = nullptr);
// Returns 0, or -1 when the (gab, epf_iters, output_kind) combination is invalid.
int LaunchFilters(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                  int output_kind, hipStream_t st);

// Register/DPP kernel for stage lists with at most one EPF pass; returns false
// when the configuration is not covered (caller falls back to LaunchFilters).
bool LaunchFiltersFast(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                       int output_kind, hipStream_t st);
// the packed formats LaunchFiltersFast has a kernel with the format fixed at compile time for
bool FastFixedFormat(const jxlhip_output_format& o);

// epf_iters = 3 (kernels_epf0.hip): [Gaborish] + EPF0 from f.xyb into a second plane set; the EPF1 + EPF2 march
// (LaunchFiltersFast with gab = 0, epf_iters = 2 on those planes) follows.  false: geometry not covered.
bool LaunchEpf0(const DevFrame& f, const FilterParams& p, int gab, float* const dst[3], hipStream_t st);

// Sparse coefficient hand-off (kernels_tables.hip k_expand_sparse): group g of [g0, g0 + n) whose entry in `offsets`
// is not 0xFFFFFFFF is expanded from `sparse + 16 * offsets[g]` -- three counts + pad, then the (position << 16 |
// value) lists of the three channels back to back -- into the dense int16 block stream `dense`
// ([group][channel][65536], zero-filled first)
void LaunchExpandSparse(const uint8_t* sparse, const uint32_t* offsets, int16_t* dense, uint32_t g0, uint32_t n, hipStream_t st);

// undo_orientation (kernels_tables.hip k_orient): coded xsize x ysize pixels of bytes_per_pixel -> display orientation
bool LaunchOrient(const void* src, size_t src_stride, uint32_t xsize, uint32_t ysize, uint32_t bytes_per_pixel,
                  uint32_t orientation, void* dst, size_t dst_stride, hipStream_t st);

// Matrix-core 32x32 IDCT (kernels_mfma.hip), opt-in through DevFrame::mfma32
void MfmaDct32Constants(float* host /* 2048 floats */);
void MfmaDct16Constants(float* host /* 256 floats */);
// emit != nullptr: the frame is DCT32X32 only and has no loop filter -- the kernel writes linear float RGB to
// emit->out itself (rows f.y0 .. f.y1) instead of XYB planes
void LaunchMfma32(const DevFrame& f, const WorkLists& wl, uint32_t cells, hipStream_t st, const FilterParams* emit = nullptr);
// Matrix-core 16x16 IDCT, opt-in through DevFrame::mfma16
void LaunchMfma16(const DevFrame& f, const WorkLists& wl, uint32_t cells, hipStream_t st);

// Fused kernel (kernels_fused.hip): the row march fed from the coefficient stream (DCT8 decoded by the
// filter wave itself, other classes copied from the planes).  FusedSupported: frames it takes --
// decided before k_prepare, which routes the DCT8 blocks (DevFrame::fused).
bool FusedSupported(const DevFrame& f, int gab, int epf_iters, int output_kind);
bool LaunchFused(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, int output_kind,
                 hipStream_t st);
// epf_iters = 3 in fused mode (kernels_fused_epf0.hip): [Gaborish] + EPF0 marched from the producer's slab into the
// second plane set, as LaunchEpf0 does from the planes; FusedEpf0Supported: whole frames it takes (decided before
// k_prepare, like FusedSupported)
bool FusedEpf0Supported(const DevFrame& f, int gab);
bool LaunchFusedEpf0(const DevFrame& f, const FilterParams& p, int gab, float* const dst[3], hipStream_t st);
// compute units of the device the calling thread has current (hipDeviceAttributeMultiprocessorCount, cached): what the
// generation-filling launch geometries are sized from (256 on a whole MI355X, fewer on a CPX / DPX partition)
unsigned DeviceCus();

// block-major plane rows <-> dense row-major staging
void LaunchZeroU32(uint32_t* p, uint32_t n, hipStream_t st);  // (kernels_tables.hip: a kernel, for captured graphs)
void LaunchRowsCopy(const DevFrame& f, float* dense, int y_first, int nrows, int ncols,
                    size_t dense_stride, size_t dense_plane, int nch, bool to_dense,
                    hipStream_t st);

// a5 / a8 helpers
void LaunchDequantTables(float* table, const jxlhip_quant_encoding* enc_dev, int32_t* status,
                         hipStream_t st);
void LaunchDequantDC(uint32_t xsb, uint32_t ysb, const int32_t* const q[3], float* const dc[3],
                     float* const tmp[3], const float mul_dc[3], float cfl_x, float cfl_b,
                     int smooth, const uint8_t* extra_precision, hipStream_t st);

}  // namespace jxlhip
#endif
