// dev_common.h -- shared between the HIP translation units of libjxl_hip.so.
// Geometry LUTs, the per-frame kernel argument block and the in-register
// 1-D transforms.  gfx950 only (wave = 64).
#ifndef JXLHIP_DEV_COMMON_H_
#define JXLHIP_DEV_COMMON_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/jxl_hip.h"

#define JXL_FMT_CONST static constexpr
#include "dct_constants.inc"
#include "format_constants.inc"

namespace jxlhip {

// ---- strategy geometry (lib/jxl/ac_strategy.h:148-173,
// lib/jxl/quant_weights.h:337-348,401-417) --------------------------------
static constexpr uint8_t kCoveredX[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1,
                                          1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
static constexpr uint8_t kCoveredY[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1,
                                          1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
static constexpr uint8_t kQuantKind[27] = {0,  1,  2,  3,  4,  5,  6,  6,  7,
                                           7,  8,  8,  9,  9,  10, 10, 10, 10,
                                           11, 12, 12, 13, 14, 14, 15, 16, 16};
static constexpr uint8_t kKindShort[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2,
                                           1, 1, 8, 4, 16, 8, 32, 16};
static constexpr uint8_t kKindLong[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4,
                                          1, 1, 8, 8, 16, 16, 32, 32};

__host__ __device__ constexpr uint32_t DequantOffset(int strategy) {
  uint32_t pos = 0;
  for (int k = 0; k < kQuantKind[strategy]; k++)
    pos += 3u * 64u * kKindShort[k] * kKindLong[k];
  return pos;
}

// Work classes: one compacted varblock list per class, built by k_prepare, so
// that every wave of the transform kernels runs ONE strategy.
static constexpr int kNumSpecial = 9;
static constexpr int kNumMedium = 11;
enum WorkClass : int {
  kClsDct8 = 0,      // strategy 0
  kClsSpecial0 = 1,  // kNumSpecial single-block kinds follow, see kSpecialStrategy
  kClsMedium0 = kClsSpecial0 + kNumSpecial,  // kNumMedium kinds, see kMediumStrategy
  kClsLarge = kClsMedium0 + kNumMedium,      // 21..26 (any side >= 128)
  kNumClasses = kClsLarge + 1
};
// counter block of one band: the list length of class c lives at
// count[c * kCounterPad] -- one 128-byte line per counter, so that the ~20
// atomics every k_prepare workgroup issues do not all serialise on one line
static constexpr uint32_t kCellFromPlanes = 0xFFFFFFFFu;
static constexpr int kCounterPad = 32;
static constexpr int kCountStride = 32 * kCounterPad;
static_assert(kNumClasses <= 32, "counter block layout");
static constexpr uint8_t kSpecialStrategy[kNumSpecial] = {1, 2, 3, 12, 13, 14, 15, 16, 17};
// medium class index -> strategy (16x8 .. 64x64)
static constexpr uint8_t kMediumStrategy[kNumMedium] = {6, 7, 4, 8, 9, 10, 11, 5, 18, 19, 20};
__host__ __device__ constexpr int ClassOfStrategy(int s) {
  if (s == 0) return kClsDct8;
  if (s >= 21) return kClsLarge;
  for (int i = 0; i < kNumSpecial; i++)
    if (kSpecialStrategy[i] == s) return kClsSpecial0 + i;
  for (int i = 0; i < kNumMedium; i++)
    if (kMediumStrategy[i] == s) return kClsMedium0 + i;
  return -1;
}
// class -> strategy LUT for device code (runtime strategy index)
struct ClassLut {
  int8_t v[27];
};
__host__ __device__ constexpr ClassLut MakeClassLut() {
  ClassLut l{};
  for (int s = 0; s < 27; s++) l.v[s] = (int8_t)ClassOfStrategy(s);
  return l;
}
static constexpr ClassLut kClassLut = MakeClassLut();
// worst-case entries per block cell of each class = 1/covered_blocks
__host__ __device__ constexpr uint32_t ClassMinCovered(int cls) {
  if (cls < kClsMedium0) return 1;
  if (cls == kClsLarge) return 128;
  const int s = kMediumStrategy[cls - kClsMedium0];
  return (uint32_t)kCoveredX[s] * kCoveredY[s];
}

// 16-byte self-contained work item: a transform kernel needs no other per-block
// side info (one dependent load less on its critical path).
struct __attribute__((aligned(16))) WorkItem {
  uint32_t pos;  // (aby << 16) | abx : absolute block coordinates
  uint32_t off;  // group*1024 + offset/64 into the coefficient stream
  uint32_t qc;   // raw_quant | (ytox & 0xff) << 16 | (ytob & 0xff) << 24 (tile of the first block)
  uint32_t pad;
};

// Per-frame kernel arguments (by value).
struct DevFrame {
  uint32_t xsize, ysize;      // true size
  uint32_t xsb, ysb;          // size in blocks
  uint32_t xsg, ysg;          // size in groups
  uint32_t xtiles;            // colour tiles per row
  uint32_t group_y0, group_rows;  // stripe (in groups)
  uint32_t y0, y1;            // stripe rows [y0, y1) in pixels (y1 clipped to ysize)
  uint32_t fy0, fy1;          // rows the filter launch writes (a band of the stripe)
  uint32_t band_g0, band_g1;  // group rows whose work lists k_prepare builds
  uint32_t halo;              // LoopFilter::Padding()
  uint32_t coeff_type;
  uint32_t used_acs;          // jxlhip_frame_params::used_acs (0 = unknown)
  uint32_t coef_stride64;     // coefficient slots / 64 between consecutive groups of coeffs[c]: 1024 (the
                              // ACImage layout of caller-owned buffers) or 3072 (the context's upload buffer,
                              // one group's three channels contiguous so that one copy moves a group)
  float inv_global_scale, quant_scale;
  float x_dm, b_dm;
  float biases[4];
  float cfl_base_x, cfl_base_b, color_scale;
  // inputs
  const void* coeffs[3];
  const uint8_t* acs;
  const int32_t* raw_quant;
  const uint8_t* sharp;
  const int8_t* ytox;
  const int8_t* ytob;
  const float* dc[3];
  const float* dequant;
  // intermediates: XYB planes in BLOCK-MAJOR layout: 8x8 tiles of 64 floats
  // (256 bytes = two full cache lines), tiles row-major with `tile_stride`
  // tiles per tile row.  Every varblock writes whole tiles, whatever its
  // strategy class, so each cache line / DRAM burst is written by one
  // workgroup at one time; the filter kernels read tile rows sequentially.
  // Rows covered: [plane_y0, plane_y0 + 8 * plane_tile_rows), plane_y0 = y0 - 8.
  float* xyb[3];
  uint32_t tile_stride;      // tiles per tile row (>= xsb)
  int32_t plane_y0;          // pixel row of the first tile row (multiple of 8; y0 - 8)
  uint32_t plane_tile_rows;
  float* inv_sigma;       // xsb*ysb, whole frame indexing
  int32_t* error_flag;
  // Fused mode (kernels_fused.hip): varblocks of the classes the fused kernel decodes itself (DCT8)
  // are not put on a work list and never reach the XYB planes; k_prepare leaves their coefficient
  // offset and quant / CfL word in cell_info[cell] (xsb*ysb entries, whole frame indexing; every
  // other cell holds kCellFromPlanes in .x: its pixels come from the planes)
  uint2* cell_info;
  uint32_t fused;
  // JXLHIP_MFMA=1 (kernels_mfma.hip): DCT32X32 varblocks go through the matrix-core kernel; operand tables
  // (2048 floats, MfmaDct32Constants), nullptr = the row-per-lane path decodes them
  const float* mfma32;
  // the same for DCT16X16 (256 floats, MfmaDct16Constants)
  const float* mfma16;
  // != 0: xyb[] are plain row-major planes of this many bytes per row, first row plane_y0 (k_epf0's output
  // as the EPF1 + EPF2 march reads it, filters_march.h SRC_LINEAR)
  uint32_t linear_stride;
  // != nullptr: k_prepare's first workgroup zeroes these kCountStride counters -- the counter block the NEXT frame's
  // k_prepare will use (two blocks alternate), which saves a memset launch per frame
  uint32_t* zero_counts;
};
// address of pixel (y, x) of channel c in the block-major planes
__device__ __forceinline__ size_t PlaneOffset(const DevFrame& f, int y, int x) {
  const uint32_t ry = (uint32_t)(y - f.plane_y0);
  return ((size_t)(ry >> 3) * f.tile_stride + ((uint32_t)x >> 3)) * 64u + ((ry & 7u) << 3) +
         ((uint32_t)x & 7u);
}
__device__ __forceinline__ float* PlanePtr(const DevFrame& f, int c, uint32_t y, uint32_t x) {
  return f.xyb[c] + PlaneOffset(f, (int)y, (int)x);
}
// first float of the tile holding block (aby, abx)
__device__ __forceinline__ float* TilePtr(const DevFrame& f, int c, uint32_t aby, uint32_t abx) {
  return f.xyb[c] + ((size_t)((int)aby - (f.plane_y0 >> 3)) * f.tile_stride + abx) * 64u;
}

// ---- in-register 1-D transforms (lib/jxl/dct-inl.h:158-232) --------------
static constexpr float kSqrt2 = 1.41421356237f;  // lib/jxl/dct_scales.h:15

// IDCT1DImpl<N>: even/odd split, recurse, B-transpose, butterfly with
// W_N[i] = 1/(2cos((i+1/2)pi/N)).  v is a fully-unrolled private array.
template <int N>
__device__ __forceinline__ void IdctReg(float* __restrict__ v) {
  if constexpr (N == 1) {
    return;
  } else if constexpr (N == 2) {
    const float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else {
    constexpr int H = N / 2;
    float e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; i++) {
      e[i] = v[2 * i];
      o[i] = v[2 * i + 1];
    }
    IdctReg<H>(e);
#pragma unroll
    for (int i = H - 1; i > 0; i--) o[i] = o[i] + o[i - 1];
    o[0] = o[0] * kSqrt2;
    IdctReg<H>(o);
#pragma unroll
    for (int i = 0; i < H; i++) {
      const float mul = kWcHost[N + i];
      v[i] = __builtin_fmaf(mul, o[i], e[i]);
      v[N - 1 - i] = __builtin_fmaf(-mul, o[i], e[i]);
    }
  }
}

// DCT1DImpl<N> (unscaled), used for LLF-from-DC.
template <int N>
__device__ __forceinline__ void DctReg(float* __restrict__ v) {
  if constexpr (N == 1) {
    return;
  } else if constexpr (N == 2) {
    const float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else {
    constexpr int H = N / 2;
    float s[H], d[H];
#pragma unroll
    for (int i = 0; i < H; i++) s[i] = v[i] + v[N - 1 - i];
    DctReg<H>(s);
#pragma unroll
    for (int i = 0; i < H; i++) d[i] = v[i] - v[N - 1 - i];
#pragma unroll
    for (int i = 0; i < H; i++) d[i] = d[i] * kWcHost[N + i];
    DctReg<H>(d);
    d[0] = __builtin_fmaf(d[0], kSqrt2, d[1]);
#pragma unroll
    for (int i = 1; i + 1 < H; i++) d[i] = d[i] + d[i + 1];
#pragma unroll
    for (int i = 0; i < H; i++) {
      v[2 * i] = s[i];
      v[2 * i + 1] = d[i];
    }
  }
}

// ComputeScaledIDCT<R,C> (dct-inl.h:376-397) entirely in registers.
// m: coefficient matrix min(R,C) x max(R,C) (transposed storage when R >= C);
// out: R x C pixels, row-major.
template <int R, int C>
__device__ __forceinline__ void Idct2dReg(const float* __restrict__ m,
                                          float* __restrict__ out) {
  float t[R * C];  // t[u][x]
#pragma unroll
  for (int u = 0; u < R; u++) {
    float v[C];
#pragma unroll
    for (int j = 0; j < C; j++) v[j] = (R < C) ? m[u * C + j] : m[j * R + u];
    IdctReg<C>(v);
#pragma unroll
    for (int j = 0; j < C; j++) t[u * C + j] = v[j];
  }
#pragma unroll
  for (int x = 0; x < C; x++) {
    float v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = t[j * C + x];
    IdctReg<R>(v);
#pragma unroll
    for (int j = 0; j < R; j++) out[j * C + x] = v[j];
  }
}

// AdjustQuantBias (lib/jxl/quantizer-inl.h:34-67).  The reciprocal is the
// hardware v_rcp_f32 (1 ulp), like the reference's ApproximateReciprocal on
// SIMD targets.
__device__ __forceinline__ float AdjustQuantBias(int32_t q, float bias_c,
                                                 float bias3) {
  const float quant = (float)q;
  const float aq = __builtin_fabsf(quant);
  // |q| <= 1: 0 or +-bias_c (quantizer-inl.h:47-58) = bias_c * q exactly for q in {-1, 0, 1}: one multiply
  // instead of compare + copysign + select
  const float small = bias_c * quant;
  const float big = __builtin_fmaf(-bias3, __builtin_amdgcn_rcpf(quant), quant);
  return aq < 1.125f ? small : big;
}

// ---- FastLog2f / FastPow2f / FastPowf (lib/jxl/base/fast_math-inl.h:46-90): (2,2)
// and (3,3) rational approximations, used by the dequant-table generator and by
// the 709 / gamma output transfer functions
__device__ __forceinline__ float FastLog2f(float x) {
  const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f,
              p2 = 7.4245873327820566E-01f;
  const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f,
              q2 = 1.7409343003366853E-01f;
  const int32_t x_bits = __float_as_int(x);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;
  const int32_t exp_shifted = exp_bits >> 23;
  const float mantissa = __int_as_float(x_bits - (int32_t)((uint32_t)exp_shifted << 23));
  const float exp_val = (float)exp_shifted;
  const float m = mantissa - 1.0f;
  const float yp = __builtin_fmaf(__builtin_fmaf(p2, m, p1), m, p0);
  const float yq = __builtin_fmaf(__builtin_fmaf(q2, m, q1), m, q0);
  return yp / yq + exp_val;
}

__device__ __forceinline__ float FastPow2f(float x) {
  const float floorx = __builtin_floorf(x);
  const float expf_ = __int_as_float(((int32_t)floorx + 127) << 23);
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = __builtin_fmaf(num, frac, 4.88687798e+01f);
  num = __builtin_fmaf(num, frac, 9.85506591e+01f);
  num = num * expf_;
  float den = __builtin_fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = __builtin_fmaf(den, frac, -1.94414990e+01f);
  den = __builtin_fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}

__device__ __forceinline__ float FastPowf(float base, float exponent) {
  return FastPow2f(FastLog2f(base) * exponent);
}

template <typename CT>
__device__ __forceinline__ int32_t LoadCoeff(const void* base, size_t i) {
  return (int32_t)((const CT*)base)[i];
}

}  // namespace jxlhip
#endif
