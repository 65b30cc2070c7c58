// emit.h -- the tail of the render pipeline fused into the filter kernels:
// XYB -> linear RGB (lib/jxl/dec_xyb-inl.h:38-86, stage_xyb.cc:42-98),
// FromLinearStage (stage_from_linear.cc:34-155) and WriteToOutputStage's sample
// conversion + interleaving (stage_write.cc:254-330,524-640).  Operation order
// follows the reference's scalar (single-lane) evaluation so that results are
// bit-identical to it: explicit fmaf, IEEE sqrt and division.
#ifndef JXLHIP_EMIT_H_
#define JXLHIP_EMIT_H_

#include "dev_common.h"
#include "kernels.h"

namespace jxlhip {

__device__ __forceinline__ void XybToRgb(float x, float y, float b, const FilterParams& P,
                                         float* rgb) {
  float gr = y + x, gg = y - x, gb = b;
  gr = gr - P.cbrt_bias[0];
  gg = gg - P.cbrt_bias[1];
  gb = gb - P.cbrt_bias[2];
  const float mr = __builtin_fmaf(gr * gr, gr, P.opsin_bias[0]);
  const float mg = __builtin_fmaf(gg * gg, gg, P.opsin_bias[1]);
  const float mb = __builtin_fmaf(gb * gb, gb, P.opsin_bias[2]);
  const float* m = P.minv;
  rgb[0] = __builtin_fmaf(m[2], mb, __builtin_fmaf(m[1], mg, m[0] * mr));
  rgb[1] = __builtin_fmaf(m[5], mb, __builtin_fmaf(m[4], mg, m[3] * mr));
  rgb[2] = __builtin_fmaf(m[8], mb, __builtin_fmaf(m[7], mg, m[6] * mr));
}

// TF_SRGB::EncodedFromDisplay (lib/jxl/cms/transfer_functions-inl.h:244-268):
// sign-symmetric; 12.92 x below 0.0031308, else a degree-4/4 rational
// polynomial in sqrt(x) evaluated by Horner's scheme with fused multiply-adds
// (base/rational_polynomial-inl.h:59-97) and one division.  The square root and
// the division are the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (the correctly
// rounded sequences cost ~25 VALU operations per sample and doubled the filter
// kernel's time); the result differs from the reference's by <= 2 ulp, far
// inside what the float pipeline in front of it already differs by.
__device__ __forceinline__ float SrgbFromLinear(float v) {
  const float x = __builtin_fabsf(v);
  const float s = __builtin_amdgcn_sqrtf(x);
  float yp = 7.352629620e-01f, yq = 2.424867759e-02f;
  yp = __builtin_fmaf(yp, s, 1.474205315e+00f);
  yq = __builtin_fmaf(yq, s, 9.258482155e-01f);
  yp = __builtin_fmaf(yp, s, 3.903842876e-01f);
  yq = __builtin_fmaf(yq, s, 1.340816930e+00f);
  yp = __builtin_fmaf(yp, s, 5.287254571e-03f);
  yq = __builtin_fmaf(yq, s, 3.036675394e-01f);
  yp = __builtin_fmaf(yp, s, -5.135152395e-04f);
  yq = __builtin_fmaf(yq, s, 1.004519624e-02f);
  const float poly = yp * __builtin_amdgcn_rcpf(yq);
  const float mag = x > 0.0031308f ? poly : x * 12.92f;
  return __builtin_copysignf(mag, v);
}

// EvalRationalPolynomial of degree 4/4 (base/rational_polynomial-inl.h:59-97) with
// the hardware reciprocal (see SrgbFromLinear)
__device__ __forceinline__ float Rational44(float x, const float* p, const float* q) {
  float yp = p[4], yq = q[4];
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    yp = __builtin_fmaf(yp, x, p[i]);
    yq = __builtin_fmaf(yq, x, q[i]);
  }
  return yp * __builtin_amdgcn_rcpf(yq);
}

// TF_PQ::EncodedFromDisplay (transfer_functions-inl.h:172-208): rational
// polynomials in x^(1/4), one below and one above 1e-4; `to_10000` =
// intensity_target / 10000
__device__ __forceinline__ float PqFromLinear(float v, float to_10000) {
  const float kP[5] = {1.351392e-02f, -1.095778e+00f, 5.522776e+01f, 1.492516e+02f, 4.838434e+01f};
  const float kQ[5] = {1.012416e+00f, 2.016708e+01f, 9.263710e+01f, 1.120607e+02f, 2.590418e+01f};
  const float kPlo[5] = {9.863406e-06f, 3.881234e-01f, 1.352821e+02f, 6.889862e+04f, -2.864824e+05f};
  const float kQlo[5] = {3.371868e+01f, 1.477719e+03f, 1.608477e+04f, -4.389884e+04f, -2.072546e+05f};
  const float x = __builtin_fabsf(v);
  const float r = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(x * to_10000));
  const float mag = x < 1e-4f ? Rational44(r, kPlo, kQlo) : Rational44(r, kP, kQ);
  return __builtin_copysignf(mag, v);
}

// TF_709::EncodedFromDisplay (transfer_functions-inl.h:104-111)
__device__ __forceinline__ float Bt709FromLinear(float x) {
  const float hi = __builtin_fmaf(1.099f, FastPowf(x, 0.45f), -0.099f);
  return x <= 0.018f ? 4.5f * x : hi;
}

// OpGamma (stage_from_linear.cc:96-109)
__device__ __forceinline__ float GammaFromLinear(float x, float inverse_gamma) {
  return x <= 1e-5f ? 0.0f : FastPowf(x, inverse_gamma);
}

// TF_HLG::EncodedFromDisplay (transfer_functions-inl.h:53-69): sqrt(3x) up to 1/12,
// kA ln(12x - kB) + kC above (through FastLog2f), odd symmetry
__device__ __forceinline__ float HlgFromLinear(float v) {
  constexpr double kA = 0.17883277, kB = 1 - 4 * kA, kC = 0.5599107295, kInvLog2e = 0.6931471805599453;
  const float x = __builtin_fabsf(v);
  const float lo = __builtin_sqrtf(3.0f * x);
  const float hi = __builtin_fmaf((float)(kA * kInvLog2e), FastLog2f(__builtin_fmaf(12.0f, x, (float)-kB)), (float)kC);
  const float mag = x <= (float)(1.0 / 12.0) ? lo : hi;
  return __builtin_copysignf(__builtin_fabsf(mag), v);
}

// HlgOOTF::Apply (cms/tone_mapping-inl.h:121-133): the whole pixel is scaled by
// luminance^exponent (exponent == 0: the reference skips the OOTF)
__device__ __forceinline__ void HlgOotf(const FilterParams& P, float* rgb) {
  if (P.hlg_exponent == 0.0f) return;
  const float lum = __builtin_fmaf(P.fmt.luminances[0], rgb[0],
                                   __builtin_fmaf(P.fmt.luminances[1], rgb[1], P.fmt.luminances[2] * rgb[2]));
  const float ratio = __builtin_fminf(FastPowf(lum, P.hlg_exponent), 1e9f);
  rgb[0] *= ratio;
  rgb[1] *= ratio;
  rgb[2] *= ratio;
}

__device__ __forceinline__ float ApplyTransfer(uint32_t tf, float v, float tf_scale) {
  switch (tf) {
    case JXLHIP_TF_HLG: return HlgFromLinear(v);
    case JXLHIP_TF_SRGB: return SrgbFromLinear(v);
    case JXLHIP_TF_PQ: return PqFromLinear(v, tf_scale);
    case JXLHIP_TF_709: return Bt709FromLinear(v);
    case JXLHIP_TF_GAMMA: return GammaFromLinear(v, tf_scale);
    default: return v;
  }
}

// Output format as seen by the emit code: FmtSel<-1> reads it from the launch
// parameters (wave-uniform branches per sample -- correct for every format, but
// the scalar branches cost as much as the arithmetic); FmtSel<ID> with
// ID = FormatId(transfer, sample_type, channels, swap_endianness) fixes it at compile time for
// the formats the launchers specialise (tf_param, the bit depth and the HLG luminances stay launch parameters).
__host__ __device__ constexpr int FormatId(int transfer, int sample_type, int channels, int swap = 0) {
  return transfer | (sample_type << 3) | ((channels - 3) << 5) | (swap << 6);
}
template <int ID>
struct FmtSel {
  static __device__ __forceinline__ uint32_t transfer(const jxlhip_output_format&) { return ID & 7; }  // JXLHIP_TF_*
  static __device__ __forceinline__ uint32_t sample_type(const jxlhip_output_format&) { return (ID >> 3) & 3; }
  static __device__ __forceinline__ uint32_t channels(const jxlhip_output_format&) { return 3 + ((ID >> 5) & 1); }
  static __device__ __forceinline__ bool swap(const jxlhip_output_format&) { return ((ID >> 6) & 1) != 0; }
};
template <>
struct FmtSel<-1> {
  static __device__ __forceinline__ uint32_t transfer(const jxlhip_output_format& F) { return F.transfer; }
  static __device__ __forceinline__ uint32_t sample_type(const jxlhip_output_format& F) { return F.sample_type; }
  static __device__ __forceinline__ uint32_t channels(const jxlhip_output_format& F) { return F.num_channels; }
  static __device__ __forceinline__ bool swap(const jxlhip_output_format& F) { return F.swap_endianness != 0; }
};

// MakeUnsigned (stage_write.cc:263-284): scale, ordered dither for 8-bit types,
// clamp, round to nearest even.  (x, y) are image coordinates, c the channel.
__device__ __forceinline__ uint16_t Bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

template <typename DitherPtr>
__device__ __forceinline__ uint32_t ToUnsigned(const FilterParams& P, DitherPtr dither, float v, int x,
                                               int y, int c, bool dithered) {
  v = v * P.sample_mul;
  if (dithered) {
    const uint32_t dx = (uint32_t)(P.dither_x0 + P.dither_xs * x), dy = (uint32_t)(P.dither_y0 + P.dither_ys * y);
    v = v + dither[((dy + 13u * c) & 31u) * 32u + ((dx + 23u * c) & 31u)];
  }
  v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), P.sample_mul);
  return (uint32_t)(int32_t)__builtin_rintf(v);
}

__device__ __forceinline__ uint16_t HalfBits(float v) {
  const _Float16 h = (_Float16)v;  // round to nearest even (DemoteTo)
  uint16_t q;
  __builtin_memcpy(&q, &h, 2);
  return q;
}

// The samples of one pixel as the integers that go to memory (before the
// endianness swap): u8 / u16 values, f16 bits or f32 bits.  rgb are LINEAR.
template <typename Sel, typename DitherPtr>
__device__ __forceinline__ void PackSamples(const FilterParams& P, DitherPtr dither, int x, int y,
                                            const float* rgb, uint32_t* q) {
  const jxlhip_output_format& F = P.fmt;
  const uint32_t st = Sel::sample_type(F);
  const uint32_t tf = Sel::transfer(F);
  float v[4] = {rgb[0], rgb[1], rgb[2], 1.0f};
  // the alpha channel goes to the output like a colour channel, minus the transfer function (stage_write.cc:350-366)
  if (Sel::channels(F) == 4 && P.alpha) v[3] = P.alpha[(size_t)y * P.alpha_stride + x];
  if (tf == JXLHIP_TF_HLG) HlgOotf(P, v);
#pragma unroll
  for (int c = 0; c < 3; c++) v[c] = ApplyTransfer(tf, v[c], P.tf_scale);
  if (st == JXLHIP_SAMPLE_U8) {
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = ToUnsigned(P, dither, v[c], x, y, c, true);
    return;
  }
  if (st == JXLHIP_SAMPLE_U16) {
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = ToUnsigned(P, dither, v[c], x, y, c, false);
  } else if (st == JXLHIP_SAMPLE_F16) {
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = HalfBits(v[c]);
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = __float_as_uint(v[c]);
  }
  if (Sel::swap(F)) {  // (general format: one uniform branch)
#pragma unroll
    for (int c = 0; c < 4; c++)
      q[c] = st == JXLHIP_SAMPLE_F32 ? __builtin_bswap32(q[c]) : (uint32_t)Bswap16((uint16_t)q[c]);
  }
}

// One pixel of the JXLHIP_OUT_PACKED output: rgb are LINEAR samples; alpha (4
// channels) is FilterParams::alpha at (x, y), or the opaque 1.0 the reference
// substitutes when the frame has no alpha channel (stage_write.cc:355-360).
// `row` = first byte of output row y.
template <typename Sel, typename DitherPtr>
__device__ __forceinline__ void StorePackedPixel(const FilterParams& P, DitherPtr dither, char* row,
                                                 int x, int y, const float* rgb) {
  const jxlhip_output_format& F = P.fmt;
  const int nc = (int)Sel::channels(F);
  const uint32_t st = Sel::sample_type(F);
  uint32_t q[4];
  PackSamples<Sel>(P, dither, x, y, rgb, q);
  if (st == JXLHIP_SAMPLE_U8) {
    uint8_t* d = (uint8_t*)row + (size_t)x * nc;
    if (nc == 4) {
      *(uint32_t*)d = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    } else {
      d[0] = (uint8_t)q[0];
      d[1] = (uint8_t)q[1];
      d[2] = (uint8_t)q[2];
    }
  } else if (st == JXLHIP_SAMPLE_F32) {
    uint32_t* d = (uint32_t*)row + (size_t)x * nc;
    for (int c = 0; c < nc; c++) d[c] = q[c];
  } else {
    uint16_t* d = (uint16_t*)row + (size_t)x * nc;
    for (int c = 0; c < nc; c++) d[c] = (uint16_t)q[c];
  }
}

// Two horizontally adjacent pixels (x even) in as few, as wide stores as the
// format allows: their samples are contiguous in memory.
template <typename Sel, typename DitherPtr>
__device__ __forceinline__ void StorePackedPair(const FilterParams& P, DitherPtr dither, char* row,
                                                int x, int y, const float* rgb0, const float* rgb1) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(4)));
  typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
  const jxlhip_output_format& F = P.fmt;
  const bool rgba = Sel::channels(F) == 4;
  const uint32_t st = Sel::sample_type(F);
  uint32_t a[4], b[4];
  PackSamples<Sel>(P, dither, x, y, rgb0, a);
  PackSamples<Sel>(P, dither, x + 1, y, rgb1, b);
  if (st == JXLHIP_SAMPLE_U8) {
    if (rgba) {
      __builtin_nontemporal_store(u2{a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24),
                                     b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)},
                                  (u2*)(row + (size_t)x * 4));
    } else {  // 6 bytes at a 2-byte aligned address
      uint16_t* d = (uint16_t*)(row + (size_t)x * 3);
      d[0] = (uint16_t)(a[0] | (a[1] << 8));
      d[1] = (uint16_t)(a[2] | (b[0] << 8));
      d[2] = (uint16_t)(b[1] | (b[2] << 8));
    }
  } else if (st == JXLHIP_SAMPLE_F32) {
    uint32_t* d = (uint32_t*)row + (size_t)x * Sel::channels(F);
    if (rgba) {
      __builtin_nontemporal_store(u4{a[0], a[1], a[2], a[3]}, (u4*)d);
      __builtin_nontemporal_store(u4{b[0], b[1], b[2], b[3]}, (u4*)(d + 4));
    } else {
      // (the fence keeps the pair of stores a pair: merged and re-split into two 12-byte stores they are 6x slower)
      __builtin_nontemporal_store(u4{a[0], a[1], a[2], b[0]}, (u4*)d);
      asm volatile("" ::: "memory");
      __builtin_nontemporal_store(u2{b[1], b[2]}, (u2*)(d + 4));
    }
  } else {  // 16-bit samples
    uint32_t* d = (uint32_t*)(row + (size_t)x * Sel::channels(F) * 2);
    if (rgba) {
      __builtin_nontemporal_store(u4{a[0] | (a[1] << 16), a[2] | (a[3] << 16), b[0] | (b[1] << 16),
                                     b[2] | (b[3] << 16)},
                                  (u4*)d);
    } else {  // 12 bytes as 8 + 4 (a 12-byte vector store measured 8x slower here)
      __builtin_nontemporal_store(u2{a[0] | (a[1] << 16), a[2] | (b[0] << 16)}, (u2*)d);
      asm volatile("" ::: "memory");  // (or the compiler merges the two into that 12-byte store)
      __builtin_nontemporal_store(b[1] | (b[2] << 16), d + 2);
    }
  }
}

}  // namespace jxlhip
#endif
