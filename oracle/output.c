/* output.c -- TEST INFRASTRUCTURE (oracle): restatement of the colour-encoding
 * and packing tail of the render pipeline for JXLHIP_OUT_PACKED:
 *   FromLinearStage   lib/jxl/render_pipeline/stage_from_linear.cc:34-155
 *     TF_SRGB::EncodedFromDisplay  lib/jxl/cms/transfer_functions-inl.h:244-268
 *     EvalRationalPolynomial       lib/jxl/base/rational_polynomial-inl.h:59-97
 *   WriteToOutputStage lib/jxl/render_pipeline/stage_write.cc:254-330,524-640
 * held bit-exact to the reference library by tests/test_reference_parity.py. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "jxl_oracle.h"
#include "dither_pattern.inc"

float jxo_srgb_from_linear(float v) {
  static const float p[5] = {-5.135152395e-04f, 5.287254571e-03f, 3.903842876e-01f, 1.474205315e+00f,
                             7.352629620e-01f};
  static const float q[5] = {1.004519624e-02f, 3.036675394e-01f, 1.340816930e+00f, 9.258482155e-01f,
                             2.424867759e-02f};
  const float x = fabsf(v);
  const float s = sqrtf(x);
  float yp = p[4], yq = q[4];
  for (int i = 3; i >= 0; i--) {
    yp = fmaf(yp, s, p[i]);
    yq = fmaf(yq, s, q[i]);
  }
  const float poly = yp / yq;
  const float mag = x > 0.0031308f ? poly : x * 12.92f;
  return copysignf(mag, v);
}

static float rational44(float x, const float* p, const float* q) {
  float yp = p[4], yq = q[4];
  for (int i = 3; i >= 0; i--) {
    yp = fmaf(yp, x, p[i]);
    yq = fmaf(yq, x, q[i]);
  }
  return yp / yq;
}

/* TF_PQ::EncodedFromDisplay (transfer_functions-inl.h:172-208) */
float jxo_pq_from_linear(float v, float intensity_target) {
  static const float p[5] = {1.351392e-02f, -1.095778e+00f, 5.522776e+01f, 1.492516e+02f, 4.838434e+01f};
  static const float q[5] = {1.012416e+00f, 2.016708e+01f, 9.263710e+01f, 1.120607e+02f, 2.590418e+01f};
  static const float plo[5] = {9.863406e-06f, 3.881234e-01f, 1.352821e+02f, 6.889862e+04f, -2.864824e+05f};
  static const float qlo[5] = {3.371868e+01f, 1.477719e+03f, 1.608477e+04f, -4.389884e+04f, -2.072546e+05f};
  const float to_10000 = intensity_target * (1.0f / 10000.0f);
  const float x = fabsf(v);
  const float r = sqrtf(sqrtf(x * to_10000));
  const float mag = x < 1e-4f ? rational44(r, plo, qlo) : rational44(r, p, q);
  return copysignf(mag, v);
}

/* TF_709::EncodedFromDisplay (transfer_functions-inl.h:104-111) */
float jxo_709_from_linear(float x) {
  const float hi = fmaf(1.099f, jxo_fast_powf(x, 0.45f), -0.099f);
  return x <= 0.018f ? 4.5f * x : hi;
}

/* OpGamma (stage_from_linear.cc:96-109) */
float jxo_gamma_from_linear(float x, float inverse_gamma) {
  return x <= 1e-5f ? 0.0f : jxo_fast_powf(x, inverse_gamma);
}

/* FastLog2f (fast_math-inl.h:46-68); same statement as in quant_tables.c */
static float fast_log2f(float x) {
  const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f, p2 = 7.4245873327820566E-01f;
  const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f, q2 = 1.7409343003366853E-01f;
  int32_t x_bits;
  memcpy(&x_bits, &x, 4);
  const int32_t exp_bits = (int32_t)((uint32_t)x_bits - 0x3f2aaaabu);
  const int32_t exp_shifted = exp_bits >> 23;
  const int32_t mant_bits = (int32_t)((uint32_t)x_bits - ((uint32_t)exp_shifted << 23));
  float mantissa;
  memcpy(&mantissa, &mant_bits, 4);
  const float m = mantissa - 1.0f;
  const float yp = fmaf(fmaf(p2, m, p1), m, p0);
  const float yq = fmaf(fmaf(q2, m, q1), m, q0);
  return yp / yq + (float)exp_shifted;
}

/* TF_HLG::EncodedFromDisplay (transfer_functions-inl.h:53-69) */
float jxo_hlg_from_linear(float v) {
  const double kA = 0.17883277, kB = 1 - 4 * kA, kC = 0.5599107295, kInvLog2e = 0.6931471805599453;
  const float x = fabsf(v);
  const float lo = sqrtf(3.0f * x);
  const float hi = fmaf((float)(kA * kInvLog2e), fast_log2f(fmaf(12.0f, x, (float)-kB)), (float)kC);
  const float mag = x <= (float)(1.0 / 12.0) ? lo : hi;
  return copysignf(fabsf(mag), v);
}

/* HlgOOTF::ToSceneLight + Apply (cms/tone_mapping-inl.h:113-133, tone_mapping.h:120-126) */
static void hlg_ootf(const jxlhip_output_format* F, float* v) {
  const float gamma = (1 / 1.2f) * powf(1.111f, -log2f(F->tf_param / 1000.f));
  const float e = gamma - 1;
  if (!(e < -0.01f || 0.01f < e)) return;
  const float lum = fmaf(F->luminances[0], v[0], fmaf(F->luminances[1], v[1], F->luminances[2] * v[2]));
  const float pw = jxo_fast_powf(lum, e);
  const float ratio = pw < 1e9f ? pw : 1e9f; /* Min(a, b) = a < b ? a : b */
  v[0] *= ratio;
  v[1] *= ratio;
  v[2] *= ratio;
}

static float apply_tf(const jxlhip_output_format* F, float v) {
  switch (F->transfer) {
    case JXLHIP_TF_HLG: return jxo_hlg_from_linear(v);
    case JXLHIP_TF_SRGB: return jxo_srgb_from_linear(v);
    case JXLHIP_TF_PQ: return jxo_pq_from_linear(v, F->tf_param);
    case JXLHIP_TF_709: return jxo_709_from_linear(v);
    case JXLHIP_TF_GAMMA: return jxo_gamma_from_linear(v, F->tf_param);
    default: return v;
  }
}

/* IEEE binary16 bits of v, round to nearest even (hwy DemoteTo(float16)) */
static uint16_t f16_bits(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127;
  uint32_t man = u & 0x7fffffu;
  if (exp == 128) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp > 15) return (uint16_t)(sign | 0x7c00u);
  if (exp >= -14) {
    uint32_t h = ((uint32_t)(exp + 15) << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
  }
  if (exp < -25) return (uint16_t)sign;
  man |= 0x800000u;
  const int shift = -exp - 14 + 13;  /* 14..24 */
  uint32_t h = man >> shift;
  const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1))) h++;
  return (uint16_t)(sign | h);
}

static uint32_t to_unsigned(float v, float mul, uint32_t x, uint32_t y, int c, int dithered) {
  v = v * mul;
  if (dithered) v = v + kDitherPattern[((y + 13u * c) & 31u) * 32u + ((x + 23u * c) & 31u)];
  v = fminf(fmaxf(v, 0.0f), mul);
  return (uint32_t)(int32_t)rintf(v);
}

void jxo_pack_output(const jxo_frame* f, const float* rgb, size_t rgb_stride, void* out,
                     size_t out_stride_bytes, uint32_t row_begin, uint32_t row_end) {
  const jxlhip_frame_params* p = &f->p;
  const jxlhip_output_format* F = &p->out_format;
  const int nc = (int)F->num_channels;
  const int is_int = F->sample_type == JXLHIP_SAMPLE_U8 || F->sample_type == JXLHIP_SAMPLE_U16;
  const float mul = is_int ? (float)((1u << F->bits_per_sample) - 1u) : 1.0f;
  for (uint32_t y = row_begin; y < row_end; y++) {
    uint8_t* row = (uint8_t*)out + (size_t)y * out_stride_bytes;
    for (uint32_t x = 0; x < p->xsize; x++) {
      float v[4];
      for (int c = 0; c < 3; c++) v[c] = rgb[(size_t)y * rgb_stride + 3 * (size_t)x + c];
      if (F->transfer == JXLHIP_TF_HLG) hlg_ootf(F, v);
      for (int c = 0; c < 3; c++) v[c] = apply_tf(F, v[c]);
      v[3] = 1.0f;
      for (int c = 0; c < nc; c++) {
        const size_t i = (size_t)x * nc + c;
        if (F->sample_type == JXLHIP_SAMPLE_U8) {
          row[i] = (uint8_t)to_unsigned(v[c], mul, x, y, c, 1);
        } else if (F->sample_type == JXLHIP_SAMPLE_F32) {
          uint32_t u;
          memcpy(&u, &v[c], 4);
          if (F->swap_endianness) u = __builtin_bswap32(u);
          memcpy(row + 4 * i, &u, 4);
        } else {
          uint16_t q = F->sample_type == JXLHIP_SAMPLE_U16 ? (uint16_t)to_unsigned(v[c], mul, x, y, c, 0)
                                                           : f16_bits(v[c]);
          if (F->swap_endianness) q = (uint16_t)((q >> 8) | (q << 8));
          memcpy(row + 2 * i, &q, 2);
        }
      }
    }
  }
}
