/* oracle/quant_tables.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates lib/jxl/quant_weights.cc:48-160 (GetQuantWeights*, Interpolate,
 * Mult), :163-358 (ComputeQuantTable), :1190-1271 (layout), and
 * lib/jxl/base/fast_math-inl.h:46-92 + rational_polynomial-inl.h:60-97
 * (FastLog2f/FastPow2f/FastPowf).  Default parameter library: data in
 * format_constants.inc. */
#include <math.h>
#include <string.h>

#include "jxl_oracle.h"
#include "format_constants.inc"

static const uint8_t kKindShort[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2,
                                       1, 1, 8, 4, 16, 8, 32, 16};
static const uint8_t kKindLong[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4,
                                      1, 1, 8, 8, 16, 16, 32, 32};
static const float kAlmostZero = 1e-8f;

/* FastLog2f (fast_math-inl.h:46-68): (2,2) rational approx of log2 */
static float fast_log2f(float x) {
  const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f,
              p2 = 7.4245873327820566E-01f;
  const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f,
              q2 = 1.7409343003366853E-01f;
  int32_t x_bits;
  memcpy(&x_bits, &x, 4);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;
  const int32_t exp_shifted = exp_bits >> 23;
  const int32_t mant_bits = x_bits - (int32_t)((uint32_t)exp_shifted << 23);
  float mantissa;
  memcpy(&mantissa, &mant_bits, 4);
  const float exp_val = (float)exp_shifted;
  const float m = mantissa - 1.0f;
  /* EvalRationalPolynomial: Horner with fma, then divide */
  float yp = fmaf(fmaf(p2, m, p1), m, p0);
  float yq = fmaf(fmaf(q2, m, q1), m, q0);
  return yp / yq + exp_val;
}

/* FastPow2f (fast_math-inl.h:70-86) */
static float fast_pow2f(float x) {
  const float floorx = floorf(x);
  const int32_t e = ((int32_t)floorx + 127) << 23;
  float expf_;
  memcpy(&expf_, &e, 4);
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = fmaf(num, frac, 4.88687798e+01f);
  num = fmaf(num, frac, 9.85506591e+01f);
  num = num * expf_;
  float den = fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = fmaf(den, frac, -1.94414990e+01f);
  den = fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}

float jxo_fast_powf(float base, float exponent) {
  return fast_pow2f(fast_log2f(base) * exponent);
}

static float mult(float v) { return v > 0.0f ? 1.0f + v : 1.0f / (1.0f - v); }

/* GetQuantWeights (quant_weights.cc:129-160): rows x cols, 3 channels */
static int dct_weights(int rows, int cols,
                       const float bands_in[3][JXLHIP_MAX_DISTANCE_BANDS], int nb,
                       float* out) {
  const float kSqrt2 = 1.41421356237f;
  for (int c = 0; c < 3; c++) {
    float bands[17];
    bands[0] = bands_in[c][0];
    if (bands[0] < kAlmostZero) return -1;
    for (int i = 1; i < nb; i++) {
      bands[i] = bands[i - 1] * mult(bands_in[c][i]);
      if (bands[i] < kAlmostZero) return -1;
    }
    const float scale = (nb - 1) / (kSqrt2 + 1e-6f);
    const float rcpcol = scale / (cols - 1);
    const float rcprow = scale / (rows - 1);
    for (int y = 0; y < rows; y++) {
      const float dy = y * rcprow;
      const float dy2 = dy * dy;
      for (int x = 0; x < cols; x++) {
        const float dx = (float)x * rcpcol;
        const float dist = sqrtf(fmaf(dx, dx, dy2));
        float w;
        if (nb == 1) {
          w = bands[0];
        } else { /* InterpolateVec */
          const int32_t idx = (int32_t)dist;
          const float frac = dist - (float)idx;
          const float a = bands[idx], b = bands[idx + 1];
          w = a * jxo_fast_powf(b / a, frac);
        }
        out[c * cols * rows + y * cols + x] = w;
      }
    }
  }
  return 0;
}

/* Interpolate (quant_weights.cc:92-100) */
static float interpolate(float pos, float max, const float* array, int len) {
  const float scaled_pos = pos * (len - 1) / max;
  const int idx = (int)scaled_pos;
  const float a = array[idx], b = array[idx + 1];
  return a * jxo_fast_powf(b / a, scaled_pos - idx);
}

/* Library encodings -> the parameters they stand for (DequantMatrices::Library,
 * quant_weights.cc:532-1188); the AFV entry's 4x8 / 4x4 band parameters are the
 * DCT4X8 / DCT4X4 entries'. */
static void resolve_library(int kind, jxlhip_quant_encoding* e) {
  static const uint32_t kModeOfLib[6] = {JXLHIP_QUANT_DCT,  JXLHIP_QUANT_ID,
                                         JXLHIP_QUANT_DCT2, JXLHIP_QUANT_DCT4,
                                         JXLHIP_QUANT_DCT4X8, JXLHIP_QUANT_AFV};
  const QuantLibEntry* l = &kQuantLib[kind];
  const QuantLibEntry* b = l->mode == 5 ? &kQuantLib[9] : l;
  memset(e, 0, sizeof(*e));
  e->mode = kModeOfLib[l->mode];
  e->num_bands = (uint32_t)b->nb;
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < 8; i++) e->bands[c][i] = b->bands[c][i];
    for (int i = 0; i < 9; i++) e->weights[c][i] = l->w[c][i];
  }
  if (l->mode == 5) {
    e->num_bands_afv_4x4 = (uint32_t)kQuantLib[3].nb;
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < 8; i++) e->bands_afv_4x4[c][i] = kQuantLib[3].bands[c][i];
  }
}

/* ComputeQuantTable (quant_weights.cc:163-358) for table `kind` under a
 * resolved (non-library) encoding */
static int compute_kind(int kind, const jxlhip_quant_encoding* enc, float* table,
                        float* inv_table) {
  /* the body below predates jxlhip_quant_encoding: view the encoding through its
   * old field names */
  struct { int mode; int nb; const float (*bands)[JXLHIP_MAX_DISTANCE_BANDS]; const float (*w)[9]; } ev, *e = &ev;
  static const int kOldMode[8] = {-1, 1, 2, 3, 4, 5, 0, -1};
  ev.mode = enc->mode < 8 ? kOldMode[enc->mode] : -1;
  ev.nb = (int)enc->num_bands;
  ev.bands = enc->bands;
  ev.w = enc->weights;
  if (ev.mode < 0) return -1;
  const int wrows = 8 * kKindShort[kind], wcols = 8 * kKindLong[kind];
  const int num = wrows * wcols;
  if (ev.mode != 0 && num != 64) return -1;
  float* w = inv_table; /* build weights in place in inv_table */
  switch (e->mode) {
    case 1: /* ID, :70-80 */
      for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 64; i++) w[64 * c + i] = e->w[c][0];
        w[64 * c + 1] = e->w[c][1];
        w[64 * c + 8] = e->w[c][1];
        w[64 * c + 9] = e->w[c][2];
      }
      break;
    case 2: /* DCT2, :48-78 */
      for (int c = 0; c < 3; c++) {
        float* s = w + c * 64;
        s[0] = 0xBAD;
        s[1] = s[8] = e->w[c][0];
        s[9] = e->w[c][1];
        for (int y = 0; y < 2; y++)
          for (int x = 0; x < 2; x++) {
            s[y * 8 + x + 2] = e->w[c][2];
            s[(y + 2) * 8 + x] = e->w[c][2];
            s[(y + 2) * 8 + x + 2] = e->w[c][3];
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            s[y * 8 + x + 4] = e->w[c][4];
            s[(y + 4) * 8 + x] = e->w[c][4];
            s[(y + 4) * 8 + x + 4] = e->w[c][5];
          }
      }
      break;
    case 3: { /* DCT4, :190-210 */
      float w44[3 * 16];
      if (dct_weights(4, 4, e->bands, e->nb, w44)) return -1;
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++)
            w[c * num + y * 8 + x] = w44[c * 16 + (y / 2) * 4 + (x / 2)];
        w[c * num + 1] /= e->w[c][0];
        w[c * num + 8] /= e->w[c][0];
        w[c * num + 9] /= e->w[c][1];
      }
      break;
    }
    case 4: { /* DCT4X8, :211-229 */
      float w48[3 * 32];
      if (dct_weights(4, 8, e->bands, e->nb, w48)) return -1;
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++)
            w[c * num + y * 8 + x] = w48[c * 32 + (y / 2) * 8 + x];
        w[c * num + 8] /= e->w[c][0];
      }
      break;
    }
    case 0: /* DCT */
      if (dct_weights(wrows, wcols, e->bands, e->nb, w)) return -1;
      break;
    case 5: { /* AFV, :246-326 */
      static const float kFreqs[16] = {
          0xBAD, 0xBAD, 0.8517778890324296f, 5.37778436506804f,
          0xBAD, 0xBAD, 4.734747904497923f, 5.449245381693219f,
          1.6598270267479331f, 4.0f, 7.275749096817861f, 10.423227632456525f,
          2.662932286148962f, 7.630657783650829f, 8.962388608184032f,
          12.97166202570235f};
      float w48[3 * 32], w44[3 * 16];
      if (dct_weights(4, 8, enc->bands, (int)enc->num_bands, w48)) return -1;
      if (dct_weights(4, 4, enc->bands_afv_4x4, (int)enc->num_bands_afv_4x4, w44)) return -1;
      const float lo = 0.8517778890324296f;
      const float hi = 12.97166202570235f - lo + 1e-6f;
      for (int c = 0; c < 3; c++) {
        float bands[4];
        bands[0] = e->w[c][5];
        if (bands[0] < kAlmostZero) return -1;
        for (int i = 1; i < 4; i++) {
          bands[i] = bands[i - 1] * mult(e->w[c][i + 5]);
          if (bands[i] < kAlmostZero) return -1;
        }
        float* s = w + c * 64;
        s[0] = 1;
        s[1 * 8 + 0] = e->w[c][0]; /* set_weight(x=0,y=1) */
        s[0 * 8 + 1] = e->w[c][1]; /* (1,0) */
        s[2 * 8 + 0] = e->w[c][2]; /* (0,2) */
        s[0 * 8 + 2] = e->w[c][3]; /* (2,0) */
        s[2 * 8 + 2] = e->w[c][4]; /* (2,2) */
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x < 2 && y < 2) continue;
            s[(2 * y) * 8 + 2 * x] =
                interpolate(kFreqs[y * 4 + x] - lo, hi, bands, 4);
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 8; x++) {
            if (x == 0 && y == 0) continue;
            s[(2 * y + 1) * 8 + x] = w48[c * 32 + y * 8 + x];
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x == 0 && y == 0) continue;
            s[(2 * y) * 8 + 2 * x + 1] = w44[c * 16 + y * 4 + x];
          }
      }
      break;
    }
  }
  for (int i = 0; i < 3 * num; i++) {
    const float inv_val = w[i];
    if (inv_val >= 1.0f / kAlmostZero || inv_val < kAlmostZero) return -1;
    table[i] = 1.0f / inv_val;
  }
  return 0;
}

int jxo_dequant_tables(const jxlhip_quant_encoding* encodings, float* table,
                       float* inv_table) {
  static __thread float scratch_inv[JXLHIP_DEQUANT_TABLE_FLOATS];
  float* inv = inv_table ? inv_table : scratch_inv;
  size_t pos = 0;
  for (int k = 0; k < 17; k++) {
    jxlhip_quant_encoding e;
    if (!encodings || encodings[k].mode == JXLHIP_QUANT_LIBRARY) resolve_library(k, &e);
    else e = encodings[k];
    if (compute_kind(k, &e, table + pos, inv + pos)) return -1;
    /* lowest frequencies get a 0 inverse table (quant_weights.cc:343-356) */
    const int xs = kKindShort[k], ys = kKindLong[k]; /* CoefficientLayout: ys>=xs */
    for (int c = 0; c < 3; c++)
      for (int y = 0; y < xs; y++)
        for (int x = 0; x < ys; x++)
          inv[pos + (size_t)c * xs * ys * 64 + (size_t)y * 8 * ys + x] = 0;
    pos += 3u * 64u * kKindShort[k] * kKindLong[k];
  }
  return pos == JXLHIP_DEQUANT_TABLE_FLOATS ? 0 : -1;
}

int jxo_default_dequant_tables(float* table, float* inv_table) {
  return jxo_dequant_tables(NULL, table, inv_table);
}
