/* oracle/transforms.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates lib/jxl/dec_transforms-inl.h:35-93 (ReinterpretingDCT,
 * IDCT2TopBlock), :95-454 (AFV), :456-689 (TransformToPixels), :691-818
 * (LowestFrequenciesFromDC), lib/jxl/dct_scales.h:34-232 (resample scales,
 * closed form from the comment), and for the test-only forward direction
 * lib/jxl/enc_transforms-inl.h:33-64,66-99,99-455,457-670,672-.  */
#include <math.h>
#include <string.h>

#include "jxl_oracle.h"
#include "format_constants.inc"

const float* jxo_afv_basis(void) { return kAfvBasis; }

/* DCTTotalResampleScale<N, 8N>(i) (dct_scales.h:140-232,371-375):
 * 1 / (cos(i pi/(2*8N)) cos(i pi/(8N)) cos(i pi/(4N))).  n = 1..32, i < n. */
static float resample_up(int n, int i) {
  const double big = 8.0 * n;
  double v = cos(i / (2 * big) * M_PI) * cos(i / big * M_PI) *
             cos(i / (big / 2) * M_PI);
  return (float)(1.0 / v);
}
/* DCTTotalResampleScale<8N, N>(i): the reciprocal direction */
static float resample_down(int n, int i) {
  const double big = 8.0 * n;
  double v = cos(i / (2 * big) * M_PI) * cos(i / big * M_PI) *
             cos(i / (big / 2) * M_PI);
  return (float)v;
}

/* ------------------------------------------------------------- LLF <- DC */
/* LowestFrequenciesFromDC (dec_transforms-inl.h:691-818) via
 * ReinterpretingDCT (:35-64). llf row stride = 8*max(cx,cy). */
void jxo_llf_from_dc(int strategy, const float* dc, size_t dc_stride,
                     float* llf) {
  const int cx = jxo_covered_blocks_x(strategy);
  const int cy = jxo_covered_blocks_y(strategy);
  if (cx == 1 && cy == 1) {
    llf[0] = dc[0];
    return;
  }
  float block[32 * 32];
  const int rows = cy, cols = cx;
  const size_t out_stride = 8 * (size_t)(cx > cy ? cx : cy);
  jxo_scaled_dct(rows, cols, dc, dc_stride, block);
  if (rows < cols) {
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++)
        llf[y * out_stride + x] =
            block[y * cols + x] * resample_up(rows, y) * resample_up(cols, x);
  } else {
    for (int y = 0; y < cols; y++)
      for (int x = 0; x < rows; x++)
        llf[y * out_stride + x] =
            block[y * rows + x] * resample_up(cols, y) * resample_up(rows, x);
  }
}

/* DCFromLowestFrequencies via ReinterpretingIDCT (enc_transforms-inl.h:33-64,
 * 672-) -- test-only inverse of the above. */
void jxo_dc_from_llf(int strategy, const float* block_in, float* dc,
                     size_t dc_stride) {
  const int cx = jxo_covered_blocks_x(strategy);
  const int cy = jxo_covered_blocks_y(strategy);
  if (cx == 1 && cy == 1) {
    dc[0] = block_in[0];
    return;
  }
  float block[32 * 32];
  const int rows = cy, cols = cx;
  const size_t in_stride = 8 * (size_t)(cx > cy ? cx : cy);
  if (rows < cols) {
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++)
        block[y * cols + x] = block_in[y * in_stride + x] *
                              resample_down(rows, y) * resample_down(cols, x);
  } else {
    for (int y = 0; y < cols; y++)
      for (int x = 0; x < rows; x++)
        block[y * rows + x] = block_in[y * in_stride + x] *
                              resample_down(cols, y) * resample_down(rows, x);
  }
  jxo_scaled_idct(rows, cols, block, dc, dc_stride);
}

/* ------------------------------------------------------------ small kinds */
/* IDCT2TopBlock<S> (dec_transforms-inl.h:66-93), in place on an 8x8 */
static void idct2_top(int s, float* b) {
  float t[64];
  const int h = s / 2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < h; x++) {
      float c00 = b[y * 8 + x], c01 = b[y * 8 + h + x];
      float c10 = b[(y + h) * 8 + x], c11 = b[(y + h) * 8 + h + x];
      t[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
      t[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
      t[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
      t[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
    }
  for (int y = 0; y < s; y++)
    for (int x = 0; x < s; x++) b[y * 8 + x] = t[y * 8 + x];
}
/* DCT2TopBlock<S> (enc_transforms-inl.h:66-99) */
static void dct2_top(int s, const float* in, size_t stride, float* out) {
  float t[64];
  const int h = s / 2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < h; x++) {
      float c00 = in[y * 2 * stride + x * 2], c01 = in[y * 2 * stride + x * 2 + 1];
      float c10 = in[(y * 2 + 1) * stride + x * 2];
      float c11 = in[(y * 2 + 1) * stride + x * 2 + 1];
      t[y * 8 + x] = (c00 + c01 + c10 + c11) * 0.25f;
      t[y * 8 + h + x] = (c00 + c01 - c10 - c11) * 0.25f;
      t[(y + h) * 8 + x] = (c00 - c01 + c10 - c11) * 0.25f;
      t[(y + h) * 8 + h + x] = (c00 - c01 - c10 + c11) * 0.25f;
    }
  for (int y = 0; y < s; y++)
    for (int x = 0; x < s; x++) out[y * 8 + x] = t[y * 8 + x];
}

/* AFVIDCT4x4 (dec_transforms-inl.h:95-397): pixel[i] = sum_j coeff[j]*B[j][i],
 * accumulated with fma in increasing j. */
static void afv_idct4x4(const float* coeffs, float* pixels) {
  for (int i = 0; i < 16; i++) {
    float p = 0.0f;
    for (int j = 0; j < 16; j++) p = fmaf(coeffs[j], kAfvBasis[j * 16 + i], p);
    pixels[i] = p;
  }
}
/* AFVDCT4x4 (enc_transforms-inl.h:99-406): coeff[j] = sum_i pixel[i]*B[j][i] */
static void afv_dct4x4(const float* pixels, float* coeffs) {
  for (int j = 0; j < 16; j++) {
    float c = 0.0f;
    for (int i = 0; i < 16; i++) c = fmaf(pixels[i], kAfvBasis[j * 16 + i], c);
    coeffs[j] = c;
  }
}

/* AFVTransformToPixels<kind> (dec_transforms-inl.h:399-454) */
static void afv_to_pixels(int kind, const float* co, float* px, size_t st) {
  const int afv_x = kind & 1, afv_y = kind / 2;
  const float b00 = co[0], b01 = co[1], b10 = co[8];
  const float dc0 = (b00 + b10 + b01) * 4.0f;
  const float dc1 = (b00 + b10 - b01);
  const float dc2 = b00 - b10;
  float coeff[16], block[32];
  coeff[0] = dc0;
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      if (ix | iy) coeff[iy * 4 + ix] = co[iy * 2 * 8 + ix * 2];
  afv_idct4x4(coeff, block);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      px[(iy + afv_y * 4) * st + afv_x * 4 + ix] =
          block[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)];
  block[0] = dc1;
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      if (ix | iy) block[iy * 4 + ix] = co[iy * 2 * 8 + ix * 2 + 1];
  jxo_scaled_idct(4, 4, block, px + afv_y * 4 * st + (afv_x == 1 ? 0 : 4), st);
  block[0] = dc2;
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 8; ix++)
      if (ix | iy) block[iy * 8 + ix] = co[(1 + iy * 2) * 8 + ix];
  jxo_scaled_idct(4, 8, block, px + (afv_y == 1 ? 0 : 4) * st, st);
}

/* AFVTransformFromPixels<kind> (enc_transforms-inl.h:411-455) */
static void afv_from_pixels(int kind, const float* px, size_t st, float* co) {
  const int afv_x = kind & 1, afv_y = kind / 2;
  float block[32] = {0}, coeff[16];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      block[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)] =
          px[(iy + 4 * afv_y) * st + ix + 4 * afv_x];
  afv_dct4x4(block, coeff);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++) co[iy * 2 * 8 + ix * 2] = coeff[iy * 4 + ix];
  jxo_scaled_dct(4, 4, px + afv_y * 4 * st + (afv_x == 1 ? 0 : 4), st, block);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      co[iy * 2 * 8 + ix * 2 + 1] = block[iy * 4 + ix];
  jxo_scaled_dct(4, 8, px + (afv_y == 1 ? 0 : 4) * st, st, block);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 8; ix++) co[(1 + iy * 2) * 8 + ix] = block[iy * 8 + ix];
  const float b00 = co[0] * 0.25f, b01 = co[1], b10 = co[8];
  co[0] = (b00 + b01 + 2 * b10) * 0.25f;
  co[1] = (b00 - b01) * 0.5f;
  co[8] = (b00 + b01 - 2 * b10) * 0.25f;
}

/* ------------------------------------------------------- TransformToPixels */
void jxo_transform_to_pixels(int strategy, float* co, float* px, size_t st) {
  switch (strategy) {
    case 1: { /* IDENTITY, dec_transforms-inl.h:463-499 */
      const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11,
                      b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float residual_sum = 0;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              if (ix | iy) residual_sum += co[(y + iy * 2) * 8 + x + ix * 2];
          const float base = dcs[y * 2 + x] - residual_sum * (1.0f / 16);
          px[(4 * y + 1) * st + 4 * x + 1] = base;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 1 && iy == 1) continue;
              px[(y * 4 + iy) * st + x * 4 + ix] =
                  co[(y + iy * 2) * 8 + x + ix * 2] + base;
            }
          px[y * 4 * st + x * 4] = co[(y + 2) * 8 + x + 2] + base;
        }
      return;
    }
    case 13: { /* DCT8X4, :500-519 */
      const float b0 = co[0], b1 = co[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int x = 0; x < 2; x++) {
        float block[32];
        block[0] = dcs[x];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            if (ix | iy) block[iy * 8 + ix] = co[(x + iy * 2) * 8 + ix];
        jxo_scaled_idct(8, 4, block, px + x * 4, st);
      }
      return;
    }
    case 12: { /* DCT4X8, :520-540 */
      const float b0 = co[0], b1 = co[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int y = 0; y < 2; y++) {
        float block[32];
        block[0] = dcs[y];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            if (ix | iy) block[iy * 8 + ix] = co[(y + iy * 2) * 8 + ix];
        jxo_scaled_idct(4, 8, block, px + y * 4 * st, st);
      }
      return;
    }
    case 3: { /* DCT4X4, :541-568 */
      const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11,
                      b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          block[0] = dcs[y * 2 + x];
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              if (ix | iy) block[iy * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2];
          jxo_scaled_idct(4, 4, block, px + y * 4 * st + x * 4, st);
        }
      return;
    }
    case 2: { /* DCT2X2, :569-581 */
      float c[64];
      memcpy(c, co, sizeof(c));
      idct2_top(2, c);
      idct2_top(4, c);
      idct2_top(8, c);
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) px[y * st + x] = c[y * 8 + x];
      return;
    }
    case 14: case 15: case 16: case 17:
      afv_to_pixels(strategy - 14, co, px, st);
      return;
    default: /* all plain DCTs, :582-687 */
      jxo_scaled_idct(8 * jxo_covered_blocks_y(strategy),
                      8 * jxo_covered_blocks_x(strategy), co, px, st);
  }
}

/* TransformFromPixels (enc_transforms-inl.h:457-670) -- test/generator only */
void jxo_transform_from_pixels(int strategy, const float* px, size_t st,
                               float* co) {
  switch (strategy) {
    case 1: {
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block_dc = 0;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              block_dc += px[(y * 4 + iy) * st + x * 4 + ix];
          block_dc *= 1.0f / 16;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 1 && iy == 1) continue;
              co[(y + iy * 2) * 8 + x + ix * 2] =
                  px[(y * 4 + iy) * st + x * 4 + ix] -
                  px[(y * 4 + 1) * st + x * 4 + 1];
            }
          co[(y + 2) * 8 + x + 2] = co[y * 8 + x];
          co[y * 8 + x] = block_dc;
        }
      break;
    }
    case 13: {
      for (int x = 0; x < 2; x++) {
        float block[32];
        jxo_scaled_dct(8, 4, px + x * 4, st, block);
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            co[(x + iy * 2) * 8 + ix] = block[iy * 8 + ix];
      }
      const float b0 = co[0], b1 = co[8];
      co[0] = (b0 + b1) * 0.5f;
      co[8] = (b0 - b1) * 0.5f;
      return;
    }
    case 12: {
      for (int y = 0; y < 2; y++) {
        float block[32];
        jxo_scaled_dct(4, 8, px + y * 4 * st, st, block);
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            co[(y + iy * 2) * 8 + ix] = block[iy * 8 + ix];
      }
      const float b0 = co[0], b1 = co[8];
      co[0] = (b0 + b1) * 0.5f;
      co[8] = (b0 - b1) * 0.5f;
      return;
    }
    case 3: {
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          jxo_scaled_dct(4, 4, px + y * 4 * st + x * 4, st, block);
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              co[(y + iy * 2) * 8 + x + ix * 2] = block[iy * 4 + ix];
        }
      break;
    }
    case 2: {
      dct2_top(8, px, st, co);
      dct2_top(4, co, 8, co);
      dct2_top(2, co, 8, co);
      return;
    }
    case 14: case 15: case 16: case 17:
      afv_from_pixels(strategy - 14, px, st, co);
      return;
    default:
      jxo_scaled_dct(8 * jxo_covered_blocks_y(strategy),
                     8 * jxo_covered_blocks_x(strategy), px, st, co);
      return;
  }
  /* IDENTITY and DCT4X4 share the 2x2 Hadamard of the four sub-DCs */
  const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
  co[0] = (b00 + b01 + b10 + b11) * 0.25f;
  co[1] = (b00 + b01 - b10 - b11) * 0.25f;
  co[8] = (b00 - b01 + b10 - b11) * 0.25f;
  co[9] = (b00 - b01 - b10 + b11) * 0.25f;
}
