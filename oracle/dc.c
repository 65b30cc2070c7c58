/* oracle/dc.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates lib/jxl/compressed_dc.cc:49-53 (weights), :63-126 (ComputePixel),
 * :128-197 (AdaptiveDCSmoothing), :201-232 (DequantDC, 4:4:4 branch). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jxl_oracle.h"

void jxo_dequant_dc(uint32_t xsb, uint32_t ysb,
                    const int32_t* const quant_dc[3], float* const dc[3],
                    const float mul_dc[3], float cfl_x_dc, float cfl_b_dc) {
  const size_t n = (size_t)xsb * ysb;
  for (size_t i = 0; i < n; i++) {
    const float in_x = (float)quant_dc[0][i] * mul_dc[0];
    const float in_y = (float)quant_dc[1][i] * mul_dc[1];
    const float in_b = (float)quant_dc[2][i] * mul_dc[2];
    dc[1][i] = in_y;
    dc[0][i] = fmaf(in_y, cfl_x_dc, in_x);
    dc[2][i] = fmaf(in_y, cfl_b_dc, in_b);
  }
}

void jxo_adaptive_dc_smoothing(uint32_t xs, uint32_t ys, const float mul_dc[3],
                               float* const dc[3]) {
  if (ys <= 2 || xs <= 2) return;
  const float w1 = 0.20345139757231578f, w2 = 0.0334829185968739f;
  const float w0 = 1.0f - 4.0f * (w1 + w2);
  const size_t n = (size_t)xs * ys;
  float* sm[3];
  for (int c = 0; c < 3; c++) {
    sm[c] = (float*)malloc(n * sizeof(float));
    memcpy(sm[c], dc[c], n * sizeof(float)); /* borders stay unsmoothed */
  }
  for (uint32_t y = 1; y + 1 < ys; y++)
    for (uint32_t x = 1; x + 1 < xs; x++) {
      float gap = 0.5f, mcv[3], smv[3];
      for (int c = 0; c < 3; c++) {
        const float* r0 = dc[c] + (size_t)(y - 1) * xs + x;
        const float* r1 = dc[c] + (size_t)y * xs + x;
        const float* r2 = dc[c] + (size_t)(y + 1) * xs + x;
        const float corner = (r0[-1] + r0[1]) + (r2[-1] + r2[1]);
        const float side = (r1[-1] + r1[1]) + (r0[0] + r2[0]);
        mcv[c] = r1[0];
        smv[c] = fmaf(corner, w2, fmaf(side, w1, mcv[c] * w0));
        const float g = fabsf((mcv[c] - smv[c]) / mul_dc[c]);
        gap = g > gap ? g : gap;
      }
      float factor = fmaf(-4.0f, gap, 3.0f);
      factor = factor < 0.0f ? 0.0f : factor;
      for (int c = 0; c < 3; c++)
        sm[c][(size_t)y * xs + x] = fmaf(smv[c] - mcv[c], factor, mcv[c]);
    }
  for (int c = 0; c < 3; c++) {
    memcpy(dc[c], sm[c], n * sizeof(float));
    free(sm[c]);
  }
}
