/* oracle/group.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates lib/jxl/quantizer-inl.h:34-67 (AdjustQuantBias),
 * lib/jxl/dec_group.cc:115-181 (DequantLane/DequantBlock), :183-457
 * (DecodeGroupImpl block walk), lib/jxl/chroma_from_luma.h:51-57,
 * lib/jxl/frame_dimensions.h:34-77, lib/jxl/epf.cc:39-133 (ComputeSigma). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jxl_oracle.h"

/* quantizer-inl.h:34-67.  ApproximateReciprocal is target dependent in the
 * reference (rcpps on x86, exact on HWY_SCALAR); the oracle uses the exact
 * reciprocal, as the scalar target does. */
float jxo_adjust_quant_bias(int c, int32_t q, const float biases[4]) {
  const float quant = (float)q;
  const float abs_quant = fabsf(quant);
  if (abs_quant < 1.125f) {
    if (!(abs_quant > 0.0f)) return 0.0f;
    return q < 0 ? -biases[c] : biases[c];
  }
  return fmaf(-biases[3], 1.0f / quant, quant);
}

static inline int32_t load_coeff(const void* base, uint32_t type, size_t i) {
  return type == JXLHIP_COEFF_I16 ? (int32_t)((const int16_t*)base)[i]
                                  : ((const int32_t*)base)[i];
}

/* One group: dec_group.cc:275-455 (non-JPEG, 4:4:4, single pass) */
static int decode_group(const jxo_frame* f, uint32_t g, float* const xyb[3],
                        size_t row_stride, float* block /*3*max area*/) {
  const jxlhip_frame_params* p = &f->p;
  const uint32_t xsb = (p->xsize + 7) / 8, ysb = (p->ysize + 7) / 8;
  const uint32_t xsg = (p->xsize + 255) / 256;
  const uint32_t xtiles = (xsb + 7) / 8;
  const uint32_t gx = g % xsg, gy = g / xsg;
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  const uint32_t gw = xsb - bx0 < 32 ? xsb - bx0 : 32; /* BlockGroupRect */
  const uint32_t gh = ysb - by0 < 32 ? ysb - by0 : 32;
  const float inv_global_scale = (float)(1.0 * 65536 / p->global_scale);
  const float color_scale = 1.0f / (float)p->cfl_color_factor;
  size_t offset = 0;
  for (uint32_t by = 0; by < gh; by++) {
    const uint32_t aby = by0 + by;
    for (uint32_t bx = 0; bx < gw;) {
      const uint32_t abx = bx0 + bx;
      const uint8_t raw = f->ac_strategy[(size_t)aby * xsb + abx];
      const int strategy = raw >> 1;
      if (strategy >= JXLHIP_NUM_STRATEGIES) return -1;
      const uint32_t cx = jxo_covered_blocks_x(strategy);
      const uint32_t cy = jxo_covered_blocks_y(strategy);
      if (!(raw & 1)) {
        bx += cx;
        continue;
      }
      if (bx + cx > gw || by + cy > gh) return -1;
      const size_t size = (size_t)64 << jxo_log2_covered_blocks(strategy);
      if (offset + size > JXLHIP_GROUP_COEFFS) return -1;
      /* CfL factors of the tile holding the first block (dec_group.cc:288,
       * 312-316) */
      const size_t tile = (size_t)(aby / 8) * xtiles + abx / 8;
      const float x_cc = p->cfl_base_x + f->ytox_map[tile] * color_scale;
      const float b_cc = p->cfl_base_b + f->ytob_map[tile] * color_scale;
      /* DequantBlock */
      const int quant = f->raw_quant[(size_t)aby * xsb + abx];
      const float s = inv_global_scale / quant;
      const float sx = s * p->x_dm_multiplier, sy = s, sb = s * p->b_dm_multiplier;
      const float* mx = f->dequant_table + jxo_dequant_table_offset(strategy, 0);
      const float* my = mx + size;
      const float* mb = mx + 2 * size;
      const size_t base = (size_t)g * JXLHIP_GROUP_COEFFS + offset;
      for (size_t k = 0; k < size; k++) {
        const float x_mul = mx[k] * sx, y_mul = my[k] * sy, b_mul = mb[k] * sb;
        const int32_t qx = load_coeff(f->coeffs[0], p->coeff_type, base + k);
        const int32_t qy = load_coeff(f->coeffs[1], p->coeff_type, base + k);
        const int32_t qb = load_coeff(f->coeffs[2], p->coeff_type, base + k);
        const float dx = jxo_adjust_quant_bias(0, qx, p->quant_biases) * x_mul;
        const float dy = jxo_adjust_quant_bias(1, qy, p->quant_biases) * y_mul;
        const float db = jxo_adjust_quant_bias(2, qb, p->quant_biases) * b_mul;
        block[k] = fmaf(x_cc, dy, dx);
        block[size + k] = dy;
        block[2 * size + k] = fmaf(b_cc, dy, db);
      }
      for (int c = 0; c < 3; c++)
        jxo_llf_from_dc(strategy, f->dc[c] + (size_t)aby * xsb + abx, xsb,
                        block + c * size);
      /* IDCT, Y first as the reference (order is irrelevant numerically) */
      static const int kOrder[3] = {1, 0, 2};
      for (int i = 0; i < 3; i++) {
        const int c = kOrder[i];
        jxo_transform_to_pixels(strategy, block + c * size,
                                xyb[c] + (size_t)aby * 8 * row_stride + abx * 8,
                                row_stride);
      }
      offset += size;
      bx += cx;
    }
  }
  return 0;
}

int jxo_decode_groups(const jxo_frame* f, float* const xyb[3],
                      size_t row_stride, uint32_t group_begin,
                      uint32_t group_end) {
  float* block = (float*)malloc(sizeof(float) * 3 * 256 * 256);
  if (!block) return -1;
  int rc = 0;
  for (uint32_t g = group_begin; g < group_end && rc == 0; g++)
    rc = decode_group(f, g, xyb, row_stride, block);
  free(block);
  return rc;
}

/* epf.cc:39-133, without the mirrored padding frame (the Simple pipeline
 * never reads it: xextra = 0).  kInvSigmaNum epf.h:18. */
void jxo_compute_sigma(const jxo_frame* f, float* inv_sigma) {
  const jxlhip_frame_params* p = &f->p;
  const uint32_t xsb = (p->xsize + 7) / 8, ysb = (p->ysize + 7) / 8;
  const float kInvSigmaNum = -1.1715728752538099024f;
  const float quant_scale = (float)(p->global_scale * (1.0 / 65536));
  for (uint32_t by = 0; by < ysb; by++)
    for (uint32_t bx = 0; bx < xsb; bx++) {
      const uint8_t raw = f->ac_strategy[(size_t)by * xsb + bx];
      if (!(raw & 1)) continue;
      const int s = raw >> 1;
      const float sigma_quant =
          p->lf.epf_quant_mul /
          (quant_scale * f->raw_quant[(size_t)by * xsb + bx] * kInvSigmaNum);
      for (int iy = 0; iy < jxo_covered_blocks_y(s); iy++)
        for (int ix = 0; ix < jxo_covered_blocks_x(s); ix++) {
          const size_t i = (size_t)(by + iy) * xsb + bx + ix;
          float sigma = sigma_quant * p->lf.epf_sharp_lut[f->epf_sharpness[i]];
          sigma = sigma < -1e-4f ? sigma : -1e-4f;
          inv_sigma[i] = 1.0f / sigma;
        }
    }
}
