/* oracle/filters.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates lib/jxl/render_pipeline/stage_gaborish.cc:31-99,
 * lib/jxl/render_pipeline/stage_epf.cc:47-494 (Weight, EPF0/1/2),
 * lib/jxl/dec_xyb-inl.h:38-86 (XybToRgb), lib/jxl/dec_xyb.cc:158-161
 * (opsin_biases_cbrt), with the border semantics of
 * lib/jxl/render_pipeline/simple_render_pipeline.cc:129-164 and
 * lib/jxl/image_ops.h:184-196 (Mirror). */
#include <math.h>
#include <string.h>

#include "jxl_oracle.h"

static inline int64_t mirror(int64_t x, int64_t n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}

typedef struct {
  const float* p;
  size_t stride;
  int64_t w, h;
} plane;
static inline float at(const plane* pl, int64_t x, int64_t y) {
  return pl->p[(size_t)mirror(y, pl->h) * pl->stride + (size_t)mirror(x, pl->w)];
}

/* stage_gaborish.cc:33-99 */
void jxo_gaborish(const jxo_frame* f, const float* const in[3],
                  float* const out[3], size_t stride, uint32_t row_begin,
                  uint32_t row_end) {
  const int64_t W = f->p.xsize, H = f->p.ysize;
  for (int c = 0; c < 3; c++) {
    float w0 = 1.0f, w1 = f->p.lf.gab_weights[2 * c], w2 = f->p.lf.gab_weights[2 * c + 1];
    const float div = w0 + 4 * (w1 + w2);
    const float mul = 1.0f / div;
    w0 *= mul;
    w1 *= mul;
    w2 *= mul;
    const plane pl = {in[c], stride, W, H};
    for (int64_t y = row_begin; y < row_end; y++)
      for (int64_t x = 0; x < W; x++) {
        const float t = at(&pl, x, y - 1), tl = at(&pl, x - 1, y - 1),
                    tr = at(&pl, x + 1, y - 1);
        const float m = at(&pl, x, y), l = at(&pl, x - 1, y), r = at(&pl, x + 1, y);
        const float b = at(&pl, x, y + 1), bl = at(&pl, x - 1, y + 1),
                    br = at(&pl, x + 1, y + 1);
        const float sum1 = (l + r) + (t + b);
        const float sum2 = (tl + tr) + (bl + br);
        out[c][(size_t)y * stride + x] = fmaf(sum2, w2, fmaf(sum1, w1, m * w0));
      }
  }
}

static inline float weight(float sad, float inv_sigma) {
  const float v = fmaf(sad, inv_sigma, 1.0f);
  return v < 0.0f ? 0.0f : v; /* ZeroIfNegative */
}

/* stage_epf.cc: EPF0 :82-181, EPF1 :225-367, EPF2 :416-494 */
void jxo_epf(const jxo_frame* f, int which, const float* inv_sigma_img,
             const float* const in[3], float* const out[3], size_t stride,
             uint32_t row_begin, uint32_t row_end) {
  const jxlhip_loop_filter* lf = &f->p.lf;
  const int64_t W = f->p.xsize, H = f->p.ysize;
  const uint32_t xsb = (f->p.xsize + 7) / 8;
  const float kMinSigma = -3.90524291751269967465540850526868f; /* epf.h:22 */
  float sm;
  if (which == 0) sm = (float)(lf->epf_pass0_sigma_scale * 1.65);
  else if (which == 1) sm = 1.65f;
  else sm = (float)(lf->epf_pass2_sigma_scale * 1.65);
  const float bsm = sm * lf->epf_border_sad_mul;
  const plane pl[3] = {{in[0], stride, W, H}, {in[1], stride, W, H}, {in[2], stride, W, H}};
  static const int8_t kOff0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1},
                                      {0, -2}, {0, -1},  {0, 1},  {0, 2},
                                      {1, -1}, {1, 0},   {1, 1},  {2, 0}};
  static const int8_t kPlus[5][2] = {{0, 0}, {-1, 0}, {0, -1}, {1, 0}, {0, 1}};
  for (int64_t y = row_begin; y < row_end; y++) {
    const int iy = (int)(y % 8);
    const int border_row = (iy == 0 || iy == 7);
    for (int64_t x = 0; x < W; x++) {
      const float is = inv_sigma_img[(size_t)(y / 8) * xsb + x / 8];
      const size_t o = (size_t)y * stride + x;
      if (is < kMinSigma) {
        for (int c = 0; c < 3; c++) out[c][o] = in[c][o];
        continue;
      }
      const int ix = (int)(x % 8);
      const float vsm = (border_row || ix == 0 || ix == 7) ? bsm : sm;
      const float inv_sigma = is * vsm;
      float wsum = 1.0f;
      float X = in[0][o], Y = in[1][o], B = in[2][o];
      if (which == 0) {
        float sads[12];
        for (int i = 0; i < 12; i++) sads[i] = 0.0f;
        for (int c = 0; c < 3; c++) {
          const float scale = lf->epf_channel_scale[c];
          for (int i = 0; i < 12; i++) {
            float sad = 0.0f;
            for (int k = 0; k < 5; k++) {
              const float r11 = at(&pl[c], x + kPlus[k][1], y + kPlus[k][0]);
              const float c11 = at(&pl[c], x + kOff0[i][1] + kPlus[k][1],
                                   y + kOff0[i][0] + kPlus[k][0]);
              sad = sad + fabsf(r11 - c11);
            }
            sads[i] = fmaf(sad, scale, sads[i]);
          }
        }
        for (int i = 0; i < 12; i++) {
          const float w = weight(sads[i], inv_sigma);
          wsum = wsum + w;
          X = fmaf(w, at(&pl[0], x + kOff0[i][1], y + kOff0[i][0]), X);
          Y = fmaf(w, at(&pl[1], x + kOff0[i][1], y + kOff0[i][0]), Y);
          B = fmaf(w, at(&pl[2], x + kOff0[i][1], y + kOff0[i][0]), B);
        }
      } else if (which == 1) {
        float sad0 = 0, sad1 = 0, sad2 = 0, sad3 = 0;
        for (int c = 0; c < 3; c++) {
          /* pXY naming of the reference: X = column (2 = centre), Y = row */
          const plane* q = &pl[c];
          const float p20 = at(q, x, y - 2), p21 = at(q, x, y - 1);
          float sad0c = fabsf(p20 - p21);
          const float p11 = at(q, x - 1, y - 1);
          float sad1c = fabsf(p11 - p21);
          const float p31 = at(q, x + 1, y - 1);
          float sad2c = fabsf(p31 - p21);
          const float p02 = at(q, x - 2, y), p12 = at(q, x - 1, y);
          sad1c = sad1c + fabsf(p02 - p12);
          sad0c = sad0c + fabsf(p11 - p12);
          const float p22 = at(q, x, y);
          float t = fabsf(p12 - p22);
          sad1c = sad1c + t;
          sad2c = sad2c + t;
          t = fabsf(p22 - p21);
          float sad3c = t;
          sad0c = sad0c + t;
          const float p32 = at(q, x + 1, y);
          sad0c = sad0c + fabsf(p31 - p32);
          t = fabsf(p22 - p32);
          sad1c = sad1c + t;
          sad2c = sad2c + t;
          const float p42 = at(q, x + 2, y);
          sad2c = sad2c + fabsf(p42 - p32);
          const float p13 = at(q, x - 1, y + 1);
          sad3c = sad3c + fabsf(p13 - p12);
          const float p23 = at(q, x, y + 1);
          t = fabsf(p22 - p23);
          sad0c = sad0c + t;
          sad3c = sad3c + t;
          sad1c = sad1c + fabsf(p13 - p23);
          const float p33 = at(q, x + 1, y + 1);
          sad2c = sad2c + fabsf(p33 - p23);
          sad3c = sad3c + fabsf(p33 - p32);
          const float p24 = at(q, x, y + 2);
          sad3c = sad3c + fabsf(p24 - p23);
          const float scale = lf->epf_channel_scale[c];
          sad0 = fmaf(sad0c, scale, sad0);
          sad1 = fmaf(sad1c, scale, sad1);
          sad2 = fmaf(sad2c, scale, sad2);
          sad3 = fmaf(sad3c, scale, sad3);
        }
        const float sads[4] = {sad0, sad1, sad2, sad3};
        static const int8_t kOff1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
        for (int i = 0; i < 4; i++) {
          const float w = weight(sads[i], inv_sigma);
          wsum = wsum + w;
          X = fmaf(w, at(&pl[0], x + kOff1[i][1], y + kOff1[i][0]), X);
          Y = fmaf(w, at(&pl[1], x + kOff1[i][1], y + kOff1[i][0]), Y);
          B = fmaf(w, at(&pl[2], x + kOff1[i][1], y + kOff1[i][0]), B);
        }
      } else {
        static const int8_t kOff2[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
        const float rx = X, ry = Y, rb = B;
        for (int i = 0; i < 4; i++) {
          const float cx = at(&pl[0], x + kOff2[i][1], y + kOff2[i][0]);
          const float cy = at(&pl[1], x + kOff2[i][1], y + kOff2[i][0]);
          const float cb = at(&pl[2], x + kOff2[i][1], y + kOff2[i][0]);
          float sad = fabsf(cx - rx) * lf->epf_channel_scale[0];
          sad = fmaf(fabsf(cy - ry), lf->epf_channel_scale[1], sad);
          sad = fmaf(fabsf(cb - rb), lf->epf_channel_scale[2], sad);
          const float w = weight(sad, inv_sigma);
          wsum = wsum + w;
          X = fmaf(w, cx, X);
          Y = fmaf(w, cy, Y);
          B = fmaf(w, cb, B);
        }
      }
      const float inv_w = 1.0f / wsum; /* JXL_HIGH_PRECISION: Div */
      out[0][o] = X * inv_w;
      out[1][o] = Y * inv_w;
      out[2][o] = B * inv_w;
    }
  }
}

/* dec_xyb-inl.h:38-86 */
void jxo_xyb_to_linear_rgb(const jxo_frame* f, const float* const in[3],
                           size_t stride, float* rgb, size_t rgb_stride,
                           uint32_t row_begin, uint32_t row_end) {
  const jxlhip_frame_params* p = &f->p;
  float cb[3];
  for (int i = 0; i < 3; i++) cb[i] = cbrtf(p->opsin_biases[i]);
  const float* m = p->inverse_opsin_matrix;
  for (uint32_t y = row_begin; y < row_end; y++)
    for (uint32_t x = 0; x < p->xsize; x++) {
      const size_t o = (size_t)y * stride + x;
      const float ox = in[0][o], oy = in[1][o], ob = in[2][o];
      float gr = oy + ox, gg = oy - ox, gb = ob;
      gr = gr - cb[0];
      gg = gg - cb[1];
      gb = gb - cb[2];
      const float mr = fmaf(gr * gr, gr, p->opsin_biases[0]);
      const float mg = fmaf(gg * gg, gg, p->opsin_biases[1]);
      const float mb = fmaf(gb * gb, gb, p->opsin_biases[2]);
      float* dst = rgb + (size_t)y * rgb_stride + 3 * (size_t)x;
      dst[0] = fmaf(m[2], mb, fmaf(m[1], mg, m[0] * mr));
      dst[1] = fmaf(m[5], mb, fmaf(m[4], mg, m[3] * mr));
      dst[2] = fmaf(m[8], mb, fmaf(m[7], mg, m[6] * mr));
    }
}

/* LinearRGBToXYB (enc_xyb.cc:83-105) with cms/opsin_params.h:19-43; for
 * generators and the opsin-inverse KAT only. */
void jxo_linear_rgb_to_xyb(float r, float g, float b, float xyb[3]) {
  const float kM02 = 0.078f, kM00 = 0.30f, kM01 = 1.0f - kM02 - kM00;
  const float kM12 = 0.078f, kM10 = 0.23f, kM11 = 1.0f - kM12 - kM10;
  const float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
              kM22 = 1.0f - kM20 - kM21;
  const float bias = 0.0037930732552754493f;
  float mr = kM00 * r + kM01 * g + kM02 * b + bias;
  float mg = kM10 * r + kM11 * g + kM12 * b + bias;
  float mb = kM20 * r + kM21 * g + kM22 * b + bias;
  mr = mr < 0 ? 0 : mr;
  mg = mg < 0 ? 0 : mg;
  mb = mb < 0 ? 0 : mb;
  const float nb = -cbrtf(bias);
  mr = cbrtf(mr) + nb;
  mg = cbrtf(mg) + nb;
  mb = cbrtf(mb) + nb;
  xyb[0] = 0.5f * (mr - mg);
  xyb[1] = 0.5f * (mr + mg);
  xyb[2] = mb;
}
