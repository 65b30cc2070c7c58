/* oracle/dct.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Restates: lib/jxl/ac_strategy.h:148-173 (LUTs), lib/jxl/quant_weights.h:302-348,
 * 364-367,401-417 (table map/offsets), lib/jxl/dct-inl.h:45-232,351-397 (fast
 * DCT/IDCT), lib/jxl/dct_scales.h:234-369 (WcMultipliers closed form),
 * lib/jxl/dct_for_test.h:20-95 (f64 matrix DCT). */
#include <math.h>
#include <string.h>

#include "jxl_oracle.h"

/* ------------------------------------------------------------------ LUTs */
static const uint8_t kCoveredX[27] = {1, 1, 1, 1, 2,  4, 1,  2,  1,
                                      4, 2, 4, 1, 1,  1, 1,  1,  1,
                                      8, 4, 8, 16, 8, 16, 32, 16, 32};
static const uint8_t kCoveredY[27] = {1, 1, 1, 1, 2,  4,  2, 1,  4,
                                      1, 4, 2, 1, 1,  1,  1, 1,  1,
                                      8, 8, 4, 16, 16, 8, 32, 32, 16};
/* strategy -> quant table kind (17 kinds; RxC and CxR share one) */
static const uint8_t kQuantKind[27] = {0,  1,  2,  3,  4,  5,  6,  6,  7,
                                       7,  8,  8,  9,  9,  10, 10, 10, 10,
                                       11, 12, 12, 13, 14, 14, 15, 16, 16};
/* 8x8-block footprint of each table kind: short side, long side */
static const uint8_t kKindShort[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2,
                                       1, 1, 8, 4, 16, 8, 32, 16};
static const uint8_t kKindLong[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4,
                                      1, 1, 8, 8, 16, 16, 32, 32};

int jxo_covered_blocks_x(int s) { return kCoveredX[s]; }
int jxo_covered_blocks_y(int s) { return kCoveredY[s]; }
int jxo_log2_covered_blocks(int s) {
  int n = kCoveredX[s] * kCoveredY[s], l = 0;
  while ((1 << l) < n) l++;
  return l;
}
int jxo_quant_table_of_strategy(int s) { return kQuantKind[s]; }
size_t jxo_dequant_table_offset(int s, int c) {
  size_t pos = 0;
  int kind = kQuantKind[s];
  for (int k = 0; k < kind; k++) pos += 3u * 64u * kKindShort[k] * kKindLong[k];
  return pos + (size_t)c * 64u * kKindShort[kind] * kKindLong[kind];
}

/* ------------------------------------------------------ twiddle factors */
#define JXO_MAX_DCT 256
static float g_wc[JXO_MAX_DCT * 2]; /* W_N[i] stored at g_wc[N + i], i < N/2 */
static const float kSqrt2f = 1.41421356237f; /* dct_scales.h:15 */

__attribute__((constructor)) static void jxo_init_wc(void) {
  for (int n = 4; n <= JXO_MAX_DCT; n *= 2) {
    for (int i = 0; i < n / 2; i++) {
      g_wc[n + i] = (float)(1.0 / (2.0 * cos((i + 0.5) * M_PI / n)));
    }
  }
}

/* ------------------------------------------------------------ fast IDCT */
/* IDCT1DImpl<N>::operator() (dct-inl.h:191-232) on ONE column. */
static void idct_rec(int n, float* v /* n values, in place */, float* tmp) {
  if (n == 1) return;
  if (n == 2) {
    float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
    return;
  }
  const int h = n / 2;
  /* ForwardEvenOdd: evens first, then odds */
  for (int i = 0; i < h; i++) tmp[i] = v[2 * i];
  for (int i = 0; i < h; i++) tmp[h + i] = v[2 * i + 1];
  idct_rec(h, tmp, tmp + n);
  /* BTranspose on the odd half */
  for (int i = h - 1; i > 0; i--) tmp[h + i] = tmp[h + i] + tmp[h + i - 1];
  tmp[h] = tmp[h] * kSqrt2f;
  idct_rec(h, tmp + h, tmp + n);
  /* MultiplyAndAdd */
  for (int i = 0; i < h; i++) {
    const float mul = g_wc[n + i];
    const float e = tmp[i], o = tmp[h + i];
    v[i] = fmaf(mul, o, e);
    v[n - 1 - i] = fmaf(-mul, o, e);
  }
}

void jxo_idct1d(int n, const float* from, size_t from_stride, float* to,
                size_t to_stride) {
  float v[JXO_MAX_DCT], tmp[2 * JXO_MAX_DCT];
  for (int i = 0; i < n; i++) v[i] = from[i * from_stride];
  idct_rec(n, v, tmp);
  for (int i = 0; i < n; i++) to[i * to_stride] = v[i];
}

/* DCT1DImpl<N>::operator() (dct-inl.h:158-189), unscaled, one column */
static void dct_rec(int n, float* v, float* tmp) {
  if (n == 1) return;
  if (n == 2) {
    float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
    return;
  }
  const int h = n / 2;
  for (int i = 0; i < h; i++) tmp[i] = v[i] + v[n - 1 - i];
  dct_rec(h, tmp, tmp + n);
  for (int i = 0; i < h; i++) tmp[h + i] = v[i] - v[n - 1 - i];
  for (int i = 0; i < h; i++) tmp[h + i] = tmp[h + i] * g_wc[n + i];
  dct_rec(h, tmp + h, tmp + n);
  /* B */
  tmp[h] = fmaf(tmp[h], kSqrt2f, tmp[h + 1]);
  for (int i = 1; i + 1 < h; i++) tmp[h + i] = tmp[h + i] + tmp[h + i + 1];
  /* InverseEvenOdd */
  for (int i = 0; i < h; i++) {
    v[2 * i] = tmp[i];
    v[2 * i + 1] = tmp[h + i];
  }
}

void jxo_dct1d(int n, float* mem, size_t stride) {
  float v[JXO_MAX_DCT], tmp[2 * JXO_MAX_DCT];
  for (int i = 0; i < n; i++) v[i] = mem[i * stride];
  dct_rec(n, v, tmp);
  for (int i = 0; i < n; i++) mem[i * stride] = v[i];
}

/* columns of an (n x m) row-major matrix, n-point transform down each column */
static void idct_columns(int n, int m, const float* from, float* to) {
  for (int x = 0; x < m; x++) jxo_idct1d(n, from + x, m, to + x, m);
}
static void dct_columns_scaled(int n, int m, const float* from,
                               size_t from_stride, float* to) {
  /* DCT1DWrapper: LoadFromBlock, DCT1DImpl, StoreToBlockAndScale (1/N) */
  const float scale = 1.0f / n;
  float v[JXO_MAX_DCT], tmp[2 * JXO_MAX_DCT];
  for (int x = 0; x < m; x++) {
    for (int i = 0; i < n; i++) v[i] = from[i * from_stride + x];
    dct_rec(n, v, tmp);
    for (int i = 0; i < n; i++) to[i * m + x] = scale * v[i];
  }
}
static void transpose(int rows, int cols, const float* from, float* to) {
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) to[x * rows + y] = from[y * cols + x];
}

/* ComputeScaledIDCT<ROWS,COLS> (dct-inl.h:376-397) */
void jxo_scaled_idct(int rows, int cols, float* from, float* pixels,
                     size_t pixels_stride) {
  static __thread float block[JXO_MAX_DCT * JXO_MAX_DCT];
  static __thread float last[JXO_MAX_DCT * JXO_MAX_DCT];
  if (rows < cols) {
    transpose(rows, cols, from, block);      /* cols x rows */
    idct_columns(cols, rows, block, from);   /* [x][u]      */
    transpose(cols, rows, from, block);      /* rows x cols */
    idct_columns(rows, cols, block, last);
  } else {
    idct_columns(cols, rows, from, block);   /* from: cols x rows */
    transpose(cols, rows, block, from);      /* rows x cols */
    idct_columns(rows, cols, from, last);
  }
  for (int y = 0; y < rows; y++)
    memcpy(pixels + (size_t)y * pixels_stride, last + (size_t)y * cols,
           sizeof(float) * cols);
}

/* ComputeScaledDCT<ROWS,COLS> (dct-inl.h:353-373) */
void jxo_scaled_dct(int rows, int cols, const float* pixels,
                    size_t pixels_stride, float* to) {
  static __thread float block[JXO_MAX_DCT * JXO_MAX_DCT];
  if (rows < cols) {
    dct_columns_scaled(rows, cols, pixels, pixels_stride, block);
    transpose(rows, cols, block, to); /* cols x rows */
    dct_columns_scaled(cols, rows, to, rows, block);
    transpose(cols, rows, block, to); /* rows x cols */
  } else {
    dct_columns_scaled(rows, cols, pixels, pixels_stride, to);
    transpose(rows, cols, to, block); /* cols x rows */
    dct_columns_scaled(cols, rows, block, rows, to);
  }
}

/* ------------------------------------------------- f64 reference (tests) */
/* dct_for_test.h: alpha(0)=1 else sqrt2; DCT: out[u]=sum_x in[x]*cos(..)*alpha(u)/N
 * IDCT: out[x] = sum_u in[u]*alpha(u)*cos((x+.5)u pi/N) */
void jxo_dct1d_slow(int n, const double* in, double* out) {
  for (int u = 0; u < n; u++) {
    double s = 0, a = u == 0 ? 1.0 : sqrt(2.0);
    for (int x = 0; x < n; x++) s += in[x] * a * cos((x + 0.5) * u * M_PI / n);
    out[u] = s / n;
  }
}
void jxo_idct1d_slow(int n, const double* in, double* out) {
  for (int x = 0; x < n; x++) {
    double s = 0;
    for (int u = 0; u < n; u++) {
      double a = u == 0 ? 1.0 : sqrt(2.0);
      s += in[u] * a * cos((x + 0.5) * u * M_PI / n);
    }
    out[x] = s;
  }
}
