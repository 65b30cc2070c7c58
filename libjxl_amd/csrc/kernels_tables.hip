// kernels_tables.hip -- SURVEY 8(a) rows a5 and a8 on the device:
//   k_dequant_tables  : DequantMatrices::EnsureComputed / ComputeQuantTable for 17
//                       resolved QuantEncodings (lib/jxl/quant_weights.cc:48-160,
//                       163-358,1190-1271; FastPowf lib/jxl/base/fast_math-inl.h:46-92),
//                       one thread per table entry
//   k_dequant_dc      : DequantDC, 4:4:4 (lib/jxl/compressed_dc.cc:201-232)
//   k_smooth_dc       : AdaptiveDCSmoothing (lib/jxl/compressed_dc.cc:63-197)
#include "dev_common.h"
#include <atomic>

#include "kernels.h"

namespace jxlhip {
__device__ const uint8_t dKindShort[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
__device__ const uint8_t dKindLong[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};

__device__ float Mult(float v) { return v > 0.0f ? 1.0f + v : 1.0f / (1.0f - v); }

// distance-band weight of coefficient (y, x) of a rows x cols table
__device__ float DctWeight(int rows, int cols, const float* bands_in, int nb, int y, int x,
                           bool* ok) {
  float bands[JXLHIP_MAX_DISTANCE_BANDS];
  bands[0] = bands_in[0];
  if (bands[0] < 1e-8f) *ok = false;
  for (int i = 1; i < nb; i++) {
    bands[i] = bands[i - 1] * Mult(bands_in[i]);
    if (bands[i] < 1e-8f) *ok = false;
  }
  if (nb == 1) return bands[0];
  const float scale = (nb - 1) / (kSqrt2 + 1e-6f);
  const float rcpcol = scale / (cols - 1);
  const float rcprow = scale / (rows - 1);
  const float dy = y * rcprow;
  const float dy2 = dy * dy;
  const float dx = (float)x * rcpcol;
  const float dist = __builtin_sqrtf(__builtin_fmaf(dx, dx, dy2));
  const int32_t idx = (int32_t)dist;
  const float frac = dist - (float)idx;
  const float a = bands[idx], b = bands[idx + 1];
  return a * FastPowf(b / a, frac);
}

__device__ float Interpolate(float pos, float max, const float* array, int len) {
  const float scaled_pos = pos * (len - 1) / max;
  const int idx = (int)scaled_pos;
  const float a = array[idx], b = array[idx + 1];
  return a * FastPowf(b / a, scaled_pos - idx);
}

__global__ __launch_bounds__(256) void k_dequant_tables(float* __restrict__ table,
                                                        const jxlhip_quant_encoding* __restrict__ enc,
                                                        int32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= JXLHIP_DEQUANT_TABLE_FLOATS) return;
  // locate (kind, c, y, x)
  uint32_t pos = 0;
  int kind = 0;
  for (; kind < 17; kind++) {
    const uint32_t n = 3u * 64u * dKindShort[kind] * dKindLong[kind];
    if (i < pos + n) break;
    pos += n;
  }
  const int wrows = 8 * dKindShort[kind], wcols = 8 * dKindLong[kind];
  const int num = wrows * wcols;
  const int c = (int)(i - pos) / num;
  const int k = (int)(i - pos) % num;
  const int y = k / wcols, x = k % wcols;
  const jxlhip_quant_encoding& e = enc[kind];
  const int nb = (int)e.num_bands;
  bool ok = true;
  float w;
  switch (e.mode) {
    case JXLHIP_QUANT_ID:
      w = (k == 1 || k == 8) ? e.weights[c][1] : (k == 9 ? e.weights[c][2] : e.weights[c][0]);
      break;
    case JXLHIP_QUANT_DCT2:
      if (y < 2 && x < 2) {
        w = k == 0 ? (float)0xBAD : (k == 9 ? e.weights[c][1] : e.weights[c][0]);
      } else if (y < 4 && x < 4) {
        w = (y >= 2 && x >= 2) ? e.weights[c][3] : e.weights[c][2];
      } else {
        w = (y >= 4 && x >= 4) ? e.weights[c][5] : e.weights[c][4];
      }
      break;
    case JXLHIP_QUANT_DCT4:
      w = DctWeight(4, 4, e.bands[c], nb, y / 2, x / 2, &ok);
      if (k == 1 || k == 8) w /= e.weights[c][0];
      if (k == 9) w /= e.weights[c][1];
      break;
    case JXLHIP_QUANT_DCT4X8:
      w = DctWeight(4, 8, e.bands[c], nb, y / 2, x, &ok);
      if (k == 8) w /= e.weights[c][0];
      break;
    case JXLHIP_QUANT_DCT:
      w = DctWeight(wrows, wcols, e.bands[c], nb, y, x, &ok);
      break;
    default: {  // AFV
      const float kFreqs[16] = {0xBAD, 0xBAD, 0.8517778890324296f, 5.37778436506804f,
                                0xBAD, 0xBAD, 4.734747904497923f, 5.449245381693219f,
                                1.6598270267479331f, 4.0f, 7.275749096817861f,
                                10.423227632456525f, 2.662932286148962f, 7.630657783650829f,
                                8.962388608184032f, 12.97166202570235f};
      const float lo = 0.8517778890324296f;
      const float hi = 12.97166202570235f - lo + 1e-6f;
      float bands[4];
      bands[0] = e.weights[c][5];
      if (bands[0] < 1e-8f) ok = false;
      for (int j = 1; j < 4; j++) {
        bands[j] = bands[j - 1] * Mult(e.weights[c][j + 5]);
        if (bands[j] < 1e-8f) ok = false;
      }
      if (y & 1) {  // odd rows: the 4x8 DCT part
        w = DctWeight(4, 8, e.bands[c], nb, y / 2, x, &ok);
      } else if (x & 1) {  // even rows, odd columns: the 4x4 DCT part
        w = DctWeight(4, 4, e.bands_afv_4x4[c], (int)e.num_bands_afv_4x4, y / 2, x / 2, &ok);
      } else {  // even rows, even columns: the AFV part
        const int ay = y / 2, ax = x / 2;
        if (ay < 2 && ax < 2) w = 0;  // placeholders, fixed up below
        else w = Interpolate(kFreqs[ay * 4 + ax] - lo, hi, bands, 4);
      }
      if (k == 0) w = 1;
      if (k == 1 * 8 + 0) w = e.weights[c][0];
      if (k == 0 * 8 + 1) w = e.weights[c][1];
      if (k == 2 * 8 + 0) w = e.weights[c][2];
      if (k == 0 * 8 + 2) w = e.weights[c][3];
      if (k == 2 * 8 + 2) w = e.weights[c][4];
      break;
    }
  }
  if (!ok || w >= 1.0f / 1e-8f || w < 1e-8f) {
    atomicOr(status, 1);
    w = 1.0f;
  }
  table[i] = 1.0f / w;
}

// extra_precision (optional): one byte per DC group (2048x2048 px = 256x256 blocks): the group's
// factors are multiplied by 1 / (1 << extra_precision) (DecodeVarDCTDC, dec_modular.cc:443-445)
__global__ __launch_bounds__(256) void k_dequant_dc(uint32_t xs, uint32_t ys, const int32_t* __restrict__ qx,
                                                    const int32_t* __restrict__ qy,
                                                    const int32_t* __restrict__ qb,
                                                    float* __restrict__ ox, float* __restrict__ oy,
                                                    float* __restrict__ ob, float mx, float my,
                                                    float mb, float cfl_x, float cfl_b,
                                                    const uint8_t* __restrict__ extra_precision) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)xs * ys) return;
  float mul = 1.0f;
  if (extra_precision) {
    const uint32_t y = (uint32_t)(i / xs), x = (uint32_t)(i % xs);
    const uint32_t p = extra_precision[(size_t)(y >> 8) * ((xs + 255) >> 8) + (x >> 8)] & 3u;
    mul = 1.0f / (float)(1u << p);
  }
  const float in_x = (float)qx[i] * (mx * mul);
  const float in_y = (float)qy[i] * (my * mul);
  const float in_b = (float)qb[i] * (mb * mul);
  oy[i] = in_y;
  ox[i] = __builtin_fmaf(in_y, cfl_x, in_x);
  ob[i] = __builtin_fmaf(in_y, cfl_b, in_b);
}

struct DcPlanes {
  const float* in[3];
  float* out[3];
  float mul[3];
};

__global__ __launch_bounds__(256) void k_smooth_dc(uint32_t xs, uint32_t ys, DcPlanes p) {
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
  const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= xs || y >= ys) return;
  const size_t o = (size_t)y * xs + x;
  if (x == 0 || y == 0 || x + 1 == xs || y + 1 == ys) {
    for (int c = 0; c < 3; c++) p.out[c][o] = p.in[c][o];
    return;
  }
  const float w1 = 0.20345139757231578f, w2 = 0.0334829185968739f;
  const float w0 = 1.0f - 4.0f * (w1 + w2);
  float gap = 0.5f, mcv[3], smv[3];
  for (int c = 0; c < 3; c++) {
    const float* r0 = p.in[c] + o - xs;
    const float* r1 = p.in[c] + o;
    const float* r2 = p.in[c] + o + xs;
    const float corner = (r0[-1] + r0[1]) + (r2[-1] + r2[1]);
    const float side = (r1[-1] + r1[1]) + (r0[0] + r2[0]);
    mcv[c] = r1[0];
    smv[c] = __builtin_fmaf(corner, w2, __builtin_fmaf(side, w1, mcv[c] * w0));
    const float g = __builtin_fabsf((mcv[c] - smv[c]) / p.mul[c]);
    gap = g > gap ? g : gap;
  }
  float factor = __builtin_fmaf(-4.0f, gap, 3.0f);
  factor = factor < 0.0f ? 0.0f : factor;
  for (int c = 0; c < 3; c++)
    p.out[c][o] = __builtin_fmaf(smv[c] - mcv[c], factor, mcv[c]);
}

// rows [y_first, y_first + nrows) x columns [0, ncols) of the block-major
// planes <-> dense row-major staging (halo exchange, test export)
__global__ __launch_bounds__(256) void k_rows_copy(DevFrame f, float* dense, int y_first,
                                                   int nrows, int ncols, int to_dense,
                                                   size_t dense_stride, size_t dense_plane) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y, c = blockIdx.z;
  if (x >= ncols || r >= nrows) return;
  float* src = f.xyb[c] + PlaneOffset(f, y_first + r, x);
  float* d = dense + (size_t)c * dense_plane + (size_t)r * dense_stride + x;
  if (to_dense) *d = *src;
  else *src = *d;
}

void LaunchRowsCopy(const DevFrame& f, float* dense, int y_first, int nrows, int ncols,
                    size_t dense_stride, size_t dense_plane, int nch, bool to_dense,
                    hipStream_t st) {
  if (nrows <= 0 || ncols <= 0) return;
  hipLaunchKernelGGL(k_rows_copy, dim3((unsigned)((ncols + 255) / 256), (unsigned)nrows, nch),
                     dim3(256), 0, st, f, dense, y_first, nrows, ncols, to_dense ? 1 : 0,
                     dense_stride, dense_plane);
}

// WriteToOutputStage's undo_orientation (stage_write.cc:441-457: flip_x for orientations 2, 3, 7, 8; flip_y for 3, 4,
// 6, 7; transpose for 5..8; the flips act in the coded frame, the transpose last, :486,:341,:664-680): pixel (x, y)
// of the coded W x H frame goes to row / column (y', x') -- or (x', y') when transposed -- with x' = W-1-x, y' = H-1-y
// where flipped.  One thread per pixel; a 32 x 8 tile of threads reads rows and, when transposed, writes columns of
// 32-byte-or-more pixels: sector-sized pieces either way.  BPP = bytes per pixel (3 .. 16).
template <int BPP>
__global__ __launch_bounds__(256) void k_orient(const unsigned char* __restrict__ src, size_t src_stride, int W, int H,
                                                unsigned char* __restrict__ dst, size_t dst_stride, int flip_x,
                                                int flip_y, int transpose) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const int xo = flip_x ? W - 1 - x : x, yo = flip_y ? H - 1 - y : y;
  const unsigned char* s = src + (size_t)y * src_stride + (size_t)x * BPP;
  unsigned char* d = transpose ? dst + (size_t)xo * dst_stride + (size_t)yo * BPP
                               : dst + (size_t)yo * dst_stride + (size_t)xo * BPP;
  if constexpr (BPP % 4 == 0) {
#pragma unroll
    for (int i = 0; i < BPP / 4; i++) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
  } else if constexpr (BPP % 2 == 0) {
#pragma unroll
    for (int i = 0; i < BPP / 2; i++) ((uint16_t*)d)[i] = ((const uint16_t*)s)[i];
  } else {
#pragma unroll
    for (int i = 0; i < BPP; i++) d[i] = s[i];
  }
}

bool LaunchOrient(const void* src, size_t src_stride, uint32_t xsize, uint32_t ysize, uint32_t bytes_per_pixel,
                  uint32_t orientation, void* dst, size_t dst_stride, hipStream_t st) {
  if (orientation < 1 || orientation > 8) return false;
  const int fx = orientation == 2 || orientation == 3 || orientation == 8 || orientation == 7;
  const int fy = orientation == 4 || orientation == 3 || orientation == 6 || orientation == 7;
  const int tr = orientation >= 5;
  const dim3 grid((xsize + 31) / 32, (ysize + 7) / 8);
#define JXLHIP_ORIENT(B)                                                                                      \
  case B:                                                                                                     \
    hipLaunchKernelGGL((k_orient<B>), grid, dim3(256), 0, st, (const unsigned char*)src, src_stride, (int)xsize, \
                       (int)ysize, (unsigned char*)dst, dst_stride, fx, fy, tr);                              \
    return true;
  switch (bytes_per_pixel) {
    JXLHIP_ORIENT(3)
    JXLHIP_ORIENT(4)
    JXLHIP_ORIENT(6)
    JXLHIP_ORIENT(8)
    JXLHIP_ORIENT(12)
    JXLHIP_ORIENT(16)
  }
#undef JXLHIP_ORIENT
  return false;
}

void LaunchDequantTables(float* table, const jxlhip_quant_encoding* enc_dev, int32_t* status,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_dequant_tables, dim3((JXLHIP_DEQUANT_TABLE_FLOATS + 255) / 256), dim3(256),
                     0, st, table, enc_dev, status);
}

void LaunchDequantDC(uint32_t xsb, uint32_t ysb, const int32_t* const q[3], float* const dc[3],
                     float* const tmp[3], const float mul_dc[3], float cfl_x, float cfl_b,
                     int smooth, const uint8_t* extra_precision, hipStream_t st) {
  const size_t n = (size_t)xsb * ysb;
  const bool do_smooth = smooth && xsb > 2 && ysb > 2;
  float* const* first = do_smooth ? tmp : dc;
  hipLaunchKernelGGL(k_dequant_dc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xsb, ysb, q[0],
                     q[1], q[2], first[0], first[1], first[2], mul_dc[0], mul_dc[1], mul_dc[2],
                     cfl_x, cfl_b, extra_precision);
  if (do_smooth) {
    DcPlanes p;
    for (int c = 0; c < 3; c++) {
      p.in[c] = tmp[c];
      p.out[c] = dc[c];
      p.mul[c] = mul_dc[c];
    }
    hipLaunchKernelGGL(k_smooth_dc, dim3((xsb + 63) / 64, (ysb + 3) / 4), dim3(256), 0, st, xsb,
                       ysb, p);
  }
}

// ---------------------------------------------------------------- sparse coefficient hand-off
// One workgroup per AC group: zero the group's dense block stream (3 x 65536 int16 = 384 KB), then scatter its
// non-zero coefficients.  What crosses PCIe is 4 bytes per NON-ZERO coefficient instead of 2 bytes per coefficient.
__global__ __launch_bounds__(1024) void k_expand_sparse(const uint8_t* __restrict__ sparse, const uint32_t* __restrict__ offsets,
                                                       int16_t* __restrict__ dense, uint32_t g0) {
  const uint32_t g = g0 + blockIdx.x;
  const uint32_t off = offsets[g];
  if (off == 0xFFFFFFFFu) return;  // handed over densely (progressive passes, a channel with too many non-zeros)
  const uint32_t* hdr = (const uint32_t*)(sparse + (size_t)off * 16u);
  const uint32_t n0 = hdr[0], n1 = hdr[1], n2 = hdr[2];
  int16_t* d = dense + (size_t)g * 3 * 65536;
  uint4* d4 = (uint4*)d;
  for (uint32_t i = threadIdx.x; i < 3u * 65536u * 2u / 16u; i += 1024) d4[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t* e = hdr + 4;
  const uint32_t total = n0 + n1 + n2;
  for (uint32_t i = threadIdx.x; i < total; i += 1024) {
    const uint32_t c = (i >= n0 ? 1u : 0u) + (i >= n0 + n1 ? 1u : 0u);
    const uint32_t v = e[i];
    d[c * 65536u + (v >> 16)] = (int16_t)(v & 0xffffu);
  }
}

// Zeroes n 32-bit words with a KERNEL (not hipMemsetAsync): the first node of a captured frame graph.  A memset node at
// the root of a hipGraph was observed (ROCm 7.2, MI355X) to start before the previous launch of the same graph on the
// same stream had finished -- it zeroed the work-list counters under the previous frame's transform kernel (memory
// fault); a kernel node keeps stream order.
__global__ __launch_bounds__(256) void k_zero_u32(uint32_t* __restrict__ p, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) p[i] = 0u;
}
void LaunchZeroU32(uint32_t* p, uint32_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_zero_u32, dim3((n + 255u) / 256u), dim3(256), 0, st, p, n);
}

// Compute units of the device the calling thread has current: what the launch geometry of the persistent /
// generation-filling kernels is sized from (256 on a whole MI355X, 32 on a CPX partition of it).
unsigned DeviceCus() {
  static std::atomic<unsigned> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  unsigned n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = (unsigned)v;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

void LaunchExpandSparse(const uint8_t* sparse, const uint32_t* offsets, int16_t* dense, uint32_t g0, uint32_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_expand_sparse, dim3(n), dim3(1024), 0, st, sparse, offsets, dense, g0);
}

}  // namespace jxlhip
