// kernels_filters_fast.hip -- phase 2 for the stage lists with at most one EPF
// pass ([Gaborish] [EPF1] XYB->RGB: every stream below distance 1.5, i.e. the
// BASELINE d1.0 configuration), written for the CDNA4 wavefront instead of LDS:
//
//   * one wave = 64 adjacent pixel COLUMNS, marching down the rows of its band;
//   * horizontal neighbours come from the neighbouring LANE through DPP
//     wave_shr/wave_shl (full VALU rate, no LDS, no barrier);
//   * vertical neighbours come from a sliding window of rows kept in registers
//     (4-slot rings, slot = row & 3, resolved at compile time by unrolling the
//     row loop 4x);
//   * input rows are prefetched 4 rows ahead (one coalesced 256-byte load per
//     wave, row and channel), output rows leave as 12-byte RGB stores.
//
// EPF1 (lib/jxl/render_pipeline/stage_epf.cc:225-367) is evaluated through an
// exact regrouping of the reference's sums: with Du(x,y) = |p(x,y-1) - p(x,y)|
// and Dl(x,y) = |p(x-1,y) - p(x,y)|, the four SADs of pixel (x,y) are the
// plus-shaped sums  PV(x,y), PH(x,y), PH(x+1,y), PV(x,y+1)  of Du / Dl -- same
// terms, same order, same rounding as the reference (N, W, E, S), but each
// plus-sum is computed once per pixel instead of four times.
//
// Border rule (simple_render_pipeline.cc:129-164): stages read their input
// mirrored at the true image edge.  Gaborish of the mirrored input IS the
// mirrored Gaborish output (symmetric kernel, commutative pair sums), so halo
// lanes/rows outside the image simply run on mirrored input; this kernel is
// only used when no stage follows an EPF stage, where that identity is all
// that is needed.  Other stage lists use the generic kernel (kernels_filters.hip).
#include "dev_common.h"
#include "kernels.h"

namespace jxlhip {

namespace {

__device__ __forceinline__ int MirrorF(int x, int n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}

// row part of a block-major plane offset (the lane adds its tile column)
__device__ __forceinline__ size_t RowOffset(const DevFrame& f, int y) {
  const uint32_t ry = (uint32_t)(y - f.plane_y0);
  return (size_t)(ry >> 3) * f.tile_stride * 64u + ((ry & 7u) << 3);
}

// value of the lane holding column x-1 / x+1
__device__ __forceinline__ float FromLeft(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float FromRight(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

__device__ __forceinline__ float EpfW(float sad, float inv_sigma) {
  const float v = __builtin_fmaf(sad, inv_sigma, 1.0f);
  return v < 0.0f ? 0.0f : v;
}

template <int GAB, int EPF>
struct State {
  float pre[3][4];  // prefetched input rows
  float in[3][4];   // GAB: input rows
  float hs[3][4];   // GAB: left + right of the input rows
  float g[3][4];    // EPF: rows entering EPF (Gaborish output or input)
  float gl[3][4], gr[3][4];
  float du[3][4], dl[3][4];
  float pv[4], ph[4];
};

template <int OUTK>
__device__ __forceinline__ void Emit(const float* v, int gx, int gy_rel, const FilterParams& P) {
  if constexpr (OUTK == JXLHIP_OUT_LINEAR_RGB_F32) {
    float gr = v[1] + v[0], gg = v[1] - v[0], gb = v[2];
    gr = gr - P.cbrt_bias[0];
    gg = gg - P.cbrt_bias[1];
    gb = gb - P.cbrt_bias[2];
    const float mr = __builtin_fmaf(gr * gr, gr, P.opsin_bias[0]);
    const float mg = __builtin_fmaf(gg * gg, gg, P.opsin_bias[1]);
    const float mb = __builtin_fmaf(gb * gb, gb, P.opsin_bias[2]);
    const float* m = P.minv;
    float* dst = (float*)((char*)P.out + (size_t)gy_rel * P.out_stride) + 3 * (size_t)gx;
    dst[0] = __builtin_fmaf(m[2], mb, __builtin_fmaf(m[1], mg, m[0] * mr));
    dst[1] = __builtin_fmaf(m[5], mb, __builtin_fmaf(m[4], mg, m[3] * mr));
    dst[2] = __builtin_fmaf(m[8], mb, __builtin_fmaf(m[7], mg, m[6] * mr));
  } else {
    float* dst = (float*)P.out + (size_t)gy_rel * P.out_stride + gx;
    dst[0] = v[0];
    dst[P.out_plane_stride] = v[1];
    dst[2 * P.out_plane_stride] = v[2];
  }
}

// One row step.  PH = (r - r_first) & 3 is the ring slot of input row r.
// Row bookkeeping: q = row leaving Gaborish (r-1 with GAB, r without),
// p = q-1 = row whose plus-sums are completed, o = q-2 = EPF output row.
template <int GAB, int EPF, int OUTK, int PH>
__device__ __forceinline__ void Step(State<GAB, EPF>& s, int r, const DevFrame& f,
                                     const FilterParams& P, const float* const* col,
                                     int prefetch_last_row, int y_begin, int y_end, int gx,
                                     bool lane_out, float lane_mul, float& inv_sigma_blk) {
  constexpr int S0 = PH & 3, S1 = (PH + 3) & 3, S2 = (PH + 2) & 3, S3 = (PH + 1) & 3;  // r, r-1, r-2, r-3
  const int H = (int)f.ysize;
  float cur[3];
  // 1. take row r from the prefetch ring, refill the slot with row r+4
  {
    int pr = r + 4;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    const size_t off = RowOffset(f, MirrorF(pr, H));
#pragma unroll
    for (int c = 0; c < 3; c++) {
      cur[c] = s.pre[c][S0];
      s.pre[c][S0] = col[c][off];
    }
  }
  // 2. Gaborish (stage_gaborish.cc:33-99) for row q = r-1
  float gq[3];
  if constexpr (GAB) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.in[c][S0] = cur[c];
      s.hs[c][S0] = FromLeft(cur[c]) + FromRight(cur[c]);
      const float sum1 = s.hs[c][S1] + (s.in[c][S2] + s.in[c][S0]);
      const float sum2 = s.hs[c][S2] + s.hs[c][S0];
      gq[c] = __builtin_fmaf(sum2, P.gab_w[c][2],
                             __builtin_fmaf(sum1, P.gab_w[c][1], s.in[c][S1] * P.gab_w[c][0]));
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) gq[c] = cur[c];
  }
  constexpr int Q0 = GAB ? S1 : S0;  // slot of row q
  constexpr int Q1 = (Q0 + 3) & 3, Q2 = (Q0 + 2) & 3, Q3 = (Q0 + 1) & 3;  // q-1, q-2, q-3
  const int q = GAB ? r - 1 : r;
  float outv[3];
  int o;
  if constexpr (EPF) {
    // 3a. differences of the new row q
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float l = FromLeft(gq[c]), rr = FromRight(gq[c]);
      s.g[c][Q0] = gq[c];
      s.gl[c][Q0] = l;
      s.gr[c][Q0] = rr;
      s.du[c][Q0] = __builtin_fabsf(s.g[c][Q1] - gq[c]);
      s.dl[c][Q0] = __builtin_fabsf(l - gq[c]);
    }
    // 3b. plus-sums of row p = q-1 (order: up, left, centre, right, down)
    float pv = 0.0f, ph = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float du_c = s.du[c][Q1], dl_c = s.dl[c][Q1];
      float v = s.du[c][Q2] + FromLeft(du_c);
      v = v + du_c;
      v = v + FromRight(du_c);
      v = v + s.du[c][Q0];
      float h = s.dl[c][Q2] + FromLeft(dl_c);
      h = h + dl_c;
      h = h + __builtin_fabsf(s.g[c][Q1] - s.gr[c][Q1]);
      h = h + s.dl[c][Q0];
      pv = __builtin_fmaf(v, P.ch_scale[c], pv);
      ph = __builtin_fmaf(h, P.ch_scale[c], ph);
    }
    s.pv[Q1] = pv;
    s.ph[Q1] = ph;
    // 3c. EPF1 output row o = q-2
    o = q - 2;
    const float kMinSigma = -3.90524291751269967465540850526868f;
    if ((o & 7) == 0 || o == y_begin) {
      const int oc = o < 0 ? 0 : (o >= H ? H - 1 : o);
      inv_sigma_blk = f.inv_sigma[(size_t)(oc >> 3) * f.xsb + (gx >> 3)];
    }
    const int iy = o & 7;
    const float mul = (iy == 0 || iy == 7) ? P.bsm[1] : lane_mul;
    const float inv_sigma = inv_sigma_blk * mul;
    const float wN = EpfW(s.pv[Q2], inv_sigma);
    const float wW = EpfW(s.ph[Q2], inv_sigma);
    const float wE = EpfW(FromRight(s.ph[Q2]), inv_sigma);
    const float wS = EpfW(s.pv[Q1], inv_sigma);
    float wsum = 1.0f + wN;
    wsum = wsum + wW;
    wsum = wsum + wE;
    wsum = wsum + wS;
    const float inv_w = __builtin_amdgcn_rcpf(wsum);
    const bool skip = inv_sigma_blk < kMinSigma;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float a = s.g[c][Q2];
      a = __builtin_fmaf(wN, s.g[c][Q3], a);
      a = __builtin_fmaf(wW, s.gl[c][Q2], a);
      a = __builtin_fmaf(wE, s.gr[c][Q2], a);
      a = __builtin_fmaf(wS, s.g[c][Q1], a);
      outv[c] = skip ? s.g[c][Q2] : a * inv_w;
    }
  } else {
    o = q;
#pragma unroll
    for (int c = 0; c < 3; c++) outv[c] = gq[c];
  }
  // 4. emit
  if (o >= y_begin && o < y_end && !lane_out) Emit<OUTK>(outv, gx, o - (int)f.y0, P);
}

template <int GAB, int EPF, int OUTK, int RH>
__global__ __launch_bounds__(256) void k_filters_fast(DevFrame f, FilterParams P) {
  constexpr int HX = GAB + 2 * EPF;  // halo columns/rows on each side
  constexpr int USE = 64 - 2 * HX;   // output columns per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * 4 + wave;
  const int W = (int)f.xsize, H = (int)f.ysize;
  const int x_first = strip * USE;  // first output column of the wave
  if (x_first >= W) return;
  const int gx = x_first - HX + lane;
  const int y_begin = (int)f.fy0 + blockIdx.y * RH;
  const int y_end = min(y_begin + RH, (int)f.fy1);
  if (y_begin >= y_end) return;
  const bool lane_out = lane < HX || lane >= 64 - HX || gx >= W;
  const int mx = MirrorF(gx, W);
  const int gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
  // block-major planes: lane part (tile column, column in tile) + row part
  const size_t lane_off = (size_t)(mx >> 3) * 64 + (mx & 7);
  const float* col[3] = {f.xyb[0] + lane_off, f.xyb[1] + lane_off, f.xyb[2] + lane_off};
  const int ix = gxc & 7;
  const float lane_mul = (ix == 0 || ix == 7) ? P.bsm[1] : P.sm[1];
  // rows: input rows r = y_begin - HX .. y_end + HX - 1; the pipeline emits
  // row r - HX at step r.
  const int r_first = y_begin - HX;
  const int r_last = y_end + HX - 1;
  // the prefetcher may run up to 4 rows ahead: clamp to the last row this
  // context holds (plane rows cover [y0 - halo, y1_padded + halo))
  const int plane_last = f.plane_y0 + (int)f.plane_tile_rows * 8 - 1;
  int prefetch_last_row = r_last;
  // mirrored rows always fall inside the plane; direct rows must too
  if (prefetch_last_row > plane_last && prefetch_last_row < H) prefetch_last_row = plane_last;
  State<GAB, EPF> s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int pr = r_first + k;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    const size_t off = RowOffset(f, MirrorF(pr, H));
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.pre[c][k] = col[c][off];
      s.in[c][k] = 0.0f;
      s.hs[c][k] = 0.0f;
      s.g[c][k] = 0.0f;
      s.gl[c][k] = 0.0f;
      s.gr[c][k] = 0.0f;
      s.du[c][k] = 0.0f;
      s.dl[c][k] = 0.0f;
    }
    s.pv[k] = 0.0f;
    s.ph[k] = 0.0f;
  }
  float inv_sigma_blk = -1.0f;
  const float* const* cp = col;
  for (int r = r_first; r <= r_last; r += 4) {
    Step<GAB, EPF, OUTK, 0>(s, r, f, P, cp, prefetch_last_row, y_begin, y_end, gxc, lane_out,
                            lane_mul, inv_sigma_blk);
    Step<GAB, EPF, OUTK, 1>(s, r + 1, f, P, cp, prefetch_last_row, y_begin, y_end, gxc, lane_out,
                            lane_mul, inv_sigma_blk);
    Step<GAB, EPF, OUTK, 2>(s, r + 2, f, P, cp, prefetch_last_row, y_begin, y_end, gxc, lane_out,
                            lane_mul, inv_sigma_blk);
    Step<GAB, EPF, OUTK, 3>(s, r + 3, f, P, cp, prefetch_last_row, y_begin, y_end, gxc, lane_out,
                            lane_mul, inv_sigma_blk);
  }
}

template <int GAB, int EPF, int OUTK>
void LaunchFastT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  constexpr int RH = 64;
  constexpr int USE = 64 - 2 * (GAB + 2 * EPF);
  const unsigned strips = (f.xsize + USE - 1) / USE;
  const dim3 grid((strips + 3) / 4, (f.fy1 - f.fy0 + RH - 1) / RH);
  hipLaunchKernelGGL((k_filters_fast<GAB, EPF, OUTK, RH>), grid, dim3(256), 0, st, f, p);
}

}  // namespace

bool LaunchFiltersFast(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                       int output_kind, hipStream_t st) {
  if (epf_iters > 1 || (gab == 0 && epf_iters == 0)) return false;
#define JXLHIP_FAST(G, E)                                  \
  if (gab == G && epf_iters == E) {                        \
    if (output_kind == 0) LaunchFastT<G, E, 0>(f, p, st);  \
    else LaunchFastT<G, E, 1>(f, p, st);                   \
    return true;                                           \
  }
  JXLHIP_FAST(1, 0)
  JXLHIP_FAST(0, 1)
  JXLHIP_FAST(1, 1)
#undef JXLHIP_FAST
  return false;
}

}  // namespace jxlhip
