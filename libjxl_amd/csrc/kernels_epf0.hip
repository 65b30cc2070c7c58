// kernels_epf0.hip -- epf_iters = 3: [Gaborish] + EPF0 as a register row march of its own.
//
// Three EPF iterations (distances above ~3: dec_cache.cc:156-170 puts EPF0 in front of EPF1 and EPF2)
// reach 7 pixels to every side.  One march carrying all four stages wants more registers than a wave
// has (the EPF0 window alone: 6 Gaborish rows, six difference images, their plus-sums -- ~190 VGPRs),
// so the stage list is split where the data is smallest:
//     k_epf0          planes --[Gaborish]--EPF0--> second plane set (plain row-major: every store and every
//                     load of the next kernel covers 512 contiguous bytes per wave)
//     k_filters_fast  second plane set --EPF1--EPF2--XYB->RGB--> output   (the existing <GAB=0, EPF=2> march)
// The second kernel reads its input mirrored at the image edges like every first stage does, which is
// exactly the reference's border rule for the stage after EPF0 (simple_render_pipeline.cc:129-164).
//
// EPF0 (stage_epf.cc:54-193): 12 neighbours (the 5x5 "plus": |dy| + |dx| <= 2), each weighted by a
// 5-pixel plus-shaped SAD over the three channels.  As in the EPF1 march (kernels_filters_fast.hip) the
// sums are regrouped: with the six channel-summed difference images
//     V1(y,x) = S |p(y,x) - p(y-1,x)|     V2: (y-2,x)     H1: (y,x-1)     H2: (y,x-2)
//     A (y,x) = S |p(y,x) - p(y-1,x-1)|   B : (y-1,x+1)           S = sum_c scale_c
// and PS_D the plus-shaped sum of D, the twelve SADs of pixel (y,x) are
//     (-2,0) PS_V2(y,x)   (+2,0) PS_V2(y+2,x)   (-1,0) PS_V1(y,x)   (+1,0) PS_V1(y+1,x)
//     (0,-2) PS_H2(y,x)   (0,+2) PS_H2(y,x+2)   (0,-1) PS_H1(y,x)   (0,+1) PS_H1(y,x+1)
//     (-1,-1) PS_A(y,x)   (+1,+1) PS_A(y+1,x+1) (-1,+1) PS_B(y,x)   (+1,-1) PS_B(y+1,x-1)
// (|a - b| is symmetric: the SAD towards +o at x is the SAD towards -o at x + o): six plus-sums per
// pixel instead of twelve 15-term SADs -- the reference's terms in another association.
// ~230 VALU issues per row step and wave (pair of columns per lane, packed fp32).
#include <stdlib.h>

#include "epf0_march.h"

namespace jxlhip {

namespace {

template <int GAB>
struct Geom0 {
  static constexpr int HX = GAB + 3;
  static constexpr int HXP = 4;            // in whole column pairs
  static constexpr int USE = 128 - 2 * HXP;
};

template <int GAB, bool EDGE>
__device__ __forceinline__ void March0(const DevFrame& f, const FilterParams& P, Lane& L, int y_begin, int y_end,
                                       float* const (&dst)[3]) {
  constexpr int HX = Geom0<GAB>::HX;
  const int H = (int)f.ysize;
  const int r_first = y_begin - HX;
  const int r_last = y_end + HX - 1;
  const int plane_last = f.plane_y0 + (int)f.plane_tile_rows * 8 - 1;
  int prefetch_last_row = r_last;
  if (prefetch_last_row > plane_last && prefetch_last_row < H) prefetch_last_row = plane_last;
  State0 s;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool fetch = k < kAhead;
    int pr = r_first + k;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    const uint32_t off = RowOffset(f, Mirror1(pr, H));
    LaneOffset(L.byte_off);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (fetch) s.x[c][k] = LoadPair<EDGE>((const char*)f.xyb[c] + off, L);
      else s.x[c][k] = v2f{0.0f, 0.0f};
      s.g[c][k] = v2f{0.0f, 0.0f};
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int c = 0; c < 3; c++) s.hs[c][k] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int d = 0; d < kNumD; d++) s.ps[d][k] = v2f{0.0f, 0.0f};
  }
#pragma unroll
  for (int d = 0; d < kNumD; d++) s.dprev[d] = s.part[d] = v2f{0.0f, 0.0f};
  float inv_sigma_blk = -1.0f;
#define JXLHIP_STEP0(K) Step0<GAB, K, EDGE>(s, r + K, f, P, L, prefetch_last_row, y_begin, y_end, inv_sigma_blk, dst)
  for (int r = r_first; r <= r_last; r += 8) {
    JXLHIP_STEP0(0);
    JXLHIP_STEP0(1);
    JXLHIP_STEP0(2);
    JXLHIP_STEP0(3);
    if (r + 4 > r_last) break;
    JXLHIP_STEP0(4);
    JXLHIP_STEP0(5);
    JXLHIP_STEP0(6);
    JXLHIP_STEP0(7);
  }
#undef JXLHIP_STEP0
}

// rows [oy0, oy1) of the frame, every image column, into dst (planes of f's geometry)
template <int GAB>
__global__ __launch_bounds__(256, 2) void k_epf0(DevFrame f, FilterParams P, int RH, int oy0, int oy1, float* d0,
                                                 float* d1, float* d2) {
  using G = Geom0<GAB>;
  constexpr int HXP = G::HXP, USE = G::USE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * 4 + wave;
  const int W = (int)f.xsize;
  const int x_first = strip * USE;
  if (x_first >= W) return;
  const int y_begin = oy0 + blockIdx.y * RH;
  const int y_end = min(y_begin + RH, oy1);
  if (y_begin >= y_end) return;
  Lane L;
  L.gx = x_first - HXP + 2 * lane;
  const int m0 = MirrorF(L.gx, W), m1 = MirrorF(L.gx + 1, W);
  const int base = m0 & ~1;
  L.sel0 = m0 & 1;
  L.sel1 = m1 & 1;
  L.byte_off = ((uint32_t)(base >> 3) * 64u + (uint32_t)(base & 7)) * 4u;
  const bool edge = x_first - HXP < 0 || x_first - HXP + 128 > W;  // wave-uniform
  const bool lane_in = lane >= HXP / 2 && lane < 64 - HXP / 2;
  L.out0 = lane_in && L.gx < W;
  L.out1 = lane_in && L.gx + 1 < W;
  const int gxc = L.gx < 0 ? 0 : (L.gx >= W ? W - 1 : L.gx);
  L.sx4 = (uint32_t)(gxc >> 3) * 4u;
  L.out_off = (uint32_t)gxc * 4u;
  const int ix = gxc & 7;
  L.mul = v2f{ix == 0 ? P.bsm[0] : P.sm[0], ix == 6 ? P.bsm[0] : P.sm[0]};
  float* const dst[3] = {d0, d1, d2};
  if (edge) March0<GAB, true>(f, P, L, y_begin, y_end, dst);
  else March0<GAB, false>(f, P, L, y_begin, y_end, dst);
}

// see FilterRowsPerWave (kernels_filters_fast.hip); two workgroups per CU
int Epf0RowsPerWave(unsigned wgx, unsigned rows, int hx) {
  const unsigned resident = DeviceCus() * 2u;
  int best = 64;
  double best_cost = 1e30;
  for (int rh = 16; rh <= 512; rh++) {
    const unsigned wgs = wgx * ((rows + rh - 1) / rh);
    const unsigned gens = (wgs + resident - 1) / resident;
    const double cost = (double)gens * (rh + 2 * hx + 6);
    if (cost < best_cost) {
      best_cost = cost;
      best = rh;
    }
  }
  return best;
}

}  // namespace

// [Gaborish] + EPF0 for the rows the following EPF1 + EPF2 march of rows [f.fy0, f.fy1) reads
// (3 more on each side, inside the frame), planes f.xyb -> dst (row-major, f.tile_stride * 32 bytes per row,
// first row f.plane_y0).  false: geometry not covered.
bool LaunchEpf0(const DevFrame& f, const FilterParams& p, int gab, float* const dst[3], hipStream_t st) {
  if (f.xsize < 16 || f.ysize < 16) return false;
  if ((uint64_t)f.plane_tile_rows * f.tile_stride * 256u >= (1ull << 32)) return false;
  const int oy0 = (int)f.fy0 - 3 < 0 ? 0 : (int)f.fy0 - 3;
  const int oy1 = f.fy1 + 3 > f.ysize ? (int)f.ysize : (int)f.fy1 + 3;
  constexpr int USE = Geom0<1>::USE;
  const unsigned strips = (f.xsize + USE - 1) / USE;
  const unsigned wgx = (strips + 3) / 4;
  const int RH = Epf0RowsPerWave(wgx, oy1 - oy0, 3 + gab);
  const dim3 grid(wgx, (oy1 - oy0 + RH - 1) / RH);
  if (gab) hipLaunchKernelGGL((k_epf0<1>), grid, dim3(256), 0, st, f, p, RH, oy0, oy1, dst[0], dst[1], dst[2]);
  else hipLaunchKernelGGL((k_epf0<0>), grid, dim3(256), 0, st, f, p, RH, oy0, oy1, dst[0], dst[1], dst[2]);
  return true;
}

}  // namespace jxlhip
