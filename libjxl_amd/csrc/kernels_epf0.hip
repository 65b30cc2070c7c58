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

#include "filters_march.h"

namespace jxlhip {

namespace {

enum { kV1 = 0, kV2, kH1, kH2, kA, kB, kNumD };

struct State0 {
  v2f x[3][8];     // input rows (see State::x)
  v2f hs[3][4];    // GAB: left + right of the input rows
  v2f g[3][8];     // rows entering EPF0 (Gaborish output), slot = step & 7
  v2f dprev[kNumD];  // difference images of row q-1
  v2f part[kNumD];   // D(q-2) + D(q-1, x-1) + D(q-1, x) + D(q-1, x+1): plus-sums of row q-1 short of D(q)
  v2f ps[kNumD][4];  // plus-sums, slot = step & 3 of the step that completed them
};

// (p(x-2), p(x-1)) and (p(x+2), p(x+3)) of the column pair (x, x+1): the neighbouring lane's pair
__device__ __forceinline__ v2f PairFromLeft(v2f p) { return v2f{FromLeft(p.x), FromLeft(p.y)}; }
__device__ __forceinline__ v2f PairFromRight(v2f p) { return v2f{FromRight(p.x), FromRight(p.y)}; }
// a + w * (p(x-2), p(x-1)) / a + w * (p(x+2), p(x+3)); p old (see FmaLeftS)
__device__ __forceinline__ v2f FmaLeft2S(v2f w, v2f p, v2f a) {
  float ax = a.x, ay = a.y;
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ax) : "v"(p.x), "v"(w.x));
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ay) : "v"(p.y), "v"(w.y));
  return v2f{ax, ay};
}
__device__ __forceinline__ v2f FmaRight2S(v2f w, v2f p, v2f a) {
  float ax = a.x, ay = a.y;
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ax) : "v"(p.x), "v"(w.x));
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ay) : "v"(p.y), "v"(w.y));
  return v2f{ax, ay};
}

// One row step: input row r arrives, Gaborish row q = r - GAB is produced, the plus-sums of row q - 1
// are completed, EPF0 output row o = q - 3 leaves.  PH = step & 7.
template <int GAB, int PH, bool EDGE>
__device__ __forceinline__ void Step0(State0& s, int r, const DevFrame& f, const FilterParams& P, Lane& L,
                                      int prefetch_last_row, int y_begin, int y_end, float& inv_sigma_blk,
                                      float* const (&dst)[3]) {
  constexpr int S0 = PH & 3, S1 = (PH + 3) & 3, S2 = (PH + 2) & 3;
  constexpr int X0 = PH & 7, X1 = (PH + 7) & 7, X2 = (PH + 6) & 7;
  const int H = (int)f.ysize;
  if constexpr (PH % kBurst == 0) {
    LaneOffset(L.byte_off);
#pragma unroll
    for (int b = 0; b < kBurst; b++) {
      int pr = r + kAhead + b;
      pr = pr > prefetch_last_row ? prefetch_last_row : pr;
      const uint32_t off = RowOffset(f, Mirror1(pr, H));
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][(PH + kAhead + b) & 7] = LoadPair<EDGE>((const char*)f.xyb[c] + off, L);
    }
  }
  // Gaborish (stage_gaborish.cc:33-99) for row q
  v2f gq[3];
  if constexpr (GAB) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2f cur = s.x[c][X0];
      s.hs[c][S0] = v2f{Scalar(FromLeft(cur.y) + cur.y), Scalar(FromRight(cur.x) + cur.x)};
      const v2f sum1 = s.hs[c][S1] + (s.x[c][X2] + cur);
      const v2f sum2 = s.hs[c][S2] + s.hs[c][S0];
      gq[c] = Fma2(sum2, P.gab_w[c][2], Fma2(sum1, P.gab_w[c][1], s.x[c][X1] * P.gab_w[c][0]));
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) gq[c] = s.x[c][X0];
  }
  constexpr int G0 = PH & 7, G1 = (PH + 7) & 7, G2 = (PH + 6) & 7, G3 = (PH + 5) & 7, G4 = (PH + 4) & 7,
                G5 = (PH + 3) & 7;  // rows q, q-1, .. q-5
  // difference images of row q
  v2f dnew[kNumD];
  {
    v2f d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) d[c] = gq[c] - s.g[c][G1];
    dnew[kV1] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) d[c] = gq[c] - s.g[c][G2];
    dnew[kV2] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) d[c] = v2f{Scalar(FromLeft(gq[c].y) - gq[c].x), Scalar(gq[c].x - gq[c].y)};
    dnew[kH1] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) d[c] = v2f{Scalar(FromLeft(gq[c].x) - gq[c].x), Scalar(FromLeft(gq[c].y) - gq[c].y)};
    dnew[kH2] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2f up = s.g[c][G1];
      d[c] = v2f{Scalar(FromLeft(up.y) - gq[c].x), Scalar(up.x - gq[c].y)};
    }
    dnew[kA] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2f up = s.g[c][G1];
      d[c] = v2f{Scalar(up.y - gq[c].x), Scalar(FromRight(up.x) - gq[c].y)};
    }
    dnew[kB] = AbsScaleSum(d, P);
#pragma unroll
    for (int c = 0; c < 3; c++) s.g[c][G0] = gq[c];
  }
  // plus-sums: row q-1 is complete with D(q); row q's start with D(q-1) and its own three columns
#pragma unroll
  for (int k = 0; k < kNumD; k++) {
    s.ps[k][S0] = s.part[k] + dnew[k];
    v2f v = AddLeftS(s.dprev[k], dnew[k]);
    v = v + dnew[k];
    s.part[k] = AddRightS(v, dnew[k]);
    s.dprev[k] = dnew[k];
  }
  // EPF0 output row o = q - 3: plus-sum rows o (slot S2), o + 1 (S1), o + 2 (S0)
  const int o = r - GAB - 3;
  const float kMinSigma = -3.90524291751269967465540850526868f;
  if ((o & 7) == 0 || o == y_begin) {
    const int oc = o < 0 ? 0 : (o >= H ? H - 1 : o);
    const float is = *(const float*)((const char*)(f.inv_sigma + (size_t)(oc >> 3) * f.xsb) + LaneOffset(L.sx4));
    inv_sigma_blk = is < kMinSigma ? -__builtin_inff() : is;  // below the threshold the stage copies (stage_epf.cc:118-125)
  }
  const int iy = o & 7;
  const v2f mul = (iy == 0 || iy == 7) ? v2f{P.bsm[0], P.bsm[0]} : L.mul;
  const v2f inv_sigma = mul * inv_sigma_blk;
  // the reference's neighbour order (sads_off, stage_epf.cc:131-134)
  const v2f hh1 = s.ps[kH1][S2], hh2 = s.ps[kH2][S2], a1 = s.ps[kA][S1], b1 = s.ps[kB][S1];
  const v2f w0 = EpfW(s.ps[kV2][S2], inv_sigma);                               // (-2, 0)
  const v2f w1 = EpfW(s.ps[kA][S2], inv_sigma);                                // (-1,-1)
  const v2f w2 = EpfW(s.ps[kV1][S2], inv_sigma);                               // (-1, 0)
  const v2f w3 = EpfW(s.ps[kB][S2], inv_sigma);                                // (-1,+1)
  const v2f w4 = EpfW(hh2, inv_sigma);                                         // ( 0,-2)
  const v2f w5 = EpfW(hh1, inv_sigma);                                         // ( 0,-1)
  const v2f w6 = EpfW(v2f{hh1.y, FromRight(hh1.x)}, inv_sigma);                // ( 0,+1)
  const v2f w7 = EpfW(PairFromRight(hh2), inv_sigma);                          // ( 0,+2)
  const v2f w8 = EpfW(v2f{FromLeft(b1.y), b1.x}, inv_sigma);                   // (+1,-1): PS_B(o+1, x-1)
  const v2f w9 = EpfW(s.ps[kV1][S1], inv_sigma);                               // (+1, 0)
  const v2f wA = EpfW(v2f{a1.y, FromRight(a1.x)}, inv_sigma);                  // (+1,+1): PS_A(o+1, x+1)
  const v2f wB = EpfW(s.ps[kV2][S0], inv_sigma);                               // (+2, 0)
  v2f wsum = v2f{1.0f, 1.0f} + w0;
  wsum = wsum + w1;
  wsum = wsum + w2;
  wsum = wsum + w3;
  wsum = wsum + w4;
  wsum = wsum + w5;
  wsum = wsum + w6;
  wsum = wsum + w7;
  wsum = wsum + w8;
  wsum = wsum + w9;
  wsum = wsum + wA;
  wsum = wsum + wB;
  const v2f inv_w = {__builtin_amdgcn_rcpf(wsum.x), __builtin_amdgcn_rcpf(wsum.y)};
  v2f outv[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const v2f up2 = s.g[c][G5], up1 = s.g[c][G4], ctr = s.g[c][G3], dn1 = s.g[c][G2], dn2 = s.g[c][G1];
    v2f a = Fma2(w0, up2, ctr);
    a = FmaLeftS(w1, up1, a);
    a = Fma2(w2, up1, a);
    a = FmaRightS(w3, up1, a);
    a = FmaLeft2S(w4, ctr, a);
    a = FmaLeftS(w5, ctr, a);
    a = FmaRightS(w6, ctr, a);
    a = FmaRight2S(w7, ctr, a);
    a = FmaLeftS(w8, dn1, a);
    a = Fma2(w9, dn1, a);
    a = FmaRightS(wA, dn1, a);
    a = Fma2(wB, dn2, a);
    outv[c] = a * inv_w;
  }
  if (o >= y_begin && o < y_end) {
    const uint32_t off = (uint32_t)(o - f.plane_y0) * (f.tile_stride * 32u);  // row-major: SRC_LINEAR
    LaneOffset(L.out_off);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float* d = (float*)((char*)dst[c] + off + L.out_off);
      if (EDGE ? (L.out0 && L.out1) : L.out0) *(v2f*)d = outv[c];
      else if (!EDGE) {
      } else if (L.out0) d[0] = outv[c].x;
      else if (L.out1) d[1] = outv[c].y;
    }
  }
}

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
  const unsigned resident = 256u * 2u;
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
