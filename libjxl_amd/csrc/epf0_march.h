// epf0_march.h -- [Gaborish] + EPF0 as a register row march (stage_epf.cc:54-193): the per-row step shared by
// k_epf0 (kernels_epf0.hip: rows from the XYB planes) and k_fused_pc0 (kernels_fused.hip, part 3: rows from the LDS
// slab a producing wave fills, DCT8 cells decoded in the wave).  See kernels_epf0.hip for the regrouping of the
// reference's twelve 15-term SADs into six plus-sums per pixel.
#ifndef JXLHIP_EPF0_MARCH_H_
#define JXLHIP_EPF0_MARCH_H_

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
// SRC_LDS (k_fused_pc0): row r sits in slab row PH of the block row at image row slab_y0 and was requested one step ago;
// sigma_lds = the producer's inv_sigma of the lane's cell for that block row (the marching wave issues no vector load).
// PART_LDS (k_fused_pc0 with Gaborish): the first PART_LDS of the six running plus-sum parts live in LDS (part_lds: the
// lane's pair of entry 0; entry k is 128 floats on) instead of registers -- read once and written once per step, and the
// eight registers are what the kernel lacks to fit three waves per SIMD.
template <int GAB, int PH, bool EDGE, int SRC = SRC_PLANES, int PART_LDS = 0>
__device__ __forceinline__ void Step0(State0& s, int r, const DevFrame& f, const FilterParams& P, Lane& L,
                                      int prefetch_last_row, int y_begin, int y_end, float& inv_sigma_blk,
                                      float* const (&dst)[3], int slab_y0 = 0, float sigma_lds = 0.0f,
                                      float __attribute__((address_space(3)))* part_lds = nullptr) {
  constexpr int S0 = PH & 3, S1 = (PH + 3) & 3, S2 = (PH + 2) & 3;
  constexpr int X0 = PH & 7, X1 = (PH + 7) & 7, X2 = (PH + 6) & 7;
  const int H = (int)f.ysize;
  if constexpr (SRC == SRC_LDS) {
    if constexpr (PH < 7) {
      const int nrow = Mirror1(r + 1, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][(PH + 1) & 7] = LdsPair<EDGE>(L, c, nrow);
    }
  } else if constexpr (PH % kBurst == 0) {
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
    typedef v2f __attribute__((address_space(3))) * P2;
    const v2f pk = k < PART_LDS ? *(P2)(part_lds + k * 128) : s.part[k];
    s.ps[k][S0] = pk + dnew[k];
    v2f v = AddLeftS(s.dprev[k], dnew[k]);
    v = v + dnew[k];
    const v2f np = AddRightS(v, dnew[k]);
    if (k < PART_LDS) *(P2)(part_lds + k * 128) = np;
    else s.part[k] = np;
    s.dprev[k] = dnew[k];
  }
  // EPF0 output row o = q - 3: plus-sum rows o (slot S2), o + 1 (S1), o + 2 (S0)
  const int o = r - GAB - 3;
  const float kMinSigma = -3.90524291751269967465540850526868f;
  if ((o & 7) == 0 || o == y_begin) {
    float is;
    if constexpr (SRC == SRC_LDS) {
      is = sigma_lds;  // (output row o opens a block row exactly when its group's rows r = o + GAB + 3 are in the slab)
    } else {
      const int oc = o < 0 ? 0 : (o >= H ? H - 1 : o);
      is = *(const float*)((const char*)(f.inv_sigma + (size_t)(oc >> 3) * f.xsb) + LaneOffset(L.sx4));
    }
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

}  // namespace

}  // namespace jxlhip

#endif  // JXLHIP_EPF0_MARCH_H_
