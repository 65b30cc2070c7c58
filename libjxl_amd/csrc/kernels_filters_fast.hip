// kernels_filters_fast.hip -- phase 2 for the stage lists with at most one EPF
// pass ([Gaborish] [EPF1] XYB->RGB: every stream below distance 1.5, i.e. the
// BASELINE d1.0 configuration), written for the CDNA4 wavefront instead of LDS:
//
//   * one wave = 128 adjacent pixel COLUMNS (each lane owns an aligned PAIR of
//     columns), marching down the rows of its band.  Two pixels per lane turn
//     the filter arithmetic into packed fp32 (v_pk_add/mul/fma_f32: two results
//     per VALU issue) and halve the cross-lane traffic per pixel;
//   * horizontal neighbours inside the pair are free, the two outside it come
//     from the neighbouring LANE through DPP wave_shr/wave_shl (no LDS, no
//     barrier);
//   * vertical neighbours come from a sliding window of rows kept in registers
//     (4-slot rings, slot = row & 3, resolved at compile time by unrolling the
//     row loop 4x);
//   * input rows are prefetched 4 rows ahead (one 8-byte load per lane, row
//     and channel; the row base is scalar), output rows leave as 24-byte
//     non-temporal RGB stores.
//
// Measured on MI355X (8K d1.0, JXLHIP_DEBUG ablations): arithmetic alone 135 us,
// + plane reads 140 us, + output stores 210 us = 4.0 TB/s of HBM traffic, where a
// plain device copy reaches 5.0-5.4 TB/s (read + write).  Tried and measured
// without gain: 8-row prefetch (-8 %), 3 waves per SIMD, and routing the RGB row
// through LDS so that every store instruction writes whole 64-byte lines (same
// time: the kernel is bound by mixed read/write HBM traffic, not by the number
// of write requests).
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
#include <stdlib.h>

#include "dev_common.h"
#include "emit.h"
#include "kernels.h"

namespace jxlhip {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int MirrorF(int x, int n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}

// row part of a block-major plane offset (the lane adds its tile column)
__device__ __forceinline__ size_t RowOffset(const DevFrame& f, int y) {
  const uint32_t ry = (uint32_t)(y - f.plane_y0);
  return (size_t)(ry >> 3) * f.tile_stride * 64u + ((ry & 7u) << 3);
}

// value held by the previous / next lane (0 at the wave's ends: those lanes
// are halo lanes whose results are never stored)
__device__ __forceinline__ float FromLeft(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float FromRight(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
// For the column pair p = (x, x+1), the left neighbours are (x-1, x) and the
// right neighbours (x+1, x+2): one of each comes from the adjacent lane.  The
// helpers keep those operations scalar so that the DPP read folds into the
// arithmetic instruction (v_add_f32_dpp, v_sub_f32_dpp, v_fmac_f32_dpp);
// everything that stays inside the lane is written on pairs and becomes packed
// fp32.
__device__ __forceinline__ v2f AddLeft(v2f acc, v2f p) {
  return v2f{acc.x + FromLeft(p.y), acc.y + p.x};
}
__device__ __forceinline__ v2f AddRight(v2f acc, v2f p) {
  return v2f{acc.x + p.y, acc.y + FromRight(p.x)};
}

// |v| materialised by an instruction the optimiser cannot see through: as an
// fabs it would be folded into its consumers as a source modifier, which forces
// them into the VOP3 encoding -- no DPP operand, no packed form.
__device__ __forceinline__ float AbsOpaque(float v) {
  float r;
  asm("v_and_b32 %0, 0x7fffffff, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ v2f Abs2(v2f v) { return v2f{AbsOpaque(v.x), AbsOpaque(v.y)}; }
__device__ __forceinline__ v2f Fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f Fma2(v2f a, float b, v2f c) {
  return __builtin_elementwise_fma(a, v2f{b, b}, c);
}

__device__ __forceinline__ v2f FmaLeft(v2f w, v2f p, v2f a) {
  return v2f{__builtin_fmaf(w.x, FromLeft(p.y), a.x), __builtin_fmaf(w.y, p.x, a.y)};
}
__device__ __forceinline__ v2f FmaRight(v2f w, v2f p, v2f a) {
  return v2f{__builtin_fmaf(w.x, p.y, a.x), __builtin_fmaf(w.y, FromRight(p.x), a.y)};
}

// max(0, 1 + sad * inv_sigma) (stage_epf.cc:62-71).  maxNum semantics on
// purpose: a block whose sigma is below the filter threshold carries
// inv_sigma = -inf, which turns every weight into 0 (also 0 * -inf = NaN).
__device__ __forceinline__ v2f EpfW(v2f sad, v2f inv_sigma) {
  const v2f v = Fma2(sad, inv_sigma, v2f{1.0f, 1.0f});
  return v2f{__builtin_fmaxf(v.x, 0.0f), __builtin_fmaxf(v.y, 0.0f)};
}

struct State {
  v2f pre[3][4];  // prefetched input rows
  v2f in[3][4];   // GAB: input rows
  v2f hs[3][4];   // GAB: left + right of the input rows
  v2f g[3][4];    // EPF: rows entering EPF (Gaborish output or input)
  v2f du[3][4], dl[3][4];
  v2f pv[4], ph[4];
  v2f e[3][4];    // EPF == 2: EPF1 output rows entering EPF2
  v2f dv[4];      // EPF == 2: channel-weighted |row - row above| of the e rows
};

// per-lane constants
struct Lane {
  uint32_t byte_off;  // byte offset of the lane's aligned column pair inside a plane row
  bool sel0, sel1;    // edge waves: which half of the loaded pair each column takes
  bool edge;          // wave-uniform: the strip touches a mirrored image edge
  int gx;             // first column of the pair (may lie outside the image)
  bool out0, out1;    // column is written by this wave
  v2f mul;            // EPF sigma multiplier of the two columns (border columns of an 8x8 block differ)
  v2f mul2;           // ... of the third EPF stage
  // EPF == 2, edge waves: the one out-of-image column EPF2 reads takes its mirror (= the
  // edge column): pair (-2,-1): .y <- column 0; pair (W, W+1): .x <- column W-1 (W even);
  // pair (W-1, W): .y <- .x (W odd)
  bool fix_left, fix_right_even, fix_right_odd;
  int sx;             // block column of the pair for the sigma look-up (clamped)
  // packed 8-bit output: the dither pattern, staged in LDS (a global load per
  // sample would queue behind the row prefetch in the in-order vmcnt)
  const float __attribute__((address_space(3))) * dither;
};

__device__ __forceinline__ v2f LoadPair(const float* rowp, const Lane& L) {
  const v2f v = *(const v2f*)((const char*)rowp + L.byte_off);
  if (!L.edge) return v;
  return v2f{L.sel0 ? v.y : v.x, L.sel1 ? v.y : v.x};
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// The output is written once and never read by this pipeline: streaming
// (non-temporal) stores keep it from displacing the XYB planes in L2 / MALL.
// 8-bit RGB: a lane's two pixels are 6 bytes, and sub-dword stores are slow
// (measured: three 16-bit stores per lane tripled the frame time).  Two
// neighbouring lanes own 12 bytes = 3 dwords starting at a multiple of 4
// columns: the first lane stores dwords 0-1 (borrowing 2 bytes from its right
// neighbour through DPP), the second lane dword 2.
template <typename Sel>
__device__ __forceinline__ void StoreRgb8Pair(const FilterParams& P, const Lane& L, char* row,
                                              int gy, const float* a, const float* b) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(4)));
  uint32_t qa[4], qb[4];
  PackSamples<Sel>(P, L.dither, L.gx, gy, a, qa);
  PackSamples<Sel>(P, L.dither, L.gx + 1, gy, b, qb);
  const uint32_t lo = qa[0] | (qa[1] << 8) | (qa[2] << 16) | (qb[0] << 24);  // bytes 0..3
  const uint32_t hi = qb[1] | (qb[2] << 8);                                  // bytes 4..5
  const uint32_t full = (L.out0 && L.out1) ? 1u : 0u;
  const uint32_t r_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xf, 0xf, true);
  const uint32_t r_full = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)full, 0x130, 0xf, 0xf, true);
  const uint32_t l_full = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)full, 0x138, 0xf, 0xf, true);
  const bool first = ((L.gx >> 1) & 1) == 0;  // gx multiple of 4
  uint8_t* d = (uint8_t*)row + (size_t)L.gx * 3;
  if (first && full && r_full) {
    __builtin_nontemporal_store(u2{lo, hi | (r_lo << 16)}, (u2*)d);
  } else if (!first && full && l_full) {
    __builtin_nontemporal_store((lo >> 16) | (hi << 16), (uint32_t*)(d + 2));
  } else {
    if (L.out0) {
      d[0] = (uint8_t)qa[0];
      d[1] = (uint8_t)qa[1];
      d[2] = (uint8_t)qa[2];
    }
    if (L.out1) {
      d[3] = (uint8_t)qb[0];
      d[4] = (uint8_t)qb[1];
      d[5] = (uint8_t)qb[2];
    }
  }
}

template <int OUTK, int FMT>
__device__ __forceinline__ void EmitPair(const v2f* v, const Lane& L, int gy, const DevFrame& f,
                                         const FilterParams& P) {
  const int gy_rel = gy - (int)f.y0;
  if constexpr (OUTK == JXLHIP_OUT_PACKED) {
    // FromLinearStage + WriteToOutputStage (emit.h); the packed formats move
    // 3..16 bytes per pixel, a fraction of the float output
    using Sel = FmtSel<FMT>;
    char* row = (char*)P.out + (size_t)gy_rel * P.out_stride;
    float a[3], b[3];
    XybToRgb(v[0].x, v[1].x, v[2].x, P, a);
    XybToRgb(v[0].y, v[1].y, v[2].y, P, b);
    if (Sel::sample_type(P.fmt) == JXLHIP_SAMPLE_U8 && Sel::channels(P.fmt) == 3) {
      StoreRgb8Pair<Sel>(P, L, row, gy, a, b);  // all lanes: uses DPP
    } else if (L.out0 && L.out1) {
      StorePackedPair<Sel>(P, L.dither, row, L.gx, gy, a, b);
    } else if (L.out0) {
      StorePackedPixel<Sel>(P, L.dither, row, L.gx, gy, a);
    } else if (L.out1) {
      StorePackedPixel<Sel>(P, L.dither, row, L.gx + 1, gy, b);
    }
  } else if constexpr (OUTK == JXLHIP_OUT_LINEAR_RGB_F32) {
    float* dst = (float*)((char*)P.out + (size_t)gy_rel * P.out_stride) + 3 * (size_t)L.gx;
    float a[3], b[3];
    XybToRgb(v[0].x, v[1].x, v[2].x, P, a);
    XybToRgb(v[0].y, v[1].y, v[2].y, P, b);
    if (L.out0 && L.out1) {  // 24 contiguous bytes
      __builtin_nontemporal_store(f4u{a[0], a[1], a[2], b[0]}, (f4u*)dst);
      __builtin_nontemporal_store(f2u{b[1], b[2]}, (f2u*)(dst + 4));
    } else if (L.out0) {
      __builtin_nontemporal_store(a[0], dst);
      __builtin_nontemporal_store(a[1], dst + 1);
      __builtin_nontemporal_store(a[2], dst + 2);
    } else if (L.out1) {
      __builtin_nontemporal_store(b[0], dst + 3);
      __builtin_nontemporal_store(b[1], dst + 4);
      __builtin_nontemporal_store(b[2], dst + 5);
    }
  } else {
    float* dst = (float*)P.out + (size_t)gy_rel * P.out_stride + L.gx;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float* d = dst + c * P.out_plane_stride;
      if (L.out0 && L.out1) {
        __builtin_nontemporal_store(f2u{v[c].x, v[c].y}, (f2u*)d);
      } else if (L.out0) {
        __builtin_nontemporal_store(v[c].x, d);
      } else if (L.out1) {
        __builtin_nontemporal_store(v[c].y, d + 1);
      }
    }
  }
}

// One row step.  PH = (r - r_first) & 3 is the ring slot of input row r.
// Row bookkeeping: q = row leaving Gaborish (r-1 with GAB, r without),
// p = q-1 = row whose plus-sums are completed, o = q-2 = EPF output row.
template <int GAB, int EPF, int OUTK, int FMT, int PH>
__device__ __forceinline__ void Step(State& s, int r, const DevFrame& f, const FilterParams& P,
                                     const Lane& L, int prefetch_last_row, int y_begin, int y_end,
                                     float& inv_sigma_blk, float& inv_sigma_blk2) {
  constexpr int S0 = PH & 3, S1 = (PH + 3) & 3, S2 = (PH + 2) & 3;  // r, r-1, r-2
  const int H = (int)f.ysize;
  v2f cur[3];
  // 1. take row r from the prefetch ring, refill the slot with row r+4
  {
    int pr = r + 4;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    if (f.debug & 8) pr = y_begin + (pr & 7);  // ablation: reads stay in L1
    const size_t off = RowOffset(f, MirrorF(pr, H));
#pragma unroll
    for (int c = 0; c < 3; c++) {
      cur[c] = s.pre[c][S0];
      s.pre[c][S0] = LoadPair(f.xyb[c] + off, L);
    }
  }
  // 2. Gaborish (stage_gaborish.cc:33-99) for row q = r-1
  v2f gq[3];
  if constexpr (GAB) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.in[c][S0] = cur[c];
      s.hs[c][S0] = v2f{FromLeft(cur[c].y) + cur[c].y, cur[c].x + FromRight(cur[c].x)};
      const v2f sum1 = s.hs[c][S1] + (s.in[c][S2] + s.in[c][S0]);
      const v2f sum2 = s.hs[c][S2] + s.hs[c][S0];
      gq[c] = Fma2(sum2, P.gab_w[c][2], Fma2(sum1, P.gab_w[c][1], s.in[c][S1] * P.gab_w[c][0]));
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) gq[c] = cur[c];
  }
  constexpr int Q0 = GAB ? S1 : S0;  // slot of row q
  constexpr int Q1 = (Q0 + 3) & 3, Q2 = (Q0 + 2) & 3, Q3 = (Q0 + 1) & 3;  // q-1, q-2, q-3
  const int q = GAB ? r - 1 : r;
  v2f outv[3];
  int o;
  if constexpr (EPF) {
    // 3a. differences of the new row q
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.g[c][Q0] = gq[c];
      s.du[c][Q0] = Abs2(s.g[c][Q1] - gq[c]);
      s.dl[c][Q0] = Abs2(v2f{FromLeft(gq[c].y) - gq[c].x, gq[c].x - gq[c].y});
    }
    // 3b. plus-sums of row p = q-1 (order: up, left, centre, right, down)
    v2f pv = {0.0f, 0.0f}, ph = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2f du_c = s.du[c][Q1], dl_c = s.dl[c][Q1];
      v2f v = AddLeft(s.du[c][Q2], du_c);
      v = v + du_c;
      v = AddRight(v, du_c);
      v = v + s.du[c][Q0];
      v2f h = AddLeft(s.dl[c][Q2], dl_c);
      h = h + dl_c;
      h = AddRight(h, dl_c);  // |p(x,y) - p(x+1,y)| = Dl(x+1,y)
      h = h + s.dl[c][Q0];
      pv = Fma2(v, P.ch_scale[c], pv);
      ph = Fma2(h, P.ch_scale[c], ph);
    }
    s.pv[Q1] = pv;
    s.ph[Q1] = ph;
    // 3c. EPF1 output row o = q-2
    o = q - 2;
    const float kMinSigma = -3.90524291751269967465540850526868f;
    // first row whose result is used: y_begin, or the row above it when EPF2 reads it
    if ((o & 7) == 0 || o == y_begin - (EPF == 2 ? 1 : 0)) {
      const int oc = o < 0 ? 0 : (o >= H ? H - 1 : o);
      const float is = f.inv_sigma[(size_t)(oc >> 3) * f.xsb + L.sx];
      // below the threshold the stage copies its input (stage_epf.cc:258-262):
      // -inf zeroes the four weights, and (c + 0) * rcp(1) == c exactly
      inv_sigma_blk = is < kMinSigma ? -__builtin_inff() : is;
    }
    const int iy = o & 7;
    const v2f mul = (iy == 0 || iy == 7) ? v2f{P.bsm[1], P.bsm[1]} : L.mul;
    const v2f inv_sigma = mul * inv_sigma_blk;
    const v2f wN = EpfW(s.pv[Q2], inv_sigma);
    const v2f wW = EpfW(s.ph[Q2], inv_sigma);
    const v2f wE = EpfW(v2f{s.ph[Q2].y, FromRight(s.ph[Q2].x)}, inv_sigma);
    const v2f wS = EpfW(s.pv[Q1], inv_sigma);
    v2f wsum = v2f{1.0f, 1.0f} + wN;
    wsum = wsum + wW;
    wsum = wsum + wE;
    wsum = wsum + wS;
    const v2f inv_w = {__builtin_amdgcn_rcpf(wsum.x), __builtin_amdgcn_rcpf(wsum.y)};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2f ctr = s.g[c][Q2];
      v2f a = ctr;
      a = Fma2(wN, s.g[c][Q3], a);
      a = FmaLeft(wW, ctr, a);
      a = FmaRight(wE, ctr, a);
      a = Fma2(wS, s.g[c][Q1], a);
      outv[c] = a * inv_w;
    }
    if constexpr (EPF == 2) {
      // 3d. third EPF stage (EPF2Stage, stage_epf.cc:393-492) on the rows the second one
      // produces: new row o enters, row o2 = o - 1 leaves.  Its SADs are single pixel
      // differences, shared between the two pixels they separate (|a - b| is symmetric).
      constexpr int E0 = Q2, E1 = Q3, E2 = Q0;  // rows o, o-1, o-2
      if (L.edge) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float from_right = FromRight(outv[c].x), from_left = FromLeft(outv[c].y);
          outv[c].y = L.fix_left ? from_right : (L.fix_right_odd ? outv[c].x : outv[c].y);
          outv[c].x = L.fix_right_even ? from_left : outv[c].x;
        }
      }
      v2f dv = Abs2(outv[0] - s.e[0][E1]) * P.ch_scale[0];
      dv = Fma2(Abs2(outv[1] - s.e[1][E1]), P.ch_scale[1], dv);
      dv = Fma2(Abs2(outv[2] - s.e[2][E1]), P.ch_scale[2], dv);
#pragma unroll
      for (int c = 0; c < 3; c++) s.e[c][E0] = outv[c];
      s.dv[E0] = dv;
      const int o2 = o - 1;
      // |p(x) - p(x-1)| per column of row o2
      v2f dh;
      {
        const v2f c0 = s.e[0][E1], c1 = s.e[1][E1], c2 = s.e[2][E1];
        dh = Abs2(v2f{c0.x - FromLeft(c0.y), c0.y - c0.x}) * P.ch_scale[0];
        dh = Fma2(Abs2(v2f{c1.x - FromLeft(c1.y), c1.y - c1.x}), P.ch_scale[1], dh);
        dh = Fma2(Abs2(v2f{c2.x - FromLeft(c2.y), c2.y - c2.x}), P.ch_scale[2], dh);
      }
      if ((o2 & 7) == 0 || o2 == y_begin) {
        const int oc = o2 < 0 ? 0 : (o2 >= H ? H - 1 : o2);
        const float is = f.inv_sigma[(size_t)(oc >> 3) * f.xsb + L.sx];
        inv_sigma_blk2 = is < kMinSigma ? -__builtin_inff() : is;
      }
      const int iy2 = o2 & 7;
      const v2f mul2 = (iy2 == 0 || iy2 == 7) ? v2f{P.bsm[2], P.bsm[2]} : L.mul2;
      const v2f inv_sigma2 = mul2 * inv_sigma_blk2;
      // rows -1 and H are the mirrors of rows 0 and H-1: a zero difference, the centre as value
      const bool top = o2 == 0, bottom = o2 == H - 1;
      const v2f zero = {0.0f, 0.0f};
      const v2f wN2 = EpfW(top ? zero : s.dv[E1], inv_sigma2);
      const v2f wW2 = EpfW(dh, inv_sigma2);
      const v2f wE2 = EpfW(v2f{dh.y, FromRight(dh.x)}, inv_sigma2);
      const v2f wS2 = EpfW(bottom ? zero : dv, inv_sigma2);
      v2f wsum2 = v2f{1.0f, 1.0f} + wN2;
      wsum2 = wsum2 + wW2;
      wsum2 = wsum2 + wE2;
      wsum2 = wsum2 + wS2;
      const v2f inv_w2 = {__builtin_amdgcn_rcpf(wsum2.x), __builtin_amdgcn_rcpf(wsum2.y)};
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const v2f ctr = s.e[c][E1];
        v2f a = ctr;
        a = Fma2(wN2, top ? ctr : s.e[c][E2], a);
        a = FmaLeft(wW2, ctr, a);
        a = FmaRight(wE2, ctr, a);
        a = Fma2(wS2, bottom ? ctr : s.e[c][E0], a);
        outv[c] = a * inv_w2;
      }
      o = o2;
    }
  } else {
    o = q;
#pragma unroll
    for (int c = 0; c < 3; c++) outv[c] = gq[c];
  }
  // 4. emit
  if (o >= y_begin && o < y_end && !((f.debug & 4) && outv[0].x != 12345.678f)) {
    EmitPair<OUTK, FMT>(outv, L, o, f, P);
  }
}

template <int GAB, int EPF>
struct FastGeom {
  static constexpr int HX = GAB + (EPF >= 1 ? 2 : 0) + (EPF == 2 ? 1 : 0);  // halo rows / columns each side
  static constexpr int HXP = (HX + 1) & ~1;       // in whole column pairs
  static constexpr int USE = 128 - 2 * HXP;       // output columns per wave
};

template <int GAB, int EPF, int OUTK, int FMT>
__global__ __launch_bounds__(256) void k_filters_fast(DevFrame f, FilterParams P, int RH) {
  using G = FastGeom<GAB, EPF>;
  constexpr int HX = G::HX, HXP = G::HXP, USE = G::USE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float __attribute__((address_space(3)))* dither_lds = nullptr;
  if constexpr (OUTK == JXLHIP_OUT_PACKED) {  // before any wave leaves: whole-workgroup barrier
    __shared__ float s_dither[1024];
    if (P.fmt.sample_type == JXLHIP_SAMPLE_U8) {  // uniform
      for (int i = threadIdx.x; i < 1024; i += 256) s_dither[i] = P.dither[i];
      __syncthreads();
    }
    dither_lds = (const float __attribute__((address_space(3)))*)s_dither;
  }
  const int strip = blockIdx.x * 4 + wave;
  const int W = (int)f.xsize, H = (int)f.ysize;
  const int x_first = strip * USE;  // first output column of the wave (even)
  if (x_first >= W) return;
  const int y_begin = (int)f.fy0 + blockIdx.y * RH;
  const int y_end = min(y_begin + RH, (int)f.fy1);
  if (y_begin >= y_end) return;
  Lane L;
  L.gx = x_first - HXP + 2 * lane;
  L.dither = dither_lds;
  {
    // the lane's two columns, mirrored into the image, always fall into one
    // aligned pair of plane columns (the planes are allocated in whole 8x8
    // tiles, so column W exists when W is odd)
    const int m0 = MirrorF(L.gx, W), m1 = MirrorF(L.gx + 1, W);
    const int base = m0 & ~1;
    L.sel0 = m0 & 1;
    L.sel1 = m1 & 1;
    L.byte_off = ((uint32_t)(base >> 3) * 64u + (uint32_t)(base & 7)) * 4u;
    L.edge = x_first - HXP < 0 || x_first - HXP + 128 > W;
    const bool lane_in = lane >= HXP / 2 && lane < 64 - HXP / 2;
    L.out0 = lane_in && L.gx < W;
    L.out1 = lane_in && L.gx + 1 < W;
    const int gxc = L.gx < 0 ? 0 : (L.gx >= W ? W - 1 : L.gx);
    L.sx = gxc >> 3;
    const int ix = gxc & 7;
    // columns gx, gx+1 (gx even inside the image): only gx can be a block's
    // first column and only gx+1 its last
    L.mul = v2f{ix == 0 ? P.bsm[1] : P.sm[1], ix == 6 ? P.bsm[1] : P.sm[1]};
    L.mul2 = v2f{ix == 0 ? P.bsm[2] : P.sm[2], ix == 6 ? P.bsm[2] : P.sm[2]};
    L.fix_left = L.gx == -2;
    L.fix_right_even = L.gx == W;       // only reached when W is even (gx is even)
    L.fix_right_odd = L.gx == W - 1;    // W odd
  }
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
  State s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int pr = r_first + k;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    const size_t off = RowOffset(f, MirrorF(pr, H));
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.pre[c][k] = LoadPair(f.xyb[c] + off, L);
      s.in[c][k] = v2f{0.0f, 0.0f};
      s.hs[c][k] = v2f{0.0f, 0.0f};
      s.g[c][k] = v2f{0.0f, 0.0f};
      s.du[c][k] = v2f{0.0f, 0.0f};
      s.dl[c][k] = v2f{0.0f, 0.0f};
      s.e[c][k] = v2f{0.0f, 0.0f};
    }
    s.pv[k] = v2f{0.0f, 0.0f};
    s.ph[k] = v2f{0.0f, 0.0f};
    s.dv[k] = v2f{0.0f, 0.0f};
  }
  float inv_sigma_blk = -1.0f, inv_sigma_blk2 = -1.0f;
  for (int r = r_first; r <= r_last; r += 4) {
    Step<GAB, EPF, OUTK, FMT, 0>(s, r, f, P, L, prefetch_last_row, y_begin, y_end, inv_sigma_blk,
                                  inv_sigma_blk2);
    Step<GAB, EPF, OUTK, FMT, 1>(s, r + 1, f, P, L, prefetch_last_row, y_begin, y_end, inv_sigma_blk,
                                  inv_sigma_blk2);
    Step<GAB, EPF, OUTK, FMT, 2>(s, r + 2, f, P, L, prefetch_last_row, y_begin, y_end, inv_sigma_blk,
                                  inv_sigma_blk2);
    Step<GAB, EPF, OUTK, FMT, 3>(s, r + 3, f, P, L, prefetch_last_row, y_begin, y_end, inv_sigma_blk,
                                  inv_sigma_blk2);
  }
}

// Rows per wave.  Every wave costs (RH + 2*HX) row steps and all waves of a
// launch take the same time, so the launch runs in ceil(workgroups / resident
// workgroups) generations: pick the RH that minimises generations * steps
// instead of leaving a mostly empty last generation.  Resident capacity: 256
// CUs x 2 workgroups (2 waves per SIMD at < 256 VGPRs).  JXLHIP_FILTER_RH
// overrides.
int FilterRowsPerWave(unsigned wgx, unsigned rows, int hx) {
  static const int forced = [] {
    const char* e = getenv("JXLHIP_FILTER_RH");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) return forced;
  static const unsigned resident = [] {
    const char* e = getenv("JXLHIP_FILTER_RESIDENT");
    return e ? (unsigned)atoi(e) : 256u * 2u;
  }();
  int best = 64;
  double best_cost = 1e30;
  for (int rh = 16; rh <= 512; rh += 1) {
    const unsigned wgs = wgx * ((rows + rh - 1) / rh);
    const unsigned gens = (wgs + resident - 1) / resident;
    const double cost = (double)gens * (rh + 2 * hx + 6);  // +6: per-wave prologue
    if (cost < best_cost) {
      best_cost = cost;
      best = rh;
    }
  }
  return best;
}

template <int GAB, int EPF, int OUTK, int FMT = -1>
void LaunchFastT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  using G = FastGeom<GAB, EPF>;
  const unsigned strips = (f.xsize + G::USE - 1) / G::USE;
  const unsigned wgx = (strips + 3) / 4;
  const int RH = FilterRowsPerWave(wgx, f.fy1 - f.fy0, G::HX);
  const dim3 grid(wgx, (f.fy1 - f.fy0 + RH - 1) / RH);
  hipLaunchKernelGGL((k_filters_fast<GAB, EPF, OUTK, FMT>), grid, dim3(256), 0, st, f, p, RH);
}

// The formats djxl writes most (8-bit sRGB for PNG / PPM / JPEG-like consumers,
// 16-bit sRGB) get a kernel with the format fixed at compile time; everything
// else takes the one that reads the format from its launch parameters.
template <int GAB, int EPF>
void LaunchPackedT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  const jxlhip_output_format& o = p.fmt;
  if (o.transfer == JXLHIP_TF_SRGB && !o.swap_endianness) {
    if (o.sample_type == JXLHIP_SAMPLE_U8 && o.num_channels == 3)
      return LaunchFastT<GAB, EPF, 2, FormatId(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U8, 3)>(f, p, st);
    if (o.sample_type == JXLHIP_SAMPLE_U8 && o.num_channels == 4)
      return LaunchFastT<GAB, EPF, 2, FormatId(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U8, 4)>(f, p, st);
    if (o.sample_type == JXLHIP_SAMPLE_U16 && o.num_channels == 3)
      return LaunchFastT<GAB, EPF, 2, FormatId(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U16, 3)>(f, p, st);
  }
  LaunchFastT<GAB, EPF, 2, -1>(f, p, st);
}

}  // namespace

bool LaunchFiltersFast(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                       int output_kind, hipStream_t st) {
  if (epf_iters > 2) return false;  // three iterations add EPF0 (7x7 reach): generic LDS kernel
  if (f.xsize < 16) return false;  // multiply mirrored columns: generic kernel
#define JXLHIP_FAST(G, E)                                  \
  if (gab == G && epf_iters == E) {                        \
    if (output_kind == 0) LaunchFastT<G, E, 0>(f, p, st);  \
    else if (output_kind == 1) LaunchFastT<G, E, 1>(f, p, st); \
    else LaunchPackedT<G, E>(f, p, st);                    \
    return true;                                           \
  }
  JXLHIP_FAST(0, 0)  // no loop filter: the same row march is a streaming block-major -> RGB conversion
  JXLHIP_FAST(1, 0)
  JXLHIP_FAST(0, 1)
  JXLHIP_FAST(1, 1)
  JXLHIP_FAST(0, 2)
  JXLHIP_FAST(1, 2)
#undef JXLHIP_FAST
  return false;
}

}  // namespace jxlhip
