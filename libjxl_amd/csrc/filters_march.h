// filters_march.h -- the register row march shared by the phase-2 kernel (kernels_filters_fast.hip:
// rows come from the block-major XYB planes) and the fused kernel (kernels_fused.hip: rows come from a
// per-wave LDS slab that the wave fills from the coefficient stream).  See kernels_filters_fast.hip
// for the design notes.
#ifndef JXLHIP_FILTERS_MARCH_H_
#define JXLHIP_FILTERS_MARCH_H_

#include "dev_common.h"
#include "emit.h"
#include "kernels.h"

namespace jxlhip {
namespace {

__device__ __forceinline__ int MirrorF(int x, int n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}
// one reflection: rows of this kernel overshoot the image by less than 16 and LaunchFiltersFast only
// takes frames of at least 16 rows
__device__ __forceinline__ int Mirror1(int y, int n) {
  y = y < 0 ? -y - 1 : y;
  return y >= n ? 2 * n - 1 - y : y;
}

// byte offset of row y inside a block-major plane (the lane adds its tile column); 32-bit: the
// launcher sends planes of 4 GB and more to the generic kernel
__device__ __forceinline__ uint32_t RowOffset(const DevFrame& f, int y) {
  const uint32_t ry = (uint32_t)(y - f.plane_y0);
  return (ry >> 3) * (f.tile_stride * 256u) + ((ry & 7u) << 5);
}

// value held by the previous / next lane (0 at the wave's ends: those lanes
// are halo lanes whose results are never stored)
__device__ __forceinline__ float FromLeft(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float FromRight(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f Fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f Fma2(v2f a, float b, v2f c) {
  return __builtin_elementwise_fma(a, v2f{b, b}, c);
}

// Keeps a scalar result scalar: without it the SLP vectoriser fuses the two halves of a pair
// that are computed by DIFFERENT instructions (one with a DPP operand, one without) into one packed
// operation fed by explicit v_mov_b32_dpp / v_mov -- three instructions instead of two.
__device__ __forceinline__ float Scalar(float v) {
  asm("" : "+v"(v));
  return v;
}

// A per-lane 32-bit byte offset that the optimiser cannot hoist out of the row loop as a 64-bit
// value: added to a wave-uniform base inside the same basic block it becomes the VGPR offset of an
// SGPR-based access (global_load v, v_off, s[base:base+1]) -- no VALU address arithmetic per access
// (hoisted, every load and store pays a v_lshl_add_u64).  The value is redefined IN PLACE: a copy
// per use would cost what the address arithmetic did.
__device__ __forceinline__ uint32_t LaneOffset(uint32_t& v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Input rows are requested in bursts of kBurst rows, the first of them kAhead rows before its step.
// Four rows of an 8x8 tile share a 128-byte line: asked for together the line crosses L2 -> L1 once
// (per-row requests found it evicted again: L1 -> L2 read requests were 3.7x the plane bytes).
// Measured on MI355X, 8K d1.0, kernel time: (kAhead, kBurst) = (2, 1) 0.237 ms, (1, 2) 0.238,
// (2, 2) 0.229, (1, 4) 0.215; prefetching further ahead with single-row requests is slower
// ((3, 1) 0.248, (4, 1) 0.251).
static constexpr int kAhead = 1;
static constexpr int kBurst = 4;  // 1, 2 or 4
static_assert(kAhead >= 1 && kAhead + kBurst <= 5 && (kBurst == 1 || kBurst == 2 || kBurst == 4),
              "input ring of 8: rows r-3 .. r+kAhead+kBurst-1 must fit");

// max(0, 1 + sad * inv_sigma) (stage_epf.cc:46-50).  sad >= 0 and inv_sigma < 0, so the value
// never exceeds 1 and the [0, 1] output clamp of the packed FMA IS ZeroIfNegative: one VALU issue
// for the weights of two pixels (v_pk_fma + 2 v_max before).  A block whose sigma is below the
// filter threshold carries inv_sigma = -inf: every weight becomes 0 (0 * -inf = NaN clamps to 0
// as well: the kernel runs with DX10_CLAMP).
__device__ __forceinline__ v2f EpfW(v2f sad, v2f inv_sigma) {
  v2f w;
  asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,1,0] clamp" : "=v"(w) : "v"(sad), "v"(inv_sigma));
  return w;
}

// sum over the channels of scale[c] * |d[c]| for the two columns of the lane: scalar FMAs with the
// |.| source modifier (a packed FMA has none: the v_and pair it needs makes it three issues)
__device__ __forceinline__ v2f AbsScaleSum(const v2f* d, const FilterParams& P) {
  float x = __builtin_fabsf(d[0].x) * P.ch_scale[0];
  float y = __builtin_fabsf(d[0].y) * P.ch_scale[0];
  x = __builtin_fmaf(__builtin_fabsf(d[1].x), P.ch_scale[1], x);
  y = __builtin_fmaf(__builtin_fabsf(d[1].y), P.ch_scale[1], y);
  x = __builtin_fmaf(__builtin_fabsf(d[2].x), P.ch_scale[2], x);
  y = __builtin_fmaf(__builtin_fabsf(d[2].y), P.ch_scale[2], y);
  return v2f{Scalar(x), Scalar(y)};
}

// acc + (p(x-1), p(x)) and acc + (p(x+1), p(x+2)) for the column pair (x, x+1)
__device__ __forceinline__ v2f AddLeftS(v2f acc, v2f p) {
  return v2f{Scalar(FromLeft(p.y) + acc.x), Scalar(acc.y + p.x)};
}
__device__ __forceinline__ v2f AddRightS(v2f acc, v2f p) {
  return v2f{Scalar(acc.x + p.y), Scalar(FromRight(p.x) + acc.y)};
}
// a + w * (p(x-1), p(x)) and a + w * (p(x+1), p(x+2)).  The neighbouring lane's value enters as the
// DPP operand of a VOP2 v_fmac_f32 (the compiler leaves a v_mov_b32_dpp in front of a VOP3 v_fma_f32).
// p must be an OLD value (written at least two VALU instructions earlier: the DPP read hazard, which
// the compiler does not track through inline asm): callers pass ring rows of earlier steps only.
__device__ __forceinline__ v2f FmaLeftS(v2f w, v2f p, v2f a) {
  float ax = a.x;
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ax) : "v"(p.y), "v"(w.x));
  return v2f{ax, Scalar(__builtin_fmaf(w.y, p.x, a.y))};
}
__device__ __forceinline__ v2f FmaRightS(v2f w, v2f p, v2f a) {
  float ay = a.y;
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ay) : "v"(p.x), "v"(w.y));
  return v2f{Scalar(__builtin_fmaf(w.x, p.y, a.x)), ay};
}

struct State {
  // Input rows, one ring of 8: row r sits in slot r & 7 from its prefetch (4 rows ahead) until the
  // stages no longer read it -- as Gaborish input (rows r .. r-2) or, without Gaborish, as the
  // rows entering EPF (r .. r-3).  (Separate prefetch / input rings cost a register copy per row and
  // channel: a ring's slots are fixed registers across loop iterations.)  The row loop is unrolled
  // 8x so that every slot index is a compile-time constant.
  v2f x[3][8];
  v2f hs[3][4];   // GAB: left + right of the input rows
  v2f g[3][4];    // GAB && EPF: Gaborish output rows entering EPF
  // EPF1's SADs are sums over the three channels of scale[c] * |difference|; the sums over the
  // channels are taken FIRST (du: against the row above, dl: against the column to the left), the
  // plus-shaped sums run on those two images -- a third of the additions of per-channel
  // plus-sums, the same 15 non-negative terms per SAD in another association (a few ulp of the
  // SAD, far below what the weight's max(0, 1 + sad * inv_sigma) resolves)
  v2f du[4], dl[4];
  v2f pv[4], ph[4];
  v2f e[3][4];    // EPF == 2: EPF1 output rows entering EPF2
  v2f dv[4];      // EPF == 2: channel-weighted |row - row above| of the e rows
};

// per-lane constants
struct Lane {
  uint32_t byte_off;  // byte offset of the lane's aligned column pair inside a plane row
  bool sel0, sel1;    // edge waves: which half of the loaded pair each column takes
  int gx;             // first column of the pair (may lie outside the image)
  bool out0, out1;    // column is written by this wave
  v2f mul;            // EPF sigma multiplier of the two columns (border columns of an 8x8 block differ)
  v2f mul2;           // ... of the third EPF stage
  // EPF == 2, edge waves: the one out-of-image column EPF2 reads takes its mirror (= the
  // edge column): pair (-2,-1): .y <- column 0; pair (W, W+1): .x <- column W-1 (W even);
  // pair (W-1, W): .y <- .x (W odd)
  bool fix_left, fix_right_even, fix_right_odd;
  uint32_t sx4;       // 4 * block column of the pair for the sigma look-up (clamped)
  uint32_t out_off;   // byte offset of the pair's first sample inside an output row (float RGB / XYB planes)
  // packed 8-bit output: the dither pattern, staged in LDS (a global load per
  // sample would queue behind the row prefetch in the in-order vmcnt)
  const float __attribute__((address_space(3))) * dither;
  // fused kernel (SRC_LDS): the lane's (mirrored, aligned) column pair in row 0 / channel 0 of its wave's slab
  const float __attribute__((address_space(3))) * slab;
};

// Row source of the march
// SRC_LINEAR: planes in plain row-major order (DevFrame::linear_stride bytes per row), what k_epf0 writes for
// the EPF1 + EPF2 march: every load / store instruction of a wave then covers 512 contiguous bytes
enum RowSource : int { SRC_PLANES = 0, SRC_LDS = 1, SRC_LINEAR = 2 };
template <int SRC>
__device__ __forceinline__ uint32_t SrcRowOffset(const DevFrame& f, int y) {
  if constexpr (SRC == SRC_LINEAR) return (uint32_t)(y - f.plane_y0) * f.linear_stride;
  else return RowOffset(f, y);
}
// per-wave LDS slab of the fused kernel: one block row (8 pixel rows) x 128 columns x 3 channels
static constexpr int kSlabCols = 128;
static constexpr int kSlabPlaneFloats = 8 * kSlabCols;
static constexpr int kSlabBytes = 3 * kSlabPlaneFloats * 4;

template <bool EDGE>
__device__ __forceinline__ v2f LdsPair(const Lane& L, int c, int row) {
  typedef v2f __attribute__((address_space(3))) * P2;
  const v2f v = *(P2)(L.slab + c * kSlabPlaneFloats + row * kSlabCols);
  if constexpr (!EDGE) return v;
  return v2f{L.sel0 ? v.y : v.x, L.sel1 ? v.y : v.x};
}

// EDGE (wave-uniform, a template parameter of the march): the strip touches a mirrored image edge
template <bool EDGE>
__device__ __forceinline__ v2f LoadPair(const char* rowp, Lane& L) {
  const v2f v = *(const v2f*)(rowp + L.byte_off);
  if constexpr (!EDGE) return v;
  return v2f{L.sel0 ? v.y : v.x, L.sel1 ? v.y : v.x};
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// XYB -> linear RGB (emit.h XybToRgb, dec_xyb-inl.h:38-86) for the lane's two pixels at once.  The
// three results come out in the order the 24-byte store wants them -- (r0 g0) (b0 r1) (g1 b1) --
// by letting each packed operation pick its matrix row per half (the operand pairs {m0,m3},
// {m6,m0}, {m3,m6} ... live in SGPR pairs) and broadcast one of the two pixels through op_sel.
// Wave-uniform constants of the XYB -> RGB tail that are kept in VGPRs on purpose: with everything
// in SGPRs the row loop needs more than the 102 a wave has and pays ~14 v_readlane_b32 (spill
// reloads) per row
struct XybConsts {
  v2f b01, b23, b45;        // xyb_bias pairs
  v2f m0a, m1a, m2a;        // mcol[j][2..3]: rows 2 | 0 of column j
};
__device__ __forceinline__ v2f InVgpr(v2f v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ XybConsts MakeXybConsts(const FilterParams& P) {
  XybConsts k;
  k.b01 = InVgpr(v2f{P.xyb_bias[0], P.xyb_bias[1]});
  k.b23 = InVgpr(v2f{P.xyb_bias[2], P.xyb_bias[3]});
  k.b45 = InVgpr(v2f{P.xyb_bias[4], P.xyb_bias[5]});
  k.m0a = InVgpr(v2f{P.mcol[0][2], P.mcol[0][3]});
  k.m1a = InVgpr(v2f{P.mcol[1][2], P.mcol[1][3]});
  k.m2a = InVgpr(v2f{P.mcol[2][2], P.mcol[2][3]});
  return k;
}

struct RgbPairs {
  v2f p0, p1, p2;  // (r0, g0), (b0, r1), (g1, b1)
};
__device__ __forceinline__ RgbPairs XybToRgbPair(const v2f* v, const FilterParams& P, const XybConsts& K) {
  v2f gr = v[1] + v[0], gg = v[1] - v[0], gb = v[2];
  // xyb_bias = (-cbrt_bias[0..2], opsin_bias[0..2]): x - b == x + (-b) exactly; as an addition the
  // splat operand is one SGPR picked by op_sel instead of a duplicated pair
  gr = gr + v2f{K.b01.x, K.b01.x};
  gg = gg + v2f{K.b01.y, K.b01.y};
  gb = gb + v2f{K.b23.x, K.b23.x};
  const v2f mr = Fma2(gr * gr, gr, v2f{K.b23.y, K.b23.y});
  const v2f mg = Fma2(gg * gg, gg, v2f{K.b45.x, K.b45.x});
  const v2f mb = Fma2(gb * gb, gb, v2f{K.b45.y, K.b45.y});
  // mcol[j] = (m[j], m[3+j], m[6+j], m[j]): column j of the matrix, wrapped -- its three aligned /
  // overlapping pairs (rows 0|1, rows 2|0, rows 1|2) are the per-half constants of the three
  // results, two SGPRs each picked by op_sel
  const float(*mc)[4] = P.mcol;
  RgbPairs o;
  // (r0, g0): pixel 0 against rows 0 and 1
  o.p0 = Fma2(v2f{mb.x, mb.x}, v2f{mc[2][0], mc[2][1]},
              Fma2(v2f{mg.x, mg.x}, v2f{mc[1][0], mc[1][1]}, v2f{mr.x, mr.x} * v2f{mc[0][0], mc[0][1]}));
  // (b0, r1): pixel 0 against row 2, pixel 1 against row 0
  o.p1 = Fma2(mb, K.m2a, Fma2(mg, K.m1a, mr * K.m0a));
  // (g1, b1): pixel 1 against rows 1 and 2
  o.p2 = Fma2(v2f{mb.y, mb.y}, v2f{mc[2][1], mc[2][2]},
              Fma2(v2f{mg.y, mg.y}, v2f{mc[1][1], mc[1][2]}, v2f{mr.y, mr.y} * v2f{mc[0][1], mc[0][2]}));
  return o;
}

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

template <int OUTK, int FMT, bool EDGE>
__device__ __forceinline__ void EmitPair(const v2f* v, Lane& L, int gy, char* out_row, const FilterParams& P,
                                         const XybConsts& K) {
  if constexpr (OUTK == JXLHIP_OUT_PACKED) {
    // FromLinearStage + WriteToOutputStage (emit.h); the packed formats move
    // 3..16 bytes per pixel, a fraction of the float output
    using Sel = FmtSel<FMT>;
    char* row = out_row;
    const RgbPairs o = XybToRgbPair(v, P, K);
    const float a[3] = {o.p0.x, o.p0.y, o.p1.x}, b[3] = {o.p1.y, o.p2.x, o.p2.y};
    if (Sel::sample_type(P.fmt) == JXLHIP_SAMPLE_U8 && Sel::channels(P.fmt) == 3) {
      StoreRgb8Pair<Sel>(P, L, row, gy, a, b);  // all lanes: uses DPP
    } else if (EDGE ? (L.out0 && L.out1) : L.out0) {
      StorePackedPair<Sel>(P, L.dither, row, L.gx, gy, a, b);
    } else if (!EDGE) {
    } else if (L.out0) {
      StorePackedPixel<Sel>(P, L.dither, row, L.gx, gy, a);
    } else if (L.out1) {
      StorePackedPixel<Sel>(P, L.dither, row, L.gx + 1, gy, b);
    }
  } else if constexpr (OUTK == JXLHIP_OUT_LINEAR_RGB_F32) {
    float* dst = (float*)(out_row + LaneOffset(L.out_off));
    const RgbPairs o = XybToRgbPair(v, P, K);
    // inside the image the two columns of a pair are written or skipped together (only column W-1
    // of an odd width separates them: an edge wave)
    // The pair's two stores under an EXEC mask set and restored inside ONE asm statement: no s_cbranch_execz around
    // them, so the eight row steps of a group form ONE basic block -- the scheduler then fills the DPP / transcendental
    // hazard slots with useful instructions (k_fused_pc's interior loop: 63 -> 2 s_nop, 132 -> 60 scalar instructions
    // per eight rows; round 5: the kernel 179.2 -> 175.7 us at 8K, the same pixels).
    if constexpr (!EDGE) {
      unsigned long long saved;
      const unsigned long long mask = __ballot(L.out0);
      const f4u a = f4u{o.p0.x, o.p0.y, o.p1.x, o.p1.y};
      const f2u b = f2u{o.p2.x, o.p2.y};
      const uint32_t off = L.out_off;
      asm volatile(
          "s_and_saveexec_b64 %0, %1\n\t"
          "global_store_dwordx4 %2, %3, %5 nt\n\t"
          "global_store_dwordx2 %2, %4, %5 offset:16 nt\n\t"
          "s_mov_b64 exec, %0"
          : "=&s"(saved)
          : "s"(mask), "v"(off), "v"(a), "v"(b), "s"(out_row)
          : "memory", "scc");
      return;
    }
    if (EDGE ? (L.out0 && L.out1) : L.out0) {  // 24 contiguous bytes
      __builtin_nontemporal_store(f4u{o.p0.x, o.p0.y, o.p1.x, o.p1.y}, (f4u*)dst);
      __builtin_nontemporal_store(f2u{o.p2.x, o.p2.y}, (f2u*)(dst + 4));
    } else if (!EDGE) {
    } else if (L.out0) {
      __builtin_nontemporal_store(o.p0.x, dst);
      __builtin_nontemporal_store(o.p0.y, dst + 1);
      __builtin_nontemporal_store(o.p1.x, dst + 2);
    } else if (L.out1) {
      __builtin_nontemporal_store(o.p1.y, dst + 3);
      __builtin_nontemporal_store(o.p2.x, dst + 4);
      __builtin_nontemporal_store(o.p2.y, dst + 5);
    }
  } else {
    LaneOffset(L.out_off);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float* d = (float*)(out_row + (size_t)c * P.out_plane_stride * 4 + L.out_off);
      if (EDGE ? (L.out0 && L.out1) : L.out0) {
        __builtin_nontemporal_store(f2u{v[c].x, v[c].y}, (f2u*)d);
      } else if (!EDGE) {
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
// DBG: JXLHIP_DEBUG ablation bits of this kernel, compiled in only for the launch that asks for
// them (4: no output stores, 8: input rows stay in L1).
//
// KNOWN (fused kernel's marching wave, SRC_LDS, kernels_fused.hip MarchPC): what the caller knows about this step at
// COMPILE time.  A wave issues one instruction of any kind per ~5 cycles (tools/probes/valu_issue.hip), and the
// generic step spends a third of its issue slots on scalar bookkeeping -- is row o the first of a block row, is it a
// border row of its block, does it lie inside [y_begin, y_end), is row r + 1 a mirror row -- that has ONE answer for
// every chunk of rows that starts and ends on block rows and touches neither the frame's top nor its bottom:
//   kStepInterior   r == PH (mod 8), rows r - 8 .. r + 8 lie inside the frame (no mirror rows), y_begin is a multiple
//                   of 8: o & 7, the slab row of r + 1 and the top / bottom tests are constants
//   kStepEmit       (with kStepInterior) the row leaving the stages is written -- otherwise it is not; no range test
//   kStepFirst      (with kStepInterior) the first whole group of the chunk: EPF == 2 picks up the inv_sigma of the
//                   block row above for the one row of it that the third stage reads
enum StepKnown : int { kStepGeneric = 0, kStepInterior = 1, kStepEmit = 2, kStepFirst = 4 };
template <int GAB, int EPF, int OUTK, int FMT, int PH, bool EDGE, int DBG, int SRC = SRC_PLANES, int KNOWN = kStepGeneric>
__device__ __forceinline__ void Step(State& s, int r, const DevFrame& f, const FilterParams& P,
                                     Lane& L, int prefetch_last_row, int y_begin, int y_end,
                                     float& inv_sigma_blk, float& inv_sigma_blk2, char* out_row, const XybConsts& K,
                                     int slab_y0 = 0, float sigma_pre = 0.0f, float sigma_prev = 0.0f) {
  constexpr int S0 = PH & 3, S1 = (PH + 3) & 3, S2 = (PH + 2) & 3;  // r, r-1, r-2 in the 4-slot rings
  constexpr int X0 = PH & 7, X1 = (PH + 7) & 7, X2 = (PH + 6) & 7, X3 = (PH + 5) & 7;  // ... in the input ring
  const int H = (int)f.ysize;
  // 1. row r has arrived in slot X0; every kBurst-th step starts the loads of the next kBurst rows
  // (r + kAhead ...) into slots whose rows are dead.  Four rows of a tile share a 128-byte line: asked
  // for together, the line crosses L2 -> L1 once instead of once per row.
  if constexpr (SRC == SRC_LDS) {
    // fused kernel: row r (= slab row PH) was requested one step ago; request row r+1.  The first row
    // of the next block row is requested by the caller once the slab is refilled.
    if constexpr (PH < 7) {
      // slab row of image row r+1 (a mirror row near the frame's top / bottom)
      const int nrow = (KNOWN & kStepInterior) ? PH + 1 : Mirror1(r + 1, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][(PH + 1) & 7] = LdsPair<EDGE>(L, c, nrow);
    }
  } else if constexpr (PH % kBurst == 0) {
    LaneOffset(L.byte_off);
#pragma unroll
    for (int b = 0; b < kBurst; b++) {
      int pr = r + kAhead + b;
      pr = pr > prefetch_last_row ? prefetch_last_row : pr;
      if constexpr (DBG & 8) pr = y_begin + (pr & 7);  // ablation: reads stay in L1
      const uint32_t off = SrcRowOffset<SRC>(f, Mirror1(pr, H));
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][(PH + kAhead + b) & 7] = LoadPair<EDGE>((const char*)f.xyb[c] + off, L);
    }
  }
  // 2. Gaborish (stage_gaborish.cc:33-99) for row q = r-1
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
  constexpr int Q0 = GAB ? S1 : S0;  // slot of row q in the 4-slot rings
  constexpr int Q1 = (Q0 + 3) & 3, Q2 = (Q0 + 2) & 3, Q3 = (Q0 + 1) & 3;  // q-1, q-2, q-3
  const int q = GAB ? r - 1 : r;
  // rows q-1, q-2, q-3 of the image entering EPF: the Gaborish ring, or the input ring itself
  v2f gq1[3], gq2[3], gq3[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    gq1[c] = GAB ? s.g[c][Q1] : s.x[c][X1];
    gq2[c] = GAB ? s.g[c][Q2] : s.x[c][X2];
    gq3[c] = GAB ? s.g[c][Q3] : s.x[c][X3];
  }
  v2f outv[3];
  int o;
  if constexpr (EPF) {
    // 3a. channel-weighted differences of the new row q against the row above / the column to the left
    {
      v2f dvert[3], dhor[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        dvert[c] = gq1[c] - gq[c];
        dhor[c] = v2f{Scalar(FromLeft(gq[c].y) - gq[c].x), Scalar(gq[c].x - gq[c].y)};
        if constexpr (GAB) s.g[c][Q0] = gq[c];
      }
      s.du[Q0] = AbsScaleSum(dvert, P);
      s.dl[Q0] = AbsScaleSum(dhor, P);
    }
    // 3b. plus-sums of row p = q-1 (up, left, centre, right, down)
    {
      const v2f du_c = s.du[Q1], dl_c = s.dl[Q1];
      v2f v = AddLeftS(s.du[Q2], du_c);
      v = v + du_c;
      v = AddRightS(v, du_c);
      s.pv[Q1] = v + s.du[Q0];
      v2f h = AddLeftS(s.dl[Q2], dl_c);
      h = h + dl_c;
      h = AddRightS(h, dl_c);  // |p(x,y) - p(x+1,y)| = Dl(x+1,y)
      s.ph[Q1] = h + s.dl[Q0];
    }
  // 3c. EPF1 output row o = q-2
    o = q - 2;
    const float kMinSigma = -3.90524291751269967465540850526868f;
    // first row whose result is used: y_begin, or the row above it when EPF2 reads it
    constexpr bool kInt = (KNOWN & kStepInterior) != 0;
    static_assert(!kInt || SRC == SRC_LDS, "kStepInterior: the fused kernel's march");
    constexpr int kIy = (PH - GAB - 2 + 16) & 7;  // kInt: o & 7
    if (kInt ? (kIy == 0 || (EPF == 2 && (KNOWN & kStepFirst) && kIy == 7)) : ((o & 7) == 0 || o == y_begin - (EPF == 2 ? 1 : 0))) {
      // fused kernel: the caller loaded the block row's value at the start of the group of 8 rows -- a load
      // here would wait (in-order vmcnt) for the LDS-DMA copies issued in between.  (Row y_begin - 1, which EPF2
      // reads as its first "north" row, lies in the block row of the previous group.)
      float is = (kInt ? kIy == 0 : (o & 7) == 0) ? sigma_pre : sigma_prev;
      if constexpr (SRC != SRC_LDS) {
        const int oc = o < 0 ? 0 : (o >= H ? H - 1 : o);
        is = *(const float*)((const char*)(f.inv_sigma + (size_t)(oc >> 3) * f.xsb) + LaneOffset(L.sx4));
      }
      // below the threshold the stage copies its input (stage_epf.cc:258-262):
      // -inf zeroes the four weights, and (c + 0) * rcp(1) == c exactly
      inv_sigma_blk = is < kMinSigma ? -__builtin_inff() : is;
    }
    const int iy = kInt ? kIy : (o & 7);
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
      const v2f ctr = gq2[c];
      v2f a = Fma2(wN, gq3[c], ctr);
      a = FmaLeftS(wW, ctr, a);
      a = FmaRightS(wE, ctr, a);
      a = Fma2(wS, gq1[c], a);
      outv[c] = a * inv_w;
    }
    if constexpr (EPF == 2) {
      // 3d. third EPF stage (EPF2Stage, stage_epf.cc:393-492) on the rows the second one
      // produces: new row o enters, row o2 = o - 1 leaves.  Its SADs are single pixel
      // differences, shared between the two pixels they separate (|a - b| is symmetric).
      constexpr int E0 = Q2, E1 = Q3, E2 = Q0;  // rows o, o-1, o-2
      if constexpr (EDGE) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float from_right = FromRight(outv[c].x), from_left = FromLeft(outv[c].y);
          outv[c].y = L.fix_left ? from_right : (L.fix_right_odd ? outv[c].x : outv[c].y);
          outv[c].x = L.fix_right_even ? from_left : outv[c].x;
        }
      }
      v2f dvert[3], dhor[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        dvert[c] = outv[c] - s.e[c][E1];
        const v2f e1 = s.e[c][E1];
        dhor[c] = v2f{Scalar(FromLeft(e1.y) - e1.x), Scalar(e1.x - e1.y)};  // |p(x) - p(x-1)| per column of row o2
      }
      const v2f dv = AbsScaleSum(dvert, P);
      const v2f dh = AbsScaleSum(dhor, P);
#pragma unroll
      for (int c = 0; c < 3; c++) s.e[c][E0] = outv[c];
      s.dv[E0] = dv;
      const int o2 = o - 1;
      constexpr int kIy2 = (kIy + 7) & 7;  // kInt: o2 & 7
      if (kInt ? kIy2 == 0 : ((o2 & 7) == 0 || o2 == y_begin)) {
        float is = sigma_pre;
        if constexpr (SRC != SRC_LDS) {
          const int oc = o2 < 0 ? 0 : (o2 >= H ? H - 1 : o2);
          is = *(const float*)((const char*)(f.inv_sigma + (size_t)(oc >> 3) * f.xsb) + LaneOffset(L.sx4));
        }
        inv_sigma_blk2 = is < kMinSigma ? -__builtin_inff() : is;
      }
      const int iy2 = kInt ? kIy2 : (o2 & 7);
      const v2f mul2 = (iy2 == 0 || iy2 == 7) ? v2f{P.bsm[2], P.bsm[2]} : L.mul2;
      const v2f inv_sigma2 = mul2 * inv_sigma_blk2;
      // rows -1 and H are the mirrors of rows 0 and H-1: a zero difference, the centre as value
      const bool top = !kInt && o2 == 0, bottom = !kInt && o2 == H - 1;
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
        v2f a = Fma2(wN2, top ? ctr : s.e[c][E2], ctr);
        a = FmaLeftS(wW2, ctr, a);
        a = FmaRightS(wE2, ctr, a);
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
  if (((KNOWN & kStepInterior) ? (KNOWN & kStepEmit) != 0 : (o >= y_begin && o < y_end)) && !((DBG & 4) && outv[0].x != 12345.678f)) {
    EmitPair<OUTK, FMT, EDGE>(outv, L, o, out_row, P, K);
  }
}


// halo rows / columns a stage list needs on each side
template <int GAB, int EPF>
struct MarchGeom {
  static constexpr int HX = GAB + (EPF >= 1 ? 2 : 0) + (EPF == 2 ? 1 : 0);
};

}  // namespace
}  // namespace jxlhip
#endif
