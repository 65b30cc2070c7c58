// kernels_filters_fast.hip -- phase 2 as a register row march: [Gaborish] [EPF1] [EPF2] XYB->RGB (every
// stream below distance ~3, i.e. the BASELINE d1.0 configuration; with three EPF iterations this kernel runs
// the EPF1 + EPF2 + output part behind k_epf0, kernels_epf0.hip), written for the CDNA4 wavefront instead of LDS:
//
//   * one wave = 128 adjacent pixel COLUMNS (each lane owns an aligned PAIR of
//     columns), marching down the rows of its band.  Two pixels per lane turn
//     the filter arithmetic into packed fp32 (v_pk_add/mul/fma_f32: two results
//     per VALU issue) and halve the cross-lane traffic per pixel;
//   * horizontal neighbours inside the pair are free, the two outside it come
//     from the neighbouring LANE through DPP wave_shr/wave_shl (no LDS, no
//     barrier);
//   * vertical neighbours come from a sliding window of rows kept in registers
//     (rings of 8 / 4 slots, slot = row & 7 / & 3, resolved at compile time by
//     unrolling the row loop 8x);
//   * input rows are prefetched in bursts of four rows = whole 128-byte lines of
//     the block-major planes (one 8-byte load per lane, row and channel; the row
//     base is scalar), output rows leave as 24-byte non-temporal RGB stores.
//
// The kernel is written against its VALU ISSUE count (SQ counters of round 1: 218 VALU instructions
// per row step and wave, the SIMDs' VALU ports 57 % busy at 2 waves per SIMD -- issue-bound, not
// latency-bound).  Round 2 brought the row step to ~118 VALU instructions (tools/isa_loops.py on the
// -S listing): channel-summed difference images before the plus-shaped sums, weights through the
// packed FMA's [0, 1] output clamp, XYB -> RGB on pixel pairs with per-half matrix rows picked by
// op_sel, DPP operands folded into VOP2 instructions (v_add/v_sub/v_fmac ..._dpp), one 8-slot input
// ring instead of prefetch + input rings (no register copies between rings), SGPR-based addressing
// for every load and store, a running output-row pointer, part of the wave-uniform constants kept in
// VGPRs (the loop wanted more than 102 SGPRs).  Measured on MI355X (8K d1.0, JXLHIP_DEBUG
// ablations): arithmetic alone 135 -> 94 us; with the plane reads 127-137 us; whole kernel 215-235 us
// (866 MB: 3.9 TB/s, where a device copy moves 4.8-5.0) -- what remains is the block-major read
// path and the mixed read / write stream, not arithmetic.  Tried and measured
// without gain: 3 workgroups per CU, deeper single-row prefetch (see kAhead), routing the RGB row
// through LDS so that every store instruction writes whole 64-byte lines.
//
// EPF1 (lib/jxl/render_pipeline/stage_epf.cc:225-367) is evaluated through an
// regrouping of the reference's sums: with Du(x,y) = sum_c scale_c |p_c(x,y-1) - p_c(x,y)|
// and Dl(x,y) = sum_c scale_c |p_c(x-1,y) - p_c(x,y)|, the four SADs of pixel (x,y) are the
// plus-shaped sums  PV(x,y), PH(x,y), PH(x+1,y), PV(x,y+1)  of Du / Dl -- the reference's 15
// non-negative terms per SAD in another association (per channel first there, per position first
// here), each plus-sum computed once per pixel instead of four times.
//
// Border rule (simple_render_pipeline.cc:129-164): stages read their input
// mirrored at the true image edge.  Gaborish of the mirrored input IS the
// mirrored Gaborish output (symmetric kernel, commutative pair sums), so halo
// lanes/rows outside the image simply run on mirrored input; this kernel is
// only used when no stage follows an EPF stage, where that identity is all
// that is needed.  Where EPF2 follows EPF1 the one out-of-image column / row it reads is the mirror = the edge pixel
// itself (Lane::fix_*).  Frames narrower or lower than 16 px use the generic kernel (kernels_filters.hip).
#include <stdlib.h>

#include "env_switches.h"

#include "filters_march.h"

// This file is compiled four times: as itself (part 0: the entry points, the float / planar outputs, the general
// packed format and the three fixed formats of round 2) and, through kernels_filters_fast_{b,c,d}.hip, for three
// more sets of packed formats fixed at compile time (parts 1..3) -- six stage lists each, in parallel.
#ifndef JXLHIP_FAST_PART
#define JXLHIP_FAST_PART 0
#endif

namespace jxlhip {

namespace {

template <int GAB, int EPF>
struct FastGeom {
  static constexpr int HX = GAB + (EPF >= 1 ? 2 : 0) + (EPF == 2 ? 1 : 0);  // halo rows / columns each side
  static constexpr int HXP = (HX + 1) & ~1;       // in whole column pairs
  static constexpr int USE = 128 - 2 * HXP;       // output columns per wave
};

template <int GAB, int EPF, int OUTK, int FMT, bool EDGE, int DBG, int SRC>
__device__ __forceinline__ void March(const DevFrame& f, const FilterParams& P, Lane& L, int y_begin,
                                      int y_end) {
  constexpr int HX = FastGeom<GAB, EPF>::HX;
  const int H = (int)f.ysize;
  // rows: input rows r = y_begin - HX .. y_end + HX - 1; the pipeline emits
  // row r - HX at step r.
  const int r_first = y_begin - HX;
  const int r_last = y_end + HX - 1;
  // the prefetcher runs kAhead rows ahead: clamp to the last row this
  // context holds (plane rows cover [y0 - halo, y1_padded + halo))
  const int plane_last = f.plane_y0 + (int)f.plane_tile_rows * 8 - 1;
  int prefetch_last_row = r_last;
  // mirrored rows always fall inside the plane; direct rows must too
  if (prefetch_last_row > plane_last && prefetch_last_row < H) prefetch_last_row = plane_last;
  State s;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // rows r_first .. r_first + kAhead - 1 are in flight when the first step runs; the slots of the
    // (not yet existing) rows above them start as zero like the other rings
    const bool fetch = k < kAhead;
    int pr = r_first + k;
    pr = pr > prefetch_last_row ? prefetch_last_row : pr;
    const uint32_t off = SrcRowOffset<SRC>(f, Mirror1(pr, H));
    LaneOffset(L.byte_off);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (fetch) s.x[c][k] = LoadPair<EDGE>((const char*)f.xyb[c] + off, L);
      else s.x[c][k] = v2f{0.0f, 0.0f};
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.hs[c][k] = v2f{0.0f, 0.0f};
      s.g[c][k] = v2f{0.0f, 0.0f};
      s.e[c][k] = v2f{0.0f, 0.0f};
    }
    s.du[k] = v2f{0.0f, 0.0f};
    s.dl[k] = v2f{0.0f, 0.0f};
    s.pv[k] = v2f{0.0f, 0.0f};
    s.ph[k] = v2f{0.0f, 0.0f};
    s.dv[k] = v2f{0.0f, 0.0f};
  }
  float inv_sigma_blk = -1.0f, inv_sigma_blk2 = -1.0f;
  const XybConsts KC = MakeXybConsts(P);
  // output row of step r is row r - HX: a running pointer instead of a 64-bit product per row
  const size_t out_row_bytes = OUTK == JXLHIP_OUT_XYB_PLANAR ? P.out_stride * 4 : P.out_stride;
  char* out_row = (char*)P.out + (ptrdiff_t)(r_first - HX - (int)f.y0) * (ptrdiff_t)out_row_bytes;
#define JXLHIP_STEP(K)                                                                                         \
  Step<GAB, EPF, OUTK, FMT, K, EDGE, DBG, SRC>(s, r + K, f, P, L, prefetch_last_row, y_begin, y_end,        \
                                               inv_sigma_blk, inv_sigma_blk2, out_row, KC);                  \
  out_row += out_row_bytes
  for (int r = r_first; r <= r_last; r += 8) {
    JXLHIP_STEP(0);
    JXLHIP_STEP(1);
    JXLHIP_STEP(2);
    JXLHIP_STEP(3);
    if (r + 4 > r_last) break;
    JXLHIP_STEP(4);
    JXLHIP_STEP(5);
    JXLHIP_STEP(6);
    JXLHIP_STEP(7);
  }
#undef JXLHIP_STEP
}

template <int GAB, int EPF, int OUTK, int FMT, int DBG, int SRC = SRC_PLANES>
__global__ __launch_bounds__(256, EPF == 2 ? 2 : 3) void k_filters_fast(DevFrame f, FilterParams P, int RH) {
  using G = FastGeom<GAB, EPF>;
  constexpr int HXP = G::HXP, USE = G::USE;
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
  const int W = (int)f.xsize;
  const int x_first = strip * USE;  // first output column of the wave (even)
  if (x_first >= W) return;
  const int y_begin = (int)f.fy0 + blockIdx.y * RH;
  const int y_end = min(y_begin + RH, (int)f.fy1);
  if (y_begin >= y_end) return;
  Lane L;
  L.gx = x_first - HXP + 2 * lane;
  L.dither = dither_lds;
  // the lane's two columns, mirrored into the image, always fall into one
  // aligned pair of plane columns (the planes are allocated in whole 8x8
  // tiles, so column W exists when W is odd)
  const int m0 = MirrorF(L.gx, W), m1 = MirrorF(L.gx + 1, W);
  const int base = m0 & ~1;
  L.sel0 = m0 & 1;
  L.sel1 = m1 & 1;
  L.byte_off = SRC == SRC_LINEAR ? (uint32_t)base * 4u : ((uint32_t)(base >> 3) * 64u + (uint32_t)(base & 7)) * 4u;
  const bool edge = x_first - HXP < 0 || x_first - HXP + 128 > W;  // wave-uniform
  const bool lane_in = lane >= HXP / 2 && lane < 64 - HXP / 2;
  L.out0 = lane_in && L.gx < W;
  L.out1 = lane_in && L.gx + 1 < W;
  const int gxc = L.gx < 0 ? 0 : (L.gx >= W ? W - 1 : L.gx);
  L.sx4 = (uint32_t)(gxc >> 3) * 4u;
  L.out_off = (uint32_t)(L.gx < 0 ? 0 : L.gx) * (OUTK == JXLHIP_OUT_LINEAR_RGB_F32 ? 12u : 4u);
  const int ix = gxc & 7;
  // columns gx, gx+1 (gx even inside the image): only gx can be a block's
  // first column and only gx+1 its last
  L.mul = v2f{ix == 0 ? P.bsm[1] : P.sm[1], ix == 6 ? P.bsm[1] : P.sm[1]};
  L.mul2 = v2f{ix == 0 ? P.bsm[2] : P.sm[2], ix == 6 ? P.bsm[2] : P.sm[2]};
  L.fix_left = L.gx == -2;
  L.fix_right_even = L.gx == W;       // only reached when W is even (gx is even)
  L.fix_right_odd = L.gx == W - 1;    // W odd
  if (edge) March<GAB, EPF, OUTK, FMT, true, DBG, SRC>(f, P, L, y_begin, y_end);
  else March<GAB, EPF, OUTK, FMT, false, DBG, SRC>(f, P, L, y_begin, y_end);
}

// Rows per wave.  Every wave costs (RH + 2*HX) row steps and all waves of a
// launch take the same time, so the launch runs in ceil(workgroups / resident
// workgroups) generations: pick the RH that minimises generations * steps
// instead of leaving a mostly empty last generation.  Resident capacity: two
// workgroups per compute unit of the device.  JXLHIP_FILTER_RH (sampled when a
// context is created, env_switches.h) overrides.
int FilterRowsPerWave(unsigned wgx, unsigned rows, int hx) {
  const int forced = jxlhip_env::Get().filter_rh.load(std::memory_order_relaxed);
  if (forced > 0) return forced;
  const unsigned resident = DeviceCus() * 2u;
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
  if constexpr (GAB == 0 && EPF == 2) {
    if (f.linear_stride) {  // the input is k_epf0's row-major plane set
      using G = FastGeom<GAB, EPF>;
      const unsigned strips = (f.xsize + G::USE - 1) / G::USE;
      const unsigned wgx = (strips + 3) / 4;
      const int RH = FilterRowsPerWave(wgx, f.fy1 - f.fy0, G::HX);
      const dim3 grid(wgx, (f.fy1 - f.fy0 + RH - 1) / RH);
      hipLaunchKernelGGL((k_filters_fast<GAB, EPF, OUTK, FMT, 0, SRC_LINEAR>), grid, dim3(256), 0, st, f, p, RH);
      return;
    }
  }
  using G = FastGeom<GAB, EPF>;
  const unsigned strips = (f.xsize + G::USE - 1) / G::USE;
  const unsigned wgx = (strips + 3) / 4;
  const int RH = FilterRowsPerWave(wgx, f.fy1 - f.fy0, G::HX);
  const dim3 grid(wgx, (f.fy1 - f.fy0 + RH - 1) / RH);
  hipLaunchKernelGGL((k_filters_fast<GAB, EPF, OUTK, FMT, 0>), grid, dim3(256), 0, st, f, p, RH);
}

// Packed formats with a kernel of their own (the format fixed at compile time): what djxl writes most -- 8-bit sRGB
// for PNG / PPM, 16-bit sRGB (round 2, part 0) -- and, round 3: 16-bit sRGB RGBA and the BIG-ENDIAN 16-bit forms
// (PNG / PNM are big-endian), float sRGB / linear (PFM, NPY, API clients), half-float RGBA (HDR canvases), 16-bit PQ
// (HDR PNG).  Everything else takes the kernel that reads the format from its launch parameters -- at twice the time
// (per-sample wave-uniform branches, 256 VGPRs and spills; profiles/r03_packed_fixed_formats.txt).
#define JXLHIP_FIXED_FORMATS_0(X)                    \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U8, 3, 0)          \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U8, 4, 0)          \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U16, 3, 0)
#define JXLHIP_FIXED_FORMATS_1(X)                    \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U16, 4, 0)         \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U16, 3, 1)         \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_U16, 4, 1)
#define JXLHIP_FIXED_FORMATS_2(X)                    \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_F32, 3, 0)         \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_F32, 4, 0)         \
  X(JXLHIP_TF_LINEAR, JXLHIP_SAMPLE_F32, 4, 0)
#define JXLHIP_FIXED_FORMATS_3(X)                    \
  X(JXLHIP_TF_SRGB, JXLHIP_SAMPLE_F16, 4, 0)         \
  X(JXLHIP_TF_LINEAR, JXLHIP_SAMPLE_F16, 4, 0)       \
  X(JXLHIP_TF_PQ, JXLHIP_SAMPLE_U16, 3, 1)           \
  X(JXLHIP_TF_PQ, JXLHIP_SAMPLE_U16, 4, 1)
#if JXLHIP_FAST_PART == 0
#define JXLHIP_FIXED_FORMATS(X) JXLHIP_FIXED_FORMATS_0(X)
#elif JXLHIP_FAST_PART == 1
#define JXLHIP_FIXED_FORMATS(X) JXLHIP_FIXED_FORMATS_1(X)
#elif JXLHIP_FAST_PART == 2
#define JXLHIP_FIXED_FORMATS(X) JXLHIP_FIXED_FORMATS_2(X)
#else
#define JXLHIP_FIXED_FORMATS(X) JXLHIP_FIXED_FORMATS_3(X)
#endif

// this part's fixed formats: launches and returns true when p.fmt is one of them
template <int GAB, int EPF>
bool LaunchFixedT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  const jxlhip_output_format& o = p.fmt;
#define JXLHIP_TRY(TF, ST, NC, SW)                                                                              \
  if (o.transfer == TF && o.sample_type == ST && o.num_channels == NC && (o.swap_endianness != 0) == (SW != 0)) { \
    LaunchFastT<GAB, EPF, 2, FormatId(TF, ST, NC, SW)>(f, p, st);                                               \
    return true;                                                                                                \
  }
  JXLHIP_FIXED_FORMATS(JXLHIP_TRY)
#undef JXLHIP_TRY
  return false;
}

bool LaunchFixedPart(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st) {
#define JXLHIP_FAST(G, E) \
  if (gab == G && epf_iters == E) return LaunchFixedT<G, E>(f, p, st);
  JXLHIP_FAST(0, 0)
  JXLHIP_FAST(1, 0)
  JXLHIP_FAST(0, 1)
  JXLHIP_FAST(1, 1)
  JXLHIP_FAST(0, 2)
  JXLHIP_FAST(1, 2)
#undef JXLHIP_FAST
  return false;
}

}  // namespace

#if JXLHIP_FAST_PART == 1
bool LaunchFastFixedB(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st) {
  return LaunchFixedPart(f, p, gab, epf_iters, st);
}
#elif JXLHIP_FAST_PART == 2
bool LaunchFastFixedC(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st) {
  return LaunchFixedPart(f, p, gab, epf_iters, st);
}
#elif JXLHIP_FAST_PART == 3
bool LaunchFastFixedD(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st) {
  return LaunchFixedPart(f, p, gab, epf_iters, st);
}
#else
bool LaunchFastFixedB(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st);
bool LaunchFastFixedC(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st);
bool LaunchFastFixedD(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, hipStream_t st);

bool FastFixedFormat(const jxlhip_output_format& o) {
#define JXLHIP_IS(TF, ST, NC, SW) \
  if (o.transfer == TF && o.sample_type == ST && o.num_channels == NC && (o.swap_endianness != 0) == (SW != 0)) return true;
  JXLHIP_FIXED_FORMATS_0(JXLHIP_IS)
  JXLHIP_FIXED_FORMATS_1(JXLHIP_IS)
  JXLHIP_FIXED_FORMATS_2(JXLHIP_IS)
  JXLHIP_FIXED_FORMATS_3(JXLHIP_IS)
#undef JXLHIP_IS
  return false;
}

bool LaunchFiltersFast(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                       int output_kind, hipStream_t st) {
  if (epf_iters > 2) return false;  // three iterations: LaunchEpf0 (kernels_epf0.hip) first, then this with (0, 2)
  if (f.xsize < 16 || f.ysize < 16) return false;  // multiply mirrored columns / rows: generic kernel
  // row offsets inside a plane are 32-bit
  if ((uint64_t)f.plane_tile_rows * f.tile_stride * 256u >= (1ull << 32)) return false;
  if (f.linear_stride && !(gab == 0 && epf_iters == 2)) return false;
  if (gab < 0 || gab > 1 || epf_iters < 0) return false;
  if (output_kind == JXLHIP_OUT_PACKED &&
      (LaunchFixedPart(f, p, gab, epf_iters, st) || LaunchFastFixedB(f, p, gab, epf_iters, st) ||
       LaunchFastFixedC(f, p, gab, epf_iters, st) || LaunchFastFixedD(f, p, gab, epf_iters, st)))
    return true;
#define JXLHIP_FAST(G, E)                                  \
  if (gab == G && epf_iters == E) {                        \
    if (output_kind == 0) LaunchFastT<G, E, 0>(f, p, st);  \
    else if (output_kind == 1) LaunchFastT<G, E, 1>(f, p, st); \
    else LaunchFastT<G, E, 2, -1>(f, p, st);               \
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
#endif

}  // namespace jxlhip
