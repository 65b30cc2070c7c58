// kernels_blocks.hip -- phase 1 of the VarDCT back-end on gfx950:
//   k_prepare      : per-group coefficient-offset scan + per-class work lists (the block walk of
//                    DecodeGroupImpl, lib/jxl/dec_group.cc:275-359, turned into a data-parallel
//                    scan) and ComputeSigma (lib/jxl/epf.cc:39-133)
//   k_transform_8  : every single-block strategy, no LDS: DCT8 row-per-lane (8 lanes per block,
//                    register transposes), the nine special 8x8 kinds lane-per-block
//   k_transform_r  : 16x8 .. 32x32 row-per-lane (k_transform_r16 / r32 when only one half has work),
//                    with the LDS-staged 64-point classes (64x64, 64x32, 32x64; k_transform_a when
//                    alone) on its first workgroups
//   k_large        : 128x64 .. 256x256, output plane used as scratch
// replacing DequantBlock + LowestFrequenciesFromDC + TransformToPixels
// (lib/jxl/dec_group.cc:115-181,431-450, lib/jxl/dec_transforms-inl.h:456-818).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include <initializer_list>

#include "blocks_common.h"
#include "env_switches.h"

namespace jxlhip {

// strategy -> covered blocks x | y << 8 | class << 16, in constant memory: indexed by a runtime strategy, the constexpr
// tables of dev_common.h would be copied to the thread's scratch
struct StrategyWord {
  uint32_t v[JXLHIP_NUM_STRATEGIES];
};
__host__ __device__ constexpr StrategyWord MakeStrategyWords() {
  StrategyWord w{};
  for (int s = 0; s < JXLHIP_NUM_STRATEGIES; s++)
    w.v[s] = (uint32_t)kCoveredX[s] | ((uint32_t)kCoveredY[s] << 8) | ((uint32_t)(uint8_t)ClassOfStrategy(s) << 16);
  return w;
}
__constant__ StrategyWord kStrategyWords = MakeStrategyWords();

// ---------------------------------------------------------------- k_prepare
// One workgroup (1024 threads) per AC group of the stripe; thread i owns cell
// (i / gw, i % gw) of the group's clipped block rectangle (BlockGroupRect,
// lib/jxl/frame_dimensions.h:70-77) in the raster order DecodeGroupImpl visits.
__global__ __launch_bounds__(1024) void k_prepare(DevFrame f, WorkLists wl, uint32_t gy_lo,
                                                  int with_sigma, float epf_quant_mul,
                                                  SharpLut lut) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint16_t wave_cls[16][kNumClasses];  // per-wave class counts -> bases
  __shared__ uint32_t wg_base[kNumClasses];
  __shared__ float cell_sq[1024];                 // sigma_quant of the covering varblock
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t gx = blockIdx.x % f.xsg;
  const uint32_t gy = gy_lo + blockIdx.x / f.xsg;
  // groups just outside the stripe only contribute their sigma cells (the EPF
  // stages evaluate halo rows of the neighbouring stripes)
  const bool in_stripe = gy >= f.band_g0 && gy < f.band_g1;
  const uint32_t g = gy * f.xsg + gx;
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  const uint32_t gw = min(32u, f.xsb - bx0), gh = min(32u, f.ysb - by0);
  const bool valid = tid < gw * gh;
  const uint32_t by = valid ? tid / gw : 0, bx = valid ? tid % gw : 0;
  const uint32_t aby = by0 + by, abx = bx0 + bx;
  const uint32_t raw = valid ? f.acs[(size_t)aby * f.xsb + abx] : 0;
  // side info of the cell, fetched up front: these loads overlap the scan
  // instead of starting after the two barriers
  const size_t cell = (size_t)aby * f.xsb + abx;
  const size_t tile = (size_t)(aby >> 3) * f.xtiles + (abx >> 3);
  const int cell_q = valid ? f.raw_quant[cell] : 1;
  const uint32_t cell_cfl = valid ? (((uint32_t)(uint8_t)f.ytox[tile] << 16) | ((uint32_t)(uint8_t)f.ytob[tile] << 24)) : 0;
  const uint32_t cell_sharp = (with_sigma && valid) ? f.sharp[cell] : 0;
  bool first = raw & 1;
  uint32_t s = raw >> 1;
  bool bad = false;
  static_assert(kCountStride == 1024, "one counter per thread");
  if (f.zero_counts && blockIdx.x == 0) f.zero_counts[tid] = 0;
  if (s >= JXLHIP_NUM_STRATEGIES) {
    bad = valid;
    s = 0;
    first = false;
  }
  const uint32_t sw = kStrategyWords.v[s];
  const uint32_t cx = sw & 0xffu, cy = (sw >> 8) & 0xffu;
  if (first && (bx + cx > gw || by + cy > gh)) {
    bad = true;
    first = false;
  }
  // a strategy the caller's used_acs mask rules out would never be decoded: report it
  if (first && f.used_acs && !((f.used_acs >> s) & 1u)) bad = true;
  const uint32_t n64 = first ? cx * cy : 0;
  // exclusive scan of n64 over the 1024 threads
  uint32_t incl = n64;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if ((int)lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  // per-wave class histogram.  Fused mode: DCT8 varblocks are decoded by the fused kernel from cell_info and stay off
  // the work list -- except, in a STRIPE (fused == 2), those of the stripe's first / last block row next to a
  // neighbouring stripe: they are decoded into the planes as well (the halo rows the neighbour pulls).
  const int cls_frame = first ? (int)(int8_t)(sw >> 16) : -1;
  const bool edge_row = f.fused == 2 && ((aby == (f.y0 >> 3) && f.group_y0 > 0) ||
                                         (aby == ((f.y1 - 1) >> 3) && f.group_y0 + f.group_rows < f.ysg));
  const int cls = (f.fused && cls_frame == kClsDct8 && !edge_row) ? -1 : cls_frame;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t rank_in_wave = 0;
#pragma unroll
  for (int c = 0; c < kNumClasses; c++) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) rank_in_wave = (uint32_t)__builtin_popcountll(m & lt);
    if (lane == 0) wave_cls[wave][c] = (uint16_t)__builtin_popcountll(m);
  }
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const uint32_t t = wave_tot[w];
    if (w < (int)wave) base += t;
    total += t;
  }
  const uint32_t off64 = base + incl - n64;
  if (total > 1024) bad = true;  // would overflow the group's 65536-coefficient stream
  if (in_stripe && __any(bad)) {
    if (bad) atomicOr(f.error_flag, 1);
  }
  const bool group_ok = total <= 1024;
  // reserve list ranges: one global atomic per class and workgroup
  if (tid < kNumClasses) {
    uint32_t n = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) n += wave_cls[w][tid];
    wg_base[tid] = (n && in_stripe && group_ok) ? atomicAdd(&wl.count[tid * kCounterPad], n) : 0;
  }
  // sigma_quant of each varblock, scattered to the cells it covers
  if (with_sigma && first) {
    const float kInvSigmaNum = -1.1715728752538099024f;
    const float sigma_quant = epf_quant_mul / (f.quant_scale * (float)cell_q * kInvSigmaNum);
    for (uint32_t iy = 0; iy < cy; iy++)
      for (uint32_t ix = 0; ix < cx; ix++) cell_sq[(by + iy) * gw + bx + ix] = sigma_quant;
  }
  __syncthreads();
  // whole frame through the fused kernel: EVERY cell says what it is (no memset of the table in front of this kernel)
  if (f.fused == 1 && valid) {
    const bool own = in_stripe && group_ok && cls_frame == kClsDct8;
    f.cell_info[cell] = own ? make_uint2(g * f.coef_stride64 + off64, ((uint32_t)cell_q & 0xffffu) | cell_cfl) : make_uint2(kCellFromPlanes, 0u);
  }
  if (in_stripe && group_ok && cls_frame >= 0) {
    WorkItem it;
    it.pos = (aby << 16) | abx;
    it.off = g * f.coef_stride64 + off64;
    it.qc = ((uint32_t)cell_q & 0xffffu) | cell_cfl;
    it.pad = 0;
    if (f.fused == 2 && cls_frame == kClsDct8) f.cell_info[cell] = make_uint2(it.off, it.qc);  // (a stripe: the rest keeps its 0xFF fill)
    if (cls >= 0) {
      uint32_t pos = wg_base[cls] + rank_in_wave;
      for (uint32_t w = 0; w < wave; w++) pos += wave_cls[w][cls];
      wl.list[cls][pos] = it;
    }
  }
  // ComputeSigma (epf.cc:69-79), one cell per thread
  if (with_sigma && valid) {
    float sharp_mul = lut.v[0];  // (a select chain: a dynamic index into the kernel argument would put the table in scratch)
#pragma unroll
    for (int i = 1; i < 8; i++) sharp_mul = (cell_sharp & 7u) == (uint32_t)i ? lut.v[i] : sharp_mul;
    float sigma = cell_sq[tid] * sharp_mul;
    sigma = sigma < -1e-4f ? sigma : -1e-4f;
    f.inv_sigma[cell] = 1.0f / sigma;
  }
}

// ------------------------------------------------- single-block transforms
// TransformToPixels for the one-block strategies, all in registers
// (dec_transforms-inl.h:463-581, 399-454).  co: 64 coefficients (co[0] = DC
// already inserted), px: 8x8 pixels row-major.
template <int S>
__device__ __forceinline__ void Idct2TopT(float* b) {
  constexpr int H = S / 2;
  float t[S * S];
#pragma unroll
  for (int y = 0; y < H; y++)
#pragma unroll
    for (int x = 0; x < H; x++) {
      const float c00 = b[y * 8 + x], c01 = b[y * 8 + H + x];
      const float c10 = b[(y + H) * 8 + x], c11 = b[(y + H) * 8 + H + x];
      t[(y * 2) * S + x * 2] = c00 + c01 + c10 + c11;
      t[(y * 2) * S + x * 2 + 1] = c00 + c01 - c10 - c11;
      t[(y * 2 + 1) * S + x * 2] = c00 - c01 + c10 - c11;
      t[(y * 2 + 1) * S + x * 2 + 1] = c00 - c01 - c10 + c11;
    }
#pragma unroll
  for (int y = 0; y < S; y++)
#pragma unroll
    for (int x = 0; x < S; x++) b[y * 8 + x] = t[y * S + x];
}

template <int KIND>
__device__ __forceinline__ void AfvToPixels(const float* co, float* px) {
  constexpr int afv_x = KIND & 1, afv_y = KIND / 2;
  const float b00 = co[0], b01 = co[1], b10 = co[8];
  const float dc0 = (b00 + b10 + b01) * 4.0f;
  const float dc1 = (b00 + b10 - b01);
  const float dc2 = b00 - b10;
  float coeff[16], block[32], out[32];
  coeff[0] = dc0;
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 4; ix++)
      if (ix | iy) coeff[iy * 4 + ix] = co[iy * 2 * 8 + ix * 2];
  // AFVIDCT4x4: pixel[i] = sum_j coeff[j] * basis[j][i]
#pragma unroll
  for (int i = 0; i < 16; i++) {
    float p = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) p = __builtin_fmaf(coeff[j], kAfvBasis[j * 16 + i], p);
    block[i] = p;
  }
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 4; ix++)
      px[(iy + afv_y * 4) * 8 + afv_x * 4 + ix] =
          block[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)];
  block[0] = dc1;
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 4; ix++)
      if (ix | iy) block[iy * 4 + ix] = co[iy * 2 * 8 + ix * 2 + 1];
  Idct2dReg<4, 4>(block, out);
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 4; ix++)
      px[(afv_y * 4 + iy) * 8 + (afv_x == 1 ? 0 : 4) + ix] = out[iy * 4 + ix];
  block[0] = dc2;
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 8; ix++)
      if (ix | iy) block[iy * 8 + ix] = co[(1 + iy * 2) * 8 + ix];
  Idct2dReg<4, 8>(block, out);
#pragma unroll
  for (int iy = 0; iy < 4; iy++)
#pragma unroll
    for (int ix = 0; ix < 8; ix++)
      px[((afv_y == 1 ? 0 : 4) + iy) * 8 + ix] = out[iy * 8 + ix];
}

template <int STRATEGY>
__device__ __forceinline__ void Transform64(const float* co, float* px) {
  if constexpr (STRATEGY == 0) {
    Idct2dReg<8, 8>(co, px);
  } else if constexpr (STRATEGY == 1) {  // IDENTITY
    const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
    const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11,
                          b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int x = 0; x < 2; x++) {
        float residual_sum = 0;
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++)
            if (ix | iy) residual_sum += co[(y + iy * 2) * 8 + x + ix * 2];
        const float base = dcs[y * 2 + x] - residual_sum * (1.0f / 16);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) {
            if (ix == 1 && iy == 1) continue;
            px[(y * 4 + iy) * 8 + x * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2] + base;
          }
        px[(4 * y + 1) * 8 + 4 * x + 1] = base;
        px[(y * 4) * 8 + x * 4] = co[(y + 2) * 8 + x + 2] + base;
      }
  } else if constexpr (STRATEGY == 2) {  // DCT2X2
    float c[64];
#pragma unroll
    for (int i = 0; i < 64; i++) c[i] = co[i];
    Idct2TopT<2>(c);
    Idct2TopT<4>(c);
    Idct2TopT<8>(c);
#pragma unroll
    for (int i = 0; i < 64; i++) px[i] = c[i];
  } else if constexpr (STRATEGY == 3) {  // DCT4X4
    const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
    const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11,
                          b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int x = 0; x < 2; x++) {
        float block[16], out[16];
        block[0] = dcs[y * 2 + x];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++)
            if (ix | iy) block[iy * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2];
        Idct2dReg<4, 4>(block, out);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) px[(y * 4 + iy) * 8 + x * 4 + ix] = out[iy * 4 + ix];
      }
  } else if constexpr (STRATEGY == 12) {  // DCT4X8: two 4-row halves
    const float b0 = co[0], b1 = co[8];
    const float dcs[2] = {b0 + b1, b0 - b1};
#pragma unroll
    for (int y = 0; y < 2; y++) {
      float block[32], out[32];
      block[0] = dcs[y];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++)
          if (ix | iy) block[iy * 8 + ix] = co[(y + iy * 2) * 8 + ix];
      Idct2dReg<4, 8>(block, out);
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++) px[(y * 4 + iy) * 8 + ix] = out[iy * 8 + ix];
    }
  } else if constexpr (STRATEGY == 13) {  // DCT8X4: two 4-column halves
    const float b0 = co[0], b1 = co[8];
    const float dcs[2] = {b0 + b1, b0 - b1};
#pragma unroll
    for (int x = 0; x < 2; x++) {
      float block[32], out[32];
      block[0] = dcs[x];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++)
          if (ix | iy) block[iy * 8 + ix] = co[(x + iy * 2) * 8 + ix];
      Idct2dReg<8, 4>(block, out);
#pragma unroll
      for (int iy = 0; iy < 8; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) px[iy * 8 + x * 4 + ix] = out[iy * 4 + ix];
    }
  } else {
    AfvToPixels<STRATEGY - 14>(co, px);
  }
}

struct TabOffsets {
  uint32_t v[27];
};
__host__ __device__ constexpr TabOffsets MakeTabOffsets() {
  TabOffsets t{};
  for (int s = 0; s < 27; s++) t.v[s] = DequantOffset(s);
  return t;
}
static constexpr TabOffsets kSingleTabOffset = MakeTabOffsets();

template <typename CT>
struct Dct8Geom {
  static constexpr int kSteps = sizeof(CT) == 2 ? 4 : 2;  // steps of 8 blocks per wave
  static constexpr uint32_t kPerWg = 4 * kSteps * 8;      // blocks per 256-thread workgroup
};

// workgroup `wg` of the DCT8 list (n entries)
template <typename CT>
__device__ __forceinline__ void Dct8Rows(const DevFrame& f, const WorkItem* __restrict__ list, uint32_t n,
                                         uint32_t wg) {
  constexpr int kSteps = Dct8Geom<CT>::kSteps;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t first = (wg * 4 + wave) * (kSteps * 8);
  if (first >= n) return;
  const int j = lane >> 3;  // matrix row (input), pixel row (output)
  const bool bit3 = (lane & 8) != 0;
  // this lane's 8 entries of the three dequant matrices (DequantLane, dec_group.cc:115-153)
  float tab[3][8];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float4 t0 = *(const float4*)(f.dequant + c * 64 + j * 8);
    const float4 t1 = *(const float4*)(f.dequant + c * 64 + j * 8 + 4);
    tab[c][0] = t0.x, tab[c][1] = t0.y, tab[c][2] = t0.z, tab[c][3] = t0.w;
    tab[c][4] = t1.x, tab[c][5] = t1.y, tab[c][6] = t1.z, tab[c][7] = t1.w;
  }
  WorkItem it[kSteps];
  Dct8Row<CT> rows[kSteps][3];
  float dcv[kSteps][3];
  bool valid[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const uint32_t b = first + s * 8 + (lane & 7);
    valid[s] = b < n;
    it[s] = list[valid[s] ? b : n - 1];
  }
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const size_t elem = (size_t)it[s].off * 64u + (size_t)j * 8u;
    const size_t cell = (size_t)(it[s].pos >> 16) * f.xsb + (it[s].pos & 0xffffu);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      rows[s][c].Load(f.coeffs[c], elem);
      dcv[s][c] = f.dc[c][cell];
    }
  }
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const BlockHdr h = MakeHdr(f, it[s]);
    int32_t q[8];
    float vy[8];
    rows[s][1].Unpack(q);
#pragma unroll
    for (int k = 0; k < 8; k++) vy[k] = AdjustQuantBias(q[k], f.biases[1], f.biases[3]) * (tab[1][k] * h.sy);
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
      float v[8];
      if (c == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = vy[k];
      } else {
        const float sc = c == 0 ? h.sx : h.sb;
        const float cc = c == 0 ? h.x_cc : h.b_cc;
        rows[s][c].Unpack(q);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const float d = AdjustQuantBias(q[k], f.biases[c], f.biases[3]) * (tab[c][k] * sc);
          v[k] = __builtin_fmaf(cc, vy[k], d);
        }
      }
      if (j == 0) v[0] = dcv[s][c];
      IdctReg<8>(v);
      Transpose8Lanes(v, bit3);
      IdctReg<8>(v);
      if (valid[s]) {
        float* dst = TilePtr(f, c, it[s].pos >> 16, it[s].pos & 0xffffu) + j * 8;
        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
}

// One special 8x8 kind (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3), one channel, 64
// blocks: lane = block, everything in registers, the block's coefficient line and its tile moved
// by the lane itself (16-byte pieces) -- no LDS, so these tasks can share a kernel (and its
// occupancy) with the DCT8 rows.  A unit of 64 blocks of one of these kinds takes ~25 us from
// first load to last store whatever the form; as tasks at the head of the DCT8 launch that
// latency disappears behind the DCT8 bulk instead of being a launch of its own.
template <typename CT>
__device__ __forceinline__ void SpecialTask(const DevFrame& f, int strategy, const WorkItem* __restrict__ list,
                                            uint32_t first, uint32_t n, int c) {
  constexpr int kVec = 64 * (int)sizeof(CT) / 16;
  const int lane = threadIdx.x & 63;
  const uint32_t idx = first + lane;
  const bool valid = idx < n;
  const WorkItem it = list[valid ? idx : n - 1];
  const BlockHdr h = MakeHdr(f, it);
  const float* __restrict__ tab = f.dequant + kSingleTabOffset.v[strategy];
  const float dcv = f.dc[c][(size_t)h.aby * f.xsb + h.abx];
  auto unpack = [&](const uint4 r, int32_t* q) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    if constexpr (sizeof(CT) == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        q[2 * k] = (int32_t)(int16_t)(w[k] & 0xffffu);
        q[2 * k + 1] = (int32_t)w[k] >> 16;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) q[k] = (int32_t)w[k];
    }
  };
  constexpr int kPer = 16 / (int)sizeof(CT);  // coefficients per 16-byte piece
  float v[64], px[64];
  {
    const uint4* p = (const uint4*)((const CT*)f.coeffs[1] + h.coef);
    uint4 raw[kVec];
#pragma unroll
    for (int i = 0; i < kVec; i++) raw[i] = p[i];
#pragma unroll
    for (int i = 0; i < kVec; i++) {
      int32_t q[kPer];
      unpack(raw[i], q);
#pragma unroll
      for (int k = 0; k < kPer; k++)
        v[i * kPer + k] = AdjustQuantBias(q[k], f.biases[1], f.biases[3]) * (tab[64 + i * kPer + k] * h.sy);
    }
  }
  if (c != 1) {  // wave-uniform
    const float sc = c == 0 ? h.sx : h.sb;
    const float cc = c == 0 ? h.x_cc : h.b_cc;
    const float bias = c == 0 ? f.biases[0] : f.biases[2];
    const float* __restrict__ tc = tab + c * 64;
    const uint4* p = (const uint4*)((const CT*)f.coeffs[c] + h.coef);
    uint4 raw[kVec];
#pragma unroll
    for (int i = 0; i < kVec; i++) raw[i] = p[i];
#pragma unroll
    for (int i = 0; i < kVec; i++) {
      int32_t q[kPer];
      unpack(raw[i], q);
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const float d = AdjustQuantBias(q[k], bias, f.biases[3]) * (tc[i * kPer + k] * sc);
        v[i * kPer + k] = __builtin_fmaf(cc, v[i * kPer + k], d);
      }
    }
  }
  v[0] = dcv;
  switch (strategy) {  // wave-uniform
    case 1: Transform64<1>(v, px); break;
    case 2: Transform64<2>(v, px); break;
    case 3: Transform64<3>(v, px); break;
    case 12: Transform64<12>(v, px); break;
    case 13: Transform64<13>(v, px); break;
    case 14: Transform64<14>(v, px); break;
    case 15: Transform64<15>(v, px); break;
    case 16: Transform64<16>(v, px); break;
    default: Transform64<17>(v, px); break;
  }
  if (valid) {
    float4* dst = (float4*)TilePtr(f, c, h.aby, h.abx);
#pragma unroll
    for (int i = 0; i < 16; i++) dst[i] = make_float4(px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3]);
  }
}

// All single-block strategies in one launch: first the (unit, channel) tasks of the nine special
// kinds, four per workgroup, then the DCT8 rows.
// The (unit, channel) tasks of the nine special kinds, four per workgroup: workgroup `wg` of them.
// Returns the number of workgroups the tasks fill (wg >= that: nothing done).
template <typename CT>
__device__ __forceinline__ uint32_t SpecialWorkgroup(const DevFrame& f, const WorkLists& wl, uint32_t wg) {
  uint32_t cnt[kNumSpecial];
  uint32_t tasks = 0;
#pragma unroll
  for (int i = 0; i < kNumSpecial; i++) {
    cnt[i] = wl.count[(kClsSpecial0 + i) * kCounterPad];
    tasks += 3 * ((cnt[i] + 63) / 64);
  }
  const uint32_t special_wgs = (tasks + 3) / 4;
  if (wg >= special_wgs) return special_wgs;
  const uint32_t task = wg * 4 + (threadIdx.x >> 6);
  if (task >= tasks) return special_wgs;
  uint32_t base = 0;
  int cls = -1;
  uint32_t first = 0, n = 0, chan = 0;
#pragma unroll
  for (int i = 0; i < kNumSpecial; i++) {
    const uint32_t t = 3 * ((cnt[i] + 63) / 64);
    if (cls < 0 && task < base + t) {
      cls = i;
      first = ((task - base) / 3) * 64;
      chan = (task - base) % 3;
      n = cnt[i];
    }
    base += t;
  }
  SpecialTask<CT>(f, (int)kSpecialStrategy[cls], wl.list[kClsSpecial0 + cls], first, n,
                  chan == 0 ? 1 : (chan == 1 ? 0 : 2));
  return special_wgs;
}

template <typename CT>
__global__ __launch_bounds__(256, 3) void k_transform_8(DevFrame f, WorkLists wl) {
  const uint32_t special_wgs = SpecialWorkgroup<CT>(f, wl, blockIdx.x);
  if (blockIdx.x >= special_wgs)
    Dct8Rows<CT>(f, wl.list[kClsDct8], wl.count[kClsDct8 * kCounterPad], blockIdx.x - special_wgs);
}

// --------------------------------------------------------------- k_rowlane
// The row-per-lane scheme of k_dct8 for the separable DCTs with sides 8..32 (16x8 .. 32x32,
// ~43 % of a d1.0 frame).  The stored coefficient matrix is S x L (S = shorter side, rows of
// L = longer side contiguous coefficients): S lanes share a varblock,
//   lane = varblock-of-the-step (low bits) | matrix row j (high log2(S) bits)
// and each lane loads, dequantises and transforms its row; the other dimension is reached by
// transposing S x S register tiles across the S lanes (one exchange primitive per lane bit:
// v_permlane32_swap, v_permlane16_swap, DPP row_ror:8, row_ror:4/12, quad_perm).  No LDS.
//   R <  C (8x16): pass 1 along the row (as IDCT2D), transpose, pass 2, transpose back
//   R >= C       : the long pass first (in the lane), one transpose, the short pass -- the
//                  opposite order of IDCT2D (rounding-level difference, see k_dct8)
// Either way a lane ends up with whole pixel rows: 16-byte stores into 8x8 tiles.
template <int BIT>
__device__ __forceinline__ void ExchangePair(float& a, float& b, int lane) {
  if constexpr (BIT == 5) {
    SwapHalves32(a, b);
  } else if constexpr (BIT == 4) {
    SwapRows16(a, b);
  } else {
    const bool set = (lane >> BIT) & 1;
    const float send = set ? a : b;
    float recv;
    if constexpr (BIT == 3) {
      recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));
    } else if constexpr (BIT == 2) {
      // row_ror:4 delivers lane i-4, row_ror:12 lane i+4 (mod 16)
      const float lo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x124, 0xf, 0xf, false));
      const float hi = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x12c, 0xf, 0xf, false));
      recv = set ? lo : hi;
    } else {
      static_assert(BIT == 1, "lane bits 1..5");
      // quad_perm [2,3,0,1]: lane i ^ 2
      recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4e, 0xf, 0xf, false));
    }
    a = set ? recv : a;
    b = set ? b : recv;
  }
}

// element (row j, register a) -> (row a, register j) of an S x S tile held by S lanes
template <int S>
__device__ __forceinline__ void TransposeTile(float* w, int lane) {
  constexpr int kLog = S == 8 ? 3 : (S == 16 ? 4 : 5);
  constexpr int kLb0 = 6 - kLog;
#pragma unroll
  for (int t = kLog - 1; t >= 0; t--) {
#pragma unroll
    for (int k = 0; k < S; k++) {
      if (k & (1 << t)) continue;
      if (kLb0 + t == 5) ExchangePair<5>(w[k], w[k | (1 << t)], lane);
      else if (kLb0 + t == 4) ExchangePair<4>(w[k], w[k | (1 << t)], lane);
      else if (kLb0 + t == 3) ExchangePair<3>(w[k], w[k | (1 << t)], lane);
      else if (kLb0 + t == 2) ExchangePair<2>(w[k], w[k | (1 << t)], lane);
      else ExchangePair<1>(w[k], w[k | (1 << t)], lane);
    }
  }
}

// The same transposition through LDS (round 6), for the units that run inside the merged k_transform_r -- whose
// workgroups carry family A's 50 KB allocation whether they use it or not: lane (varblock b, row j) writes its S values
// as row j of a padded S x (S + 1) tile and reads column j back.  Pure data movement (bit-identical); 2 S LDS
// instructions instead of the exchange network's ~8 S VALU instructions and its send / receive temporaries -- the
// 32-point classes no longer need more than the kernel's 168 registers.  Row stride S + 1: the writes of a wave
// instruction (lanes = rows) and its reads (lanes = columns) fall on distinct banks.  The tile belongs to ONE wave
// (a unit's varblocks are split by wave): LDS operations of a wave execute in order, no barrier.
typedef __attribute__((address_space(3))) float LdsTile;
template <int S>
__device__ __forceinline__ void TransposeTileLds(float* w, int b, int j, LdsTile* wave_lds) {
  LdsTile* tile = wave_lds + b * (S * (S + 1));
  LdsTile* row = tile + j * (S + 1);
#pragma unroll
  for (int a = 0; a < S; a++) row[a] = w[a];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int a = 0; a < S; a++) w[a] = tile[a * (S + 1) + j];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by the next transposition)
}
// bytes of LDS a workgroup of four waves needs for it
template <int S>
constexpr int TransposeLdsBytes() { return 4 * (64 / S) * S * (S + 1) * 4; }

template <int R, int C, int STRATEGY, typename CT, bool LDS_T = false>
__device__ __forceinline__ void RowLaneUnit(const DevFrame& f, const WorkItem* __restrict__ list,
                                            uint32_t first, uint32_t n, unsigned char* smem = nullptr) {
  constexpr int S = R < C ? R : C, L = R < C ? C : R;
  constexpr int BPS = 64 / S;  // varblocks per wave
  constexpr int CY = R / 8, CX = C / 8;
  constexpr int kTiles = L / S;
  constexpr int kVec = L * (int)sizeof(CT) / 16;  // 16-byte loads per row and channel
  const int tid = Tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane / BPS;
  const uint32_t vb = first + wave * BPS + (lane & (BPS - 1));
  if (first + wave * BPS >= n) return;
  const bool valid = vb < n;
  const WorkItem it = list[valid ? vb : n - 1];
  const BlockHdr h = MakeHdr(f, it);
  // The lane's coefficient rows, one channel per register set and at most TWO sets alive: Y and X are requested up front,
  // B when Y's set has been unpacked -- its loads travel during the Y and X transforms.  (Rounds 1-5 requested all three
  // up front: 48 VGPRs of raw rows for the 32-point classes, which pushed the merged k_transform_r over its 168 registers:
  // 148 spilled VGPRs, and the spill traffic reached HBM -- 8K frames of DCT32X32 / DCT32X8 alone read 1.44-1.47x and wrote
  // 1.21-1.27x their bytes, profiles/r06_transform_overread.txt.)
  uint4 raw_a[kVec], raw_b[kVec];
  auto request = [&](int c, uint4* r) {
    const uint4* p = (const uint4*)((const CT*)f.coeffs[c] + h.coef + (size_t)j * L);
#pragma unroll
    for (int i = 0; i < kVec; i++) r[i] = p[i];
  };
  request(1, raw_a);
  request(0, raw_b);
  // lowest frequencies from the DC patch (LowestFrequenciesFromDC, dec_transforms-inl.h:691-818):
  // CY-point DCTs down the columns, CX-point DCTs along the rows, resampling scales; the lanes
  // holding the LLF corner compute the whole (at most 2x2) patch and keep their entries
  constexpr int kLlfLanes = CY < CX ? CY : CX;
  constexpr int kLlfRegs = CY < CX ? CX : CY;
  // (per channel, when the channel is transformed: three channels' patches held from the start cost the 32-point classes
  // twelve registers they do not have)
  auto llf_of = [&](int c, float* out) {
    const float* dc = f.dc[c] + (size_t)h.aby * f.xsb + h.abx;
    float dp[CY][CX];
#pragma unroll
    for (int x = 0; x < CX; x++) {
      float v[CY];
#pragma unroll
      for (int y = 0; y < CY; y++) v[y] = dc[(size_t)y * f.xsb + x];
      DctReg<CY>(v);
#pragma unroll
      for (int y = 0; y < CY; y++) dp[y][x] = (1.0f / CY) * v[y];
    }
#pragma unroll
    for (int y = 0; y < CY; y++) {
      float v[CX];
#pragma unroll
      for (int x = 0; x < CX; x++) v[x] = dp[y][x];
      DctReg<CX>(v);
      const float ry = kResampleUpHost[CY + y];
#pragma unroll
      for (int x = 0; x < CX; x++) {
        const float val = (1.0f / CX) * v[x];
        if constexpr (CY < CX) {
          if (j == y) out[x] = val * ry * kResampleUpHost[CX + x];
        } else {
          if (j == x) out[y] = val * kResampleUpHost[CX + x] * ry;
        }
      }
    }
  };
  const float* __restrict__ tab = f.dequant + DequantOffset(STRATEGY) + j * L;
  // one 16-byte piece of a row -> its coefficients as integers (8 of 16 bits, 4 of 32 bits)
  constexpr int kPer = 16 / (int)sizeof(CT);
  auto unpack_piece = [&](const uint4 r, int32_t* q) {
    if constexpr (sizeof(CT) == 2) {
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        q[2 * k] = (int32_t)(int16_t)(w[k] & 0xffffu);
        q[2 * k + 1] = (int32_t)w[k] >> 16;
      }
    } else {
      q[0] = (int32_t)r.x, q[1] = (int32_t)r.y, q[2] = (int32_t)r.z, q[3] = (int32_t)r.w;
    }
  };
  // Dequantisation piece by piece (DequantLane, dec_group.cc:115-153): the 32-point classes fence the pieces off from
  // each other -- left alone the scheduler unpacks a whole row and requests its whole table row first (32 + 32 registers
  // beside the 32 of vy, the 32 being produced and the next channel's pending rows: over the kernel's 168)
  float vy[L];
#pragma unroll
  for (int i = 0; i < kVec; i++) {
    int32_t q[kPer];
    unpack_piece(raw_a[i], q);
#pragma unroll
    for (int k = 0; k < kPer; k++)
      vy[i * kPer + k] = AdjustQuantBias(q[k], f.biases[1], f.biases[3]) * (tab[R * C + i * kPer + k] * h.sy);
    if constexpr (L > 16) __builtin_amdgcn_sched_barrier(0);
  }
  request(2, raw_a);  // B: in flight while Y and X are transformed
#pragma unroll
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    float v[L];
    if (c == 1) {
#pragma unroll
      for (int k = 0; k < L; k++) v[k] = vy[k];
    } else {
      const float sc = c == 0 ? h.sx : h.sb;
      const float cc = c == 0 ? h.x_cc : h.b_cc;
#pragma unroll
      for (int i = 0; i < kVec; i++) {
        int32_t q[kPer];
        unpack_piece(c == 0 ? raw_b[i] : raw_a[i], q);
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          const float d = AdjustQuantBias(q[k], f.biases[c], f.biases[3]) * (tab[c * R * C + i * kPer + k] * sc);
          v[i * kPer + k] = __builtin_fmaf(cc, vy[i * kPer + k], d);
        }
        if constexpr (L > 16) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (j < kLlfLanes) {
      float llf[kLlfRegs];
      llf_of(c, llf);
#pragma unroll
      for (int k = 0; k < kLlfRegs; k++) v[k] = llf[k];
    }
    IdctReg<L>(v);
#pragma unroll
    for (int t = 0; t < kTiles; t++) {
      if constexpr (LDS_T) TransposeTileLds<S>(v + t * S, lane & (BPS - 1), j, (LdsTile*)smem + wave * (BPS * S * (S + 1)));
      else TransposeTile<S>(v + t * S, lane);
    }
#pragma unroll
    for (int t = 0; t < kTiles; t++) IdctReg<S>(v + t * S);
    if constexpr (R < C) {
      // v[t*S + a] = pixel (a, t*S + j): back to rows
#pragma unroll
      for (int t = 0; t < kTiles; t++) {
        if constexpr (LDS_T) TransposeTileLds<S>(v + t * S, lane & (BPS - 1), j, (LdsTile*)smem + wave * (BPS * S * (S + 1)));
        else TransposeTile<S>(v + t * S, lane);
      }
      // lane j = pixel row j (R = S <= 8 rows... or 16), all C columns
      if (valid) {
#pragma unroll
        for (int g = 0; g < C / 8; g++) {
          float* dst = TilePtr(f, c, h.aby + (j >> 3), h.abx + g) + (j & 7) * 8;
          *(float4*)dst = make_float4(v[g * 8], v[g * 8 + 1], v[g * 8 + 2], v[g * 8 + 3]);
          *(float4*)(dst + 4) = make_float4(v[g * 8 + 4], v[g * 8 + 5], v[g * 8 + 6], v[g * 8 + 7]);
        }
      }
    } else {
      // tile t: pixel row a = t*S + j, its C = S columns in v[t*S ..]
      if (valid) {
#pragma unroll
        for (int t = 0; t < kTiles; t++) {
          const int a = t * S + j;
#pragma unroll
          for (int g = 0; g < C / 8; g++) {
            float* dst = TilePtr(f, c, h.aby + (a >> 3), h.abx + g) + (a & 7) * 8;
            const float* src = v + t * S + g * 8;
            *(float4*)dst = make_float4(src[0], src[1], src[2], src[3]);
            *(float4*)(dst + 4) = make_float4(src[4], src[5], src[6], src[7]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ k_medium
// compile-time-sized select from the resample table with a runtime index
template <int N>
__device__ __forceinline__ float ResampleUpSel(int i) {
  float r = kResampleUpHost[N];
#pragma unroll
  for (int j = 1; j < N; j++) r = (i == j) ? kResampleUpHost[N + j] : r;
  return r;
}

// R x C pixel varblocks (R rows tall, C cols wide), 16x8 .. 64x64.
// Workgroup = 3 waves, wave w handles channel w in the transform passes; each
// 1-D transform lives in one lane's registers; LDS holds the coefficient
// matrix (padded rows) between the passes.
//   dequant + CfL  : all threads, 4 coefficients per step (DequantLane)
//   LLF <- DC      : LowestFrequenciesFromDC via ReinterpretingDCT
//                    (dec_transforms-inl.h:35-64,691-818): CX lanes do the
//                    vertical CY-point DCTs, then CY lanes the horizontal ones
//   pass 1 / pass 2: ComputeScaledIDCT (dct-inl.h:376-397): R lanes run the
//                    C-point IDCT of one row of frequencies, then C lanes the
//                    R-point IDCT of one pixel column and store it
template <int R, int C>
struct MediumGeom {
  static constexpr int S = R < C ? R : C, L = R < C ? C : R;
  static constexpr int ML = L;        // lanes per varblock and channel
  static constexpr int NB = 64 / ML;  // varblocks per batch
  static constexpr int LP = L + 1;    // coefficient matrix row stride
  static constexpr int TP = C + 1;    // intermediate T[u][x] / pixel row stride
  static constexpr int BUF = (S * LP > R * TP ? S * LP : R * TP);
  static constexpr int CY = R / 8, CX = C / 8;
  static constexpr int kUnitVarblocks = (64 / (CY * CX)) > NB ? 64 / (CY * CX) : NB;  // 64 blocks of area
  static constexpr int kHdrOffset = (12 * NB * (BUF + CY * CX) + 15) & ~15;
  // pipelined classes (MediumLoads) keep their dequant table in LDS behind the headers
  static constexpr bool kPipelined = (NB * R * C + 767) / 768 <= 2;
  static constexpr int kTabOffset = kHdrOffset + kUnitVarblocks * 48;
  static constexpr int kLdsBytes = kTabOffset + (kPipelined ? 12 * R * C : 0);
};

// What one thread fetches from global memory for one batch: its share of the
// quantized coefficients (4 per step and channel) and, for the few lanes that
// start LowestFrequenciesFromDC, a column of DC values.  Kept in registers so
// that the NEXT batch's loads are in flight while the current one is decoded
// (classes with few steps per batch only; the 64-point classes load in place).
template <int R, int C, typename CT>
struct MediumLoads {
  using G = MediumGeom<R, C>;
  static constexpr int SIZE = R * C;
  static constexpr int kSteps = (G::NB * SIZE + 767) / 768;
  static constexpr bool kPipelined = G::kPipelined;
  using Raw = typename std::conditional<sizeof(CT) == 2, uint2, int4>::type;
  Raw x[kSteps], y[kSteps], b[kSteps];
  float dc[G::CY];
};

template <typename CT>
__device__ __forceinline__ void UnpackCoeffs(const uint2 v, int32_t* q) {
  q[0] = (int16_t)(v.x & 0xffff);
  q[1] = (int32_t)v.x >> 16;
  q[2] = (int16_t)(v.y & 0xffff);
  q[3] = (int32_t)v.y >> 16;
}
template <typename CT>
__device__ __forceinline__ void UnpackCoeffs(const int4 v, int32_t* q) {
  q[0] = v.x;
  q[1] = v.y;
  q[2] = v.z;
  q[3] = v.w;
}

template <int R, int C, typename CT>
__device__ __forceinline__ void MediumFetch(const DevFrame& f, const BlockHdr* hdr, int nb,
                                            MediumLoads<R, C, CT>& ld) {
  using G = MediumGeom<R, C>;
  using LD = MediumLoads<R, C, CT>;
  using Raw = typename LD::Raw;
  const int tid = Tid();
  const int c = tid >> 6, lane = tid & 63;
  const int b = lane / G::ML, i = lane % G::ML;
  if (b < nb && i < G::CX) {
    const BlockHdr& h = hdr[b];
    const float* dc = f.dc[c] + (size_t)h.aby * f.xsb + h.abx + i;
#pragma unroll
    for (int y = 0; y < G::CY; y++) ld.dc[y] = dc[(size_t)y * f.xsb];
  }
#pragma unroll
  for (int it = 0; it < LD::kSteps; it++) {
    const int k4 = tid * 4 + it * 768;
    if (k4 < nb * LD::SIZE) {
      const int vb = k4 / LD::SIZE, k = k4 % LD::SIZE;
      const size_t at = hdr[vb].coef + k;
      ld.x[it] = *(const Raw*)((const CT*)f.coeffs[0] + at);
      ld.y[it] = *(const Raw*)((const CT*)f.coeffs[1] + at);
      ld.b[it] = *(const Raw*)((const CT*)f.coeffs[2] + at);
    }
  }
}

// the three dequant-table vectors of one step of one thread
struct TabStep {
  float4 x, y, b;
};

// One batch of NB varblocks whose headers are hdr[0..nb) (already in LDS).
template <int R, int C, int STRATEGY, typename CT>
__device__ __forceinline__ void MediumBatch(const DevFrame& f, const BlockHdr* hdr, int nb,
                                            const MediumLoads<R, C, CT>& ld, unsigned char* smem) {
  using G = MediumGeom<R, C>;
  using LD = MediumLoads<R, C, CT>;
  constexpr int L = G::L, ML = G::ML, NB = G::NB, LP = G::LP, TP = G::TP, BUF = G::BUF;
  constexpr int CY = G::CY, CX = G::CX;
  constexpr int SIZE = R * C;
  constexpr uint32_t kTab = DequantOffset(STRATEGY);
  float(*buf)[NB][BUF] = reinterpret_cast<float(*)[NB][BUF]>(smem);
  float(*dcp)[NB][CY * CX] = reinterpret_cast<float(*)[NB][CY * CX]>(smem + 12 * NB * BUF);
  const int tid = Tid();

  const int c = tid >> 6, lane = tid & 63;
  const int b = lane / ML, i = lane % ML;
  const bool active = b < nb;
  const int bb = active ? b : 0;
  float* m = &buf[c][bb][0];
  float* dp = &dcp[c][bb][0];

  // LLF step 1: lane i < CX takes DC column i, vertical CY-point DCT (x 1/CY)
  if (active && i < CX) {
    float v[CY];
    if constexpr (LD::kPipelined) {
#pragma unroll
      for (int y = 0; y < CY; y++) v[y] = ld.dc[y];
    } else {
      const BlockHdr& h = hdr[b];
      const float* dc = f.dc[c] + (size_t)h.aby * f.xsb + h.abx + i;
#pragma unroll
      for (int y = 0; y < CY; y++) v[y] = dc[(size_t)y * f.xsb];
    }
    DctReg<CY>(v);
#pragma unroll
    for (int y = 0; y < CY; y++) dp[y * CX + i] = (1.0f / CY) * v[y];
  }

  // dequant + CfL, 4 coefficients per thread and step
  const float* __restrict__ tab = f.dequant + kTab;
  auto dequant_step = [&](int k4, const typename LD::Raw rx, const typename LD::Raw ry,
                          const typename LD::Raw rb, const TabStep& t) {
    const int vb = k4 / SIZE, k = k4 % SIZE;
    const BlockHdr& h = hdr[vb];
    int32_t qx[4], qy[4], qb[4];
    UnpackCoeffs<CT>(rx, qx);
    UnpackCoeffs<CT>(ry, qy);
    UnpackCoeffs<CT>(rb, qb);
    const float mx[4] = {t.x.x, t.x.y, t.x.z, t.x.w};
    const float my[4] = {t.y.x, t.y.y, t.y.z, t.y.w};
    const float mb[4] = {t.b.x, t.b.y, t.b.z, t.b.w};
    const int row = k / L, col = k % L;
    float* ox = &buf[0][vb][row * LP + col];
    float* oy = &buf[1][vb][row * LP + col];
    float* ob = &buf[2][vb][row * LP + col];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float dy = AdjustQuantBias(qy[j], f.biases[1], f.biases[3]) * (my[j] * h.sy);
      const float dx = AdjustQuantBias(qx[j], f.biases[0], f.biases[3]) * (mx[j] * h.sx);
      const float db = AdjustQuantBias(qb[j], f.biases[2], f.biases[3]) * (mb[j] * h.sb);
      ox[j] = __builtin_fmaf(h.x_cc, dy, dx);
      oy[j] = dy;
      ob[j] = __builtin_fmaf(h.b_cc, dy, db);
    }
  };
  if constexpr (LD::kPipelined) {
    // table from LDS: a global load here would sit behind the next batch's
    // prefetch in the in-order vmcnt queue and serialise with it
    const float* lt = reinterpret_cast<const float*>(smem + G::kTabOffset);
#pragma unroll
    for (int it = 0; it < LD::kSteps; it++) {
      const int k4 = tid * 4 + it * 768;
      if (k4 < nb * SIZE) {
        const int k = k4 % SIZE;
        TabStep t;
        t.x = *(const float4*)(lt + k);
        t.y = *(const float4*)(lt + SIZE + k);
        t.b = *(const float4*)(lt + 2 * SIZE + k);
        dequant_step(k4, ld.x[it], ld.y[it], ld.b[it], t);
      }
    }
  } else {
    // all loads of the batch first, then the arithmetic: inside one loop every step would wait
    // for its own loads (~1.5 us each, six steps for a 64x64 varblock)
    using Raw = typename LD::Raw;
    constexpr int kIter = (NB * SIZE + 767) / 768;
    Raw rx[kIter], ry[kIter], rb[kIter];
    TabStep t[kIter];
#pragma unroll
    for (int it = 0; it < kIter; it++) {
      const int k4 = tid * 4 + it * 768;
      if (k4 < nb * SIZE) {
        const int vb = k4 / SIZE, k = k4 % SIZE;
        const size_t at = hdr[vb].coef + k;
        rx[it] = *(const Raw*)((const CT*)f.coeffs[0] + at);
        ry[it] = *(const Raw*)((const CT*)f.coeffs[1] + at);
        rb[it] = *(const Raw*)((const CT*)f.coeffs[2] + at);
        t[it].x = *(const float4*)(tab + k);
        t[it].y = *(const float4*)(tab + SIZE + k);
        t[it].b = *(const float4*)(tab + 2 * SIZE + k);
      }
    }
#pragma unroll
    for (int it = 0; it < kIter; it++) {
      const int k4 = tid * 4 + it * 768;
      if (k4 < nb * SIZE) dequant_step(k4, rx[it], ry[it], rb[it], t[it]);
    }
  }
  __syncthreads();
  // LLF step 2: lane i < CY takes row i of the half-transformed patch,
  // horizontal CX-point DCT (x 1/CX), resample scale, store into the LLF corner
  // (transposed when CY >= CX, like the rest of the coefficient matrix)
  if (active && i < CY) {
    float v[CX];
#pragma unroll
    for (int x = 0; x < CX; x++) v[x] = dp[i * CX + x];
    DctReg<CX>(v);
    const float ry = ResampleUpSel<CY>(i);
#pragma unroll
    for (int x = 0; x < CX; x++) {
      const float val = (1.0f / CX) * v[x];
      if constexpr (CY < CX) {
        m[i * LP + x] = val * ry * kResampleUpHost[CX + x];
      } else {
        m[x * LP + i] = val * kResampleUpHost[CX + x] * ry;
      }
    }
  }
  __syncthreads();
  // pass 1: for each vertical frequency u, C-point IDCT along v -> T[u][x]
  {
    float v[C];
    if (active && i < R) {
#pragma unroll
      for (int j = 0; j < C; j++) v[j] = (R < C) ? m[i * LP + j] : m[j * LP + i];
      IdctReg<C>(v);
    }
    __syncthreads();
    if (active && i < R) {
#pragma unroll
      for (int j = 0; j < C; j++) m[i * TP + j] = v[j];
    }
  }
  __syncthreads();
  // pass 2: for each pixel column x, R-point IDCT along u -> pixels, in place
  // (lane i owns column i of T)
  if (active && i < C) {
    float v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = m[j * TP + i];
    IdctReg<R>(v);
#pragma unroll
    for (int j = 0; j < R; j++) m[j * TP + i] = v[j];
  }
  __builtin_amdgcn_wave_barrier();
  // the wave of channel c moves its NB pixel rectangles to the block-major
  // planes as whole 16-byte tile parts: 16 consecutive lanes write one 256-byte
  // tile, the next 16 the tile to its right (contiguous in memory)
  {
    constexpr int kParts = SIZE / 4;
    for (int e = lane; e < nb * kParts; e += 64) {
      const int vb = e / kParts, r = e % kParts;
      const int t = r >> 4, part = r & 15;
      const int ty = t / CX, tx = t % CX;
      const float* src = &buf[c][vb][(ty * 8 + (part >> 1)) * TP + tx * 8 + (part & 1) * 4];
      const BlockHdr& h = hdr[vb];
      *(float4*)(TilePtr(f, c, h.aby + ty, h.abx + tx) + part * 4) =
          make_float4(src[0], src[1], src[2], src[3]);
    }
  }
}

// One unit = 64 blocks of area of one medium class (kUnitVarblocks varblocks
// starting at list[first]), decoded batch by batch; the coefficient loads of
// batch b+1 are issued before batch b is decoded.
template <int R, int C, int STRATEGY, typename CT>
__device__ __forceinline__ void MediumUnit(const DevFrame& f, const WorkItem* __restrict__ list,
                                           uint32_t first, uint32_t n, unsigned char* smem) {
  using G = MediumGeom<R, C>;
  using LD = MediumLoads<R, C, CT>;
  BlockHdr* hdr = reinterpret_cast<BlockHdr*>(smem + G::kHdrOffset);
  const int nvb = (int)min((uint32_t)G::kUnitVarblocks, n - first);
  const int tid0 = Tid();
  if (tid0 < nvb) hdr[tid0] = MakeHdr(f, list[first + tid0]);
  if constexpr (LD::kPipelined) {
    const float4* __restrict__ tab = (const float4*)(f.dequant + DequantOffset(STRATEGY));
    float4* lt = reinterpret_cast<float4*>(smem + G::kTabOffset);
    for (int k = tid0; k < 3 * LD::SIZE / 4; k += 192) lt[k] = tab[k];
  }
  __syncthreads();
  LD cur;
  if constexpr (LD::kPipelined) MediumFetch<R, C, CT>(f, hdr, min(G::NB, nvb), cur);
  for (int b0 = 0; b0 < nvb; b0 += G::NB) {
    LD nxt;
    if constexpr (LD::kPipelined) {
      if (b0 + G::NB < nvb) MediumFetch<R, C, CT>(f, hdr + b0 + G::NB, min(G::NB, nvb - b0 - G::NB), nxt);
    }
    MediumBatch<R, C, STRATEGY, CT>(f, hdr + b0, min(G::NB, nvb - b0), cur, smem);
    __syncthreads();
    if constexpr (LD::kPipelined) cur = nxt;
  }
}

// ---------------------------------------------------- family A, the next varblock's loads in flight (round 6)
// MediumUnit above handles a 64-point varblock as load -> wait -> dequantise -> two passes -> store, one varblock at a
// time and three workgroups per CU (its 50 KB of LDS): measured per phase (tools/r06/medium_timing.py,
// profiles/r06_medium_timing.txt) the two IDCT passes are 3 % of a varblock's time, the load phase 52 % and the store
// phase 39 % -- a wave's vmcnt counter retires loads AND stores in issue order, so the loads of varblock k + 1, issued
// behind the 16 KB of stores of varblock k, are only "there" when those stores have drained.  Here a workgroup walks its
// varblocks itself (one varblock per task, whatever the class) and requests varblock k + 1's coefficient rows, dequant
// table vectors and DC column -- 116 registers that nothing else needs at that point -- BETWEEN varblock k's second pass
// and its stores: they travel during the store phase, and waiting for them does not wait for the stores behind them.
// 16-bit coefficients only (32-bit rows would be 72 registers more: MediumUnit keeps those frames).
struct APrefetch {
  uint2 x[6], y[6], b[6];     // the thread's 4 coefficients per step and channel (6 steps of 768 for 64x64, 3 for the halves)
  float dc[8];                // lanes < CX of each channel wave: the DC column of LowestFrequenciesFromDC's first step
};

template <int R, int C, int STRATEGY>
__device__ __forceinline__ void ARequest(const DevFrame& f, const BlockHdr& h, APrefetch& P) {
  using G = MediumGeom<R, C>;
  static_assert(G::NB == 1 && G::ML == 64, "one 64-point varblock per task");
  constexpr int SIZE = R * C;
  constexpr int kIter = (SIZE + 767) / 768;
  static_assert(kIter <= 6 && G::CY <= 8, "APrefetch capacity");
  const int tid = Tid();
  const int c = tid >> 6, lane = tid & 63;
  {
    const float* dc = f.dc[c] + (size_t)h.aby * f.xsb + h.abx + (lane < G::CX ? lane : 0);
#pragma unroll
    for (int y = 0; y < 8; y++) P.dc[y] = y < G::CY ? dc[(size_t)y * f.xsb] : 0.0f;
  }
  // EVERY element of P is written here, unconditionally (threads past the varblock's end repeat its first vector, steps
  // the class does not have get zeros): an element that is only sometimes written would stay alive, as far as the register
  // allocator can tell, across the passes of every varblock -- 54 registers the 64-point IDCTs do not have
#pragma unroll
  for (int it = 0; it < 6; it++) {
    if (it < kIter) {
      const int k4 = tid * 4 + it * 768;
      const int kk = k4 < SIZE ? k4 : 0;
      const size_t at = h.coef + kk;
      P.x[it] = *(const uint2*)((const int16_t*)f.coeffs[0] + at);
      P.y[it] = *(const uint2*)((const int16_t*)f.coeffs[1] + at);
      P.b[it] = *(const uint2*)((const int16_t*)f.coeffs[2] + at);
    } else {
      P.x[it] = P.y[it] = P.b[it] = make_uint2(0u, 0u);
    }
  }
}

// dequantisation + LLF + both passes of one varblock whose loads are in P; the pixels end up in the LDS buffer
// (the arithmetic of MediumBatch with one varblock per batch, statement for statement)
template <int R, int C, int STRATEGY>
__device__ __forceinline__ void AHead(const DevFrame& f, const BlockHdr& h, const APrefetch& P, unsigned char* smem) {
  using G = MediumGeom<R, C>;
  constexpr int L = G::L, LP = G::LP, TP = G::TP, BUF = G::BUF, CY = G::CY, CX = G::CX;
  constexpr int SIZE = R * C;
  constexpr int kIter = (SIZE + 767) / 768;
  float(*buf)[BUF] = reinterpret_cast<float(*)[BUF]>(smem);
  float(*dcp)[CY * CX] = reinterpret_cast<float(*)[CY * CX]>(smem + 12 * BUF);
  const int tid = Tid();
  const int c = tid >> 6, i = tid & 63;
  float* m = &buf[c][0];
  float* dp = &dcp[c][0];
  if (i < CX) {  // LLF step 1
    float v[CY];
#pragma unroll
    for (int y = 0; y < CY; y++) v[y] = P.dc[y];
    DctReg<CY>(v);
#pragma unroll
    for (int y = 0; y < CY; y++) dp[y * CX + i] = (1.0f / CY) * v[y];
  }
  // the table vectors: L2 hits, requested here (a table row per thread and step: 72 registers that are free now and were
  // not while the previous varblock's passes ran)
  const float* __restrict__ tab = f.dequant + DequantOffset(STRATEGY);
  float4 tx[kIter], ty[kIter], tb[kIter];
#pragma unroll
  for (int it = 0; it < kIter; it++) {
    const int k4 = tid * 4 + it * 768;
    const int kk = k4 < SIZE ? k4 : 0;
    tx[it] = *(const float4*)(tab + kk);
    ty[it] = *(const float4*)(tab + SIZE + kk);
    tb[it] = *(const float4*)(tab + 2 * SIZE + kk);
  }
#pragma unroll
  for (int it = 0; it < kIter; it++) {
    const int k = tid * 4 + it * 768;
    if (k < SIZE) {
      int32_t qx[4], qy[4], qb[4];
      UnpackCoeffs<int16_t>(P.x[it], qx);
      UnpackCoeffs<int16_t>(P.y[it], qy);
      UnpackCoeffs<int16_t>(P.b[it], qb);
      const float mx[4] = {tx[it].x, tx[it].y, tx[it].z, tx[it].w};
      const float my[4] = {ty[it].x, ty[it].y, ty[it].z, ty[it].w};
      const float mb[4] = {tb[it].x, tb[it].y, tb[it].z, tb[it].w};
      const int row = k / L, col = k % L;
      float* ox = &buf[0][row * LP + col];
      float* oy = &buf[1][row * LP + col];
      float* ob = &buf[2][row * LP + col];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float dy = AdjustQuantBias(qy[j], f.biases[1], f.biases[3]) * (my[j] * h.sy);
        const float dx = AdjustQuantBias(qx[j], f.biases[0], f.biases[3]) * (mx[j] * h.sx);
        const float db = AdjustQuantBias(qb[j], f.biases[2], f.biases[3]) * (mb[j] * h.sb);
        ox[j] = __builtin_fmaf(h.x_cc, dy, dx);
        oy[j] = dy;
        ob[j] = __builtin_fmaf(h.b_cc, dy, db);
      }
    }
  }
  __syncthreads();
  if (i < CY) {  // LLF step 2
    float v[CX];
#pragma unroll
    for (int x = 0; x < CX; x++) v[x] = dp[i * CX + x];
    DctReg<CX>(v);
    const float ry = ResampleUpSel<CY>(i);
#pragma unroll
    for (int x = 0; x < CX; x++) {
      const float val = (1.0f / CX) * v[x];
      if constexpr (CY < CX) {
        m[i * LP + x] = val * ry * kResampleUpHost[CX + x];
      } else {
        m[x * LP + i] = val * kResampleUpHost[CX + x] * ry;
      }
    }
  }
  __syncthreads();
  {  // pass 1
    float v[C];
    if (i < R) {
#pragma unroll
      for (int j = 0; j < C; j++) v[j] = (R < C) ? m[i * LP + j] : m[j * LP + i];
      IdctReg<C>(v);
    }
    __syncthreads();
    if (i < R) {
#pragma unroll
      for (int j = 0; j < C; j++) m[i * TP + j] = v[j];
    }
  }
  __syncthreads();
  if (i < C) {  // pass 2
    float v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = m[j * TP + i];
    IdctReg<R>(v);
#pragma unroll
    for (int j = 0; j < R; j++) m[j * TP + i] = v[j];
  }
  __builtin_amdgcn_wave_barrier();
  // (the next varblock's requests follow: not above this pass, whose 64-point IDCT needs the registers they land in)
  __builtin_amdgcn_sched_barrier(0);
}

// the channel wave moves its pixel rectangle to the block-major planes (as MediumBatch: whole 16-byte tile parts)
template <int R, int C>
__device__ __forceinline__ void AStore(const DevFrame& f, const BlockHdr& h, unsigned char* smem) {
  using G = MediumGeom<R, C>;
  constexpr int TP = G::TP, BUF = G::BUF, CX = G::CX;
  constexpr int kParts = R * C / 4;
  float(*buf)[BUF] = reinterpret_cast<float(*)[BUF]>(smem);
  const int tid = Tid();
  const int c = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int e0 = 0; e0 < kParts; e0 += 64) {
    const int r = e0 + lane;
    const int t = r >> 4, part = r & 15;
    const int ty = t / CX, tx = t % CX;
    const float* src = &buf[c][(ty * 8 + (part >> 1)) * TP + tx * 8 + (part & 1) * 4];
    *(float4*)(TilePtr(f, c, h.aby + ty, h.abx + tx) + part * 4) = make_float4(src[0], src[1], src[2], src[3]);
    // (four parts at a time: left alone the scheduler reads all sixteen parts ahead -- 64 registers beside the 116 of the
    // next varblock's loads in flight)
    if ((e0 & 192) == 192) __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------- k_large
// Strategies 21..26 (128x128 .. 256x256; legal but never emitted by libjxl).  One workgroup per varblock; 1-D
// transforms of up to 256 points run on per-thread scratch arrays; the
// intermediate T[u][x] is written into the varblock's own output rectangle.
template <int N>
__device__ __noinline__ void IdctMemT(float* v, float* tmp, const float* __restrict__ wc) {
  if constexpr (N == 1) {
    return;
  } else if constexpr (N == 2) {
    const float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else {
    constexpr int h = N / 2;
#pragma unroll 1
    for (int i = 0; i < h; i++) tmp[i] = v[2 * i];
#pragma unroll 1
    for (int i = 0; i < h; i++) tmp[h + i] = v[2 * i + 1];
    IdctMemT<h>(tmp, tmp + N, wc);
#pragma unroll 1
    for (int i = h - 1; i > 0; i--) tmp[h + i] = tmp[h + i] + tmp[h + i - 1];
    tmp[h] = tmp[h] * kSqrt2;
    IdctMemT<h>(tmp + h, tmp + N, wc);
#pragma unroll 1
    for (int i = 0; i < h; i++) {
      const float mul = wc[N + i];
      const float e = tmp[i], o = tmp[h + i];
      v[i] = __builtin_fmaf(mul, o, e);
      v[N - 1 - i] = __builtin_fmaf(-mul, o, e);
    }
  }
}

template <int N>
__device__ __noinline__ void DctMemT(float* v, float* tmp, const float* __restrict__ wc) {
  if constexpr (N == 1) {
    return;
  } else if constexpr (N == 2) {
    const float a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else {
    constexpr int h = N / 2;
#pragma unroll 1
    for (int i = 0; i < h; i++) tmp[i] = v[i] + v[N - 1 - i];
    DctMemT<h>(tmp, tmp + N, wc);
#pragma unroll 1
    for (int i = 0; i < h; i++) tmp[h + i] = v[i] - v[N - 1 - i];
#pragma unroll 1
    for (int i = 0; i < h; i++) tmp[h + i] = tmp[h + i] * wc[N + i];
    DctMemT<h>(tmp + h, tmp + N, wc);
    tmp[h] = __builtin_fmaf(tmp[h], kSqrt2, tmp[h + 1]);
#pragma unroll 1
    for (int i = 1; i + 1 < h; i++) tmp[h + i] = tmp[h + i] + tmp[h + i + 1];
#pragma unroll 1
    for (int i = 0; i < h; i++) {
      v[2 * i] = tmp[i];
      v[2 * i + 1] = tmp[h + i];
    }
  }
}

__device__ __forceinline__ void IdctMem(int n, float* v, float* tmp, const float* wc) {
  switch (n) {
    case 32: IdctMemT<32>(v, tmp, wc); break;
    case 64: IdctMemT<64>(v, tmp, wc); break;
    case 128: IdctMemT<128>(v, tmp, wc); break;
    case 256: IdctMemT<256>(v, tmp, wc); break;
    default: break;
  }
}
__device__ __forceinline__ void DctMem(int n, float* v, float* tmp, const float* wc) {
  switch (n) {
    case 2: DctMemT<2>(v, tmp, wc); break;
    case 4: DctMemT<4>(v, tmp, wc); break;
    case 8: DctMemT<8>(v, tmp, wc); break;
    case 16: DctMemT<16>(v, tmp, wc); break;
    case 32: DctMemT<32>(v, tmp, wc); break;
    default: break;
  }
}

template <typename CT>
__global__ __launch_bounds__(256) void k_large(DevFrame f, const WorkItem* __restrict__ list,
                                               const uint32_t* __restrict__ count,
                                               const float* __restrict__ wc,
                                               const float* __restrict__ resample) {
  __shared__ float llf[32 * 33];  // LLF corner of the current channel
  __shared__ float dcs[32 * 33];
  const uint32_t n = *count;
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
  const BlockHdr h = MakeHdr(f, list[item]);
  const int strategy = f.acs[(size_t)h.aby * f.xsb + h.abx] >> 1;
  const int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  const int R = cy * 8, C = cx * 8;
  const int L = R < C ? C : R;
  const size_t size = (size_t)R * C;
  const float* __restrict__ tab = f.dequant + DequantOffset(strategy);
  const int tid = threadIdx.x;
  float v[256], tmp[512];
  for (int c = 0; c < 3; c++) {
    // ---- LLF <- DC (cy x cx patch), jxo_llf_from_dc order: vertical then
    // horizontal forward DCT, each scaled by 1/N
    for (int i = tid; i < cy * cx; i += 256) {
      const int y = i / cx, x = i % cx;
      dcs[y * 33 + x] = f.dc[c][(size_t)(h.aby + y) * f.xsb + h.abx + x];
    }
    __syncthreads();
    if (tid < cx) {
      for (int y = 0; y < cy; y++) v[y] = dcs[y * 33 + tid];
      DctMem(cy, v, tmp, wc);
      const float sc = 1.0f / cy;
      for (int y = 0; y < cy; y++) dcs[y * 33 + tid] = sc * v[y];
    }
    __syncthreads();
    if (tid < cy) {
      for (int x = 0; x < cx; x++) v[x] = dcs[tid * 33 + x];
      DctMem(cx, v, tmp, wc);
      const float sc = 1.0f / cx;
      for (int x = 0; x < cx; x++) {
        const float val = sc * v[x];
        // coefficient-matrix position of (u = tid, v = x)
        if (cy < cx) llf[tid * 33 + x] = val * resample[cy + tid] * resample[cx + x];
        else llf[x * 33 + tid] = val * resample[cx + x] * resample[cy + tid];
      }
    }
    __syncthreads();
    // ---- pass 1: thread u, C-point IDCT along v, dequant on the fly
    const float bias_c = f.biases[c], bias3 = f.biases[3];
    const float sc = c == 0 ? h.sx : (c == 1 ? h.sy : h.sb);
    const float cc = c == 0 ? h.x_cc : (c == 2 ? h.b_cc : 0.0f);
    const int srows = R < C ? R : C;  // LLF corner: srows/8 x L/8
    const int llf_r = srows / 8, llf_c = L / 8;
    for (int u = tid; u < R; u += 256) {
      for (int j = 0; j < C; j++) {
        // matrix element (row, col) holding F[u][v=j]
        const int row = R < C ? u : j, col = R < C ? j : u;
        float val;
        if (row < llf_r && col < llf_c) {
          val = llf[row * 33 + col];
        } else {
          const size_t k = (size_t)row * L + col;
          const int32_t q = LoadCoeff<CT>(f.coeffs[c], h.coef + k);
          val = AdjustQuantBias(q, bias_c, bias3) * (tab[c * size + k] * sc);
          if (c != 1) {
            const int32_t qy = LoadCoeff<CT>(f.coeffs[1], h.coef + k);
            const float dy = AdjustQuantBias(qy, f.biases[1], bias3) * (tab[size + k] * h.sy);
            val = __builtin_fmaf(cc, dy, val);
          }
        }
        v[j] = val;
      }
      IdctMem(C, v, tmp, wc);
      for (int j = 0; j < C; j++) *PlanePtr(f, c, h.aby * 8 + u, h.abx * 8 + j) = v[j];
    }
    __threadfence_block();
    __syncthreads();
    // ---- pass 2: thread x, R-point IDCT along u, in place in the plane
    for (int x = tid; x < C; x += 256) {
      for (int j = 0; j < R; j++) v[j] = *PlanePtr(f, c, h.aby * 8 + j, h.abx * 8 + x);
      IdctMem(R, v, tmp, wc);
      for (int j = 0; j < R; j++) *PlanePtr(f, c, h.aby * 8 + j, h.abx * 8 + x) = v[j];
    }
    __syncthreads();
  }
  }
}

// ------------------------------------------------------ class-family dispatch
// Phase 1 is five launches, not one per class:
//   k_transform_8   every single-block strategy, no LDS: the nine special 8x8 kinds as
//                   lane-per-block (unit, channel) tasks at the head of the grid, then DCT8
//                   (~45 % of a d1.0 frame) row-per-lane
//   k_transform_r   16x8 .. 32x32 row-per-lane, no LDS: the L = 32 classes (32x32, 32x16, 16x32,
//                   32x8, 8x32; 32 values per lane) at the head of the grid, then 16x16, 16x8,
//                   8x16.  (k_transform_r16 / r32: the two halves, launched alone when used_acs
//                   says the other half has no work)
//   k_transform_a   64x64, 64x32, 32x64 (LDS-staged MediumUnit)
//   k_large         128x128 .. 256x256 (never emitted by libjxl), private scratch
// A family kernel owns several work classes; a workgroup decodes UNITS of the family -- 64 or
// 128 blocks of area of ONE class, located from the class list lengths k_prepare left on the
// device (a prefix over the family's counters in SGPRs).  The host never learns the list
// lengths; it bounds the unit count by cells/64 + N and caps the grid at a few resident
// generations, the workgroups loop (UnitDispatch).
struct FamilyEntry {
  int cls;
  int unit_varblocks;
};

struct UnitPick {
  int index;       // entry of the family, -1 = past the end
  int cls;         // its work class
  uint32_t first;  // first varblock of the unit in the class list
  uint32_t n;      // class list length
};

template <int N>
__device__ __forceinline__ UnitPick PickUnit(const FamilyEntry (&fam)[N], const uint32_t* cnt,
                                             uint32_t u) {
  UnitPick p{-1, 0, 0, 0};
  uint32_t base = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const uint32_t units = (cnt[i] + fam[i].unit_varblocks - 1) / fam[i].unit_varblocks;
    if (p.index < 0 && u < base + units) {
      p.index = i;
      p.cls = fam[i].cls;
      p.first = (u - base) * fam[i].unit_varblocks;
      p.n = cnt[i];
    }
    base += units;
  }
  return p;
}

static constexpr int kLdsFamilyA = MediumGeom<64, 64>::kLdsBytes;
static_assert(TransposeLdsBytes<32>() <= kLdsFamilyA && TransposeLdsBytes<16>() <= kLdsFamilyA && TransposeLdsBytes<8>() <= kLdsFamilyA,
              "the row-per-lane units of k_transform_r transpose through family A's allocation");
static_assert(sizeof(BlockHdr) <= 48, "header slots are 48 bytes");
static_assert(MediumGeom<64, 32>::kLdsBytes <= kLdsFamilyA && MediumGeom<32, 64>::kLdsBytes <= kLdsFamilyA, "");

// A: 64x64, 64x32, 32x64 (long units first)
static constexpr FamilyEntry kFamilyA[3] = {{kClsMedium0 + 8, 1}, {kClsMedium0 + 9, 2}, {kClsMedium0 + 10, 2}};
// B: 16x8 .. 32x32, long units first so that the drain ends on short ones
// row-per-lane kernels, 4 waves x (64 / S) varblocks per unit.  R16: longer side 16 (4 waves per
// SIMD), R32: longer side 32 (twice the registers per lane)
static constexpr FamilyEntry kFamilyR16[3] = {{kClsMedium0 + 2, 16}, {kClsMedium0 + 0, 32}, {kClsMedium0 + 1, 32}};
static constexpr FamilyEntry kFamilyR32[5] = {{kClsMedium0 + 7, 8},  {kClsMedium0 + 5, 16}, {kClsMedium0 + 6, 16},
                                              {kClsMedium0 + 3, 32}, {kClsMedium0 + 4, 32}};
static_assert(kMediumStrategy[8] == 18 && kMediumStrategy[9] == 19 && kMediumStrategy[10] == 20 &&
              kMediumStrategy[7] == 5 && kMediumStrategy[5] == 10 && kMediumStrategy[6] == 11 &&
              kMediumStrategy[2] == 4 && kMediumStrategy[3] == 8 && kMediumStrategy[4] == 9 &&
              kMediumStrategy[0] == 6 && kMediumStrategy[1] == 7, "class table mismatch");

// Workgroup w decodes units w, w + gridDim.x, ... of the family.  The host only knows the
// bound cells/64 + N on the unit count; a grid of that size costs ~10 us per launch in
// workgroups that find nothing to do when the family covers a small part of the frame, so
// the grid is capped at a few resident generations and the workgroups loop.  (The unit
// functions read their thread index through Tid(), which keeps the compiler from hoisting
// every class's lane-dependent invariants out of this loop.)
template <int N, typename Body>
__device__ __forceinline__ void UnitDispatch(const FamilyEntry (&fam)[N], const WorkLists& wl,
                                             Body&& body, uint32_t wg_index, uint32_t num_wgs,
                                             uint32_t skip_mask = 0) {
  uint32_t cnt[N];
#pragma unroll
  for (int i = 0; i < N; i++)  // skip_mask: classes decoded on the matrix cores (kernels_mfma.hip)
    cnt[i] = (skip_mask >> i) & 1u ? 0u : wl.count[fam[i].cls * kCounterPad];
  // workgroup wg_index of the num_wgs that share the family
  for (uint32_t u = wg_index;; u += num_wgs) {
    const UnitPick pick = PickUnit(fam, cnt, u);
    if (pick.index < 0) return;
    body(pick.index, wl.list[pick.cls], pick.first, pick.n);
    __syncthreads();  // the next unit reuses the LDS
  }
}

// one varblock per task for every class of the family
static constexpr FamilyEntry kFamilyA1[3] = {{kClsMedium0 + 8, 1}, {kClsMedium0 + 9, 1}, {kClsMedium0 + 10, 1}};

__device__ __forceinline__ void FamilyALoop16(const DevFrame& f, const WorkLists& wl, unsigned char* smem, uint32_t wg_index,
                                              uint32_t num_wgs) {
  uint32_t cnt[3];
#pragma unroll
  for (int i = 0; i < 3; i++) cnt[i] = wl.count[kFamilyA1[i].cls * kCounterPad];
  uint32_t u = wg_index;
  UnitPick cur = PickUnit(kFamilyA1, cnt, u);
  if (cur.index < 0) return;
  BlockHdr hc = MakeHdr(f, wl.list[cur.cls][cur.first]);
  APrefetch P;
#define JXLHIP_AREQUEST(INDEX, H)                      \
  switch (INDEX) {                                     \
    case 0: ARequest<64, 64, 18>(f, H, P); break;      \
    case 1: ARequest<64, 32, 19>(f, H, P); break;      \
    default: ARequest<32, 64, 20>(f, H, P); break;     \
  }
  JXLHIP_AREQUEST(cur.index, hc)
  for (;;) {
    const UnitPick nxt = PickUnit(kFamilyA1, cnt, u + num_wgs);
    WorkItem in{};
    if (nxt.index >= 0) in = wl.list[nxt.cls][nxt.first];  // (travels during this varblock's arithmetic)
    switch (cur.index) {
      case 0: AHead<64, 64, 18>(f, hc, P, smem); break;
      case 1: AHead<64, 32, 19>(f, hc, P, smem); break;
      default: AHead<32, 64, 20>(f, hc, P, smem); break;
    }
    BlockHdr hn{};
    if (nxt.index >= 0) {
      hn = MakeHdr(f, in);
      JXLHIP_AREQUEST(nxt.index, hn)  // in front of this varblock's stores: see above
    }
    switch (cur.index) {
      case 0: AStore<64, 64>(f, hc, smem); break;
      case 1: AStore<64, 32>(f, hc, smem); break;
      default: AStore<32, 64>(f, hc, smem); break;
    }
    if (nxt.index < 0) return;
    __syncthreads();  // the next varblock reuses the LDS
    cur = nxt;
    hc = hn;
    u += num_wgs;
  }
#undef JXLHIP_AREQUEST
}

// Family A (64-point transforms, ~5 % of a d1.0 frame) is compiled for three waves per SIMD
// (168 VGPRs, what its LDS use allows anyway); the 64-point transforms would like ~180 and
// spill a few values to scratch instead.  In row-per-lane form (64 values per lane, > 256
// registers, one wave per SIMD) these classes were 30 % slower.
template <typename CT>
__global__ __launch_bounds__(192, sizeof(CT) == 2 ? 3 : 2) void k_transform_a(DevFrame f, WorkLists wl) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsFamilyA];
  if constexpr (sizeof(CT) == 2) {
    FamilyALoop16(f, wl, smem, blockIdx.x, gridDim.x);
  } else {
    UnitDispatch(kFamilyA, wl,
                 [&](int index, const WorkItem* __restrict__ list, uint32_t first, uint32_t n) {
                   switch (index) {
                     case 0: MediumUnit<64, 64, 18, CT>(f, list, first, n, smem); break;
                     case 1: MediumUnit<64, 32, 19, CT>(f, list, first, n, smem); break;
                     default: MediumUnit<32, 64, 20, CT>(f, list, first, n, smem); break;
                   }
                 },
                 blockIdx.x, gridDim.x);
  }
}

template <typename CT>
__global__ __launch_bounds__(256) void k_transform_r16(DevFrame f, WorkLists wl) {
  UnitDispatch(kFamilyR16, wl,
               [&](int index, const WorkItem* __restrict__ list, uint32_t first, uint32_t n) {
                 switch (index) {
                   case 0: RowLaneUnit<16, 16, 4, CT>(f, list, first, n); break;
                   case 1: RowLaneUnit<16, 8, 6, CT>(f, list, first, n); break;
                   default: RowLaneUnit<8, 16, 7, CT>(f, list, first, n); break;
                 }
               },
               blockIdx.x, gridDim.x, f.mfma16 != nullptr ? 1u : 0u);
}

template <typename CT>
__global__ __launch_bounds__(256) void k_transform_r32(DevFrame f, WorkLists wl) {
  UnitDispatch(kFamilyR32, wl,
               [&](int index, const WorkItem* __restrict__ list, uint32_t first, uint32_t n) {
                 switch (index) {
                   case 0: RowLaneUnit<32, 32, 5, CT>(f, list, first, n); break;
                   case 1: RowLaneUnit<32, 16, 10, CT>(f, list, first, n); break;
                   case 2: RowLaneUnit<16, 32, 11, CT>(f, list, first, n); break;
                   case 3: RowLaneUnit<32, 8, 8, CT>(f, list, first, n); break;
                   default: RowLaneUnit<8, 32, 9, CT>(f, list, first, n); break;
                 }
               },
               blockIdx.x, gridDim.x, f.mfma32 != nullptr ? 1u : 0u);
}

// Both row-per-lane families in one launch, the (few, long, register-heavy) L = 32 units first:
// on a mixed frame their single-generation latency (~27 us as a launch of its own) disappears
// behind the L = 16 bulk, at the price of the L = 16 units running at the L = 32 occupancy.
static constexpr FamilyEntry kFamilyR[8] = {{kClsMedium0 + 7, 8},  {kClsMedium0 + 5, 16}, {kClsMedium0 + 6, 16},
                                            {kClsMedium0 + 3, 32}, {kClsMedium0 + 4, 32}, {kClsMedium0 + 2, 16},
                                            {kClsMedium0 + 0, 32}, {kClsMedium0 + 1, 32}};
// ... and, on its first `big_wgs` workgroups, the LDS-staged 64-point classes of family A: their
// workgroups are three waves (the fourth ends at once; a finished wave no longer counts at the
// workgroup barrier) looping over the big units, while the others loop over the row-per-lane units.
// The 50 KB of LDS every workgroup of this launch then reserves cost the row-per-lane units
// nothing: three workgroups per CU is what their registers allow anyway.  As a launch of its own
// family A is ~22 us of pure latency per 8K d1.0 frame.
template <typename CT>
__global__ __launch_bounds__(256, sizeof(CT) == 2 ? 3 : 2) void k_transform_r(DevFrame f, WorkLists wl, uint32_t big_wgs,
                                                                               uint32_t special_wgs, uint32_t r_wgs,
                                                                               uint32_t dct8_wgs) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsFamilyA];
  // The special 8x8 kinds ride at the head of this launch instead of being a latency-bound launch of their own
  // (fused mode: DCT8 is decoded by the fused kernel; two-phase: the DCT8 rows ride here as well, below)
  if (blockIdx.x < special_wgs) {
    SpecialWorkgroup<CT>(f, wl, blockIdx.x);
    return;
  }
  // Two-phase: the DCT8 workgroups (short, LDS-free, the bulk of a d1.0 frame) alternate with the persistent
  // workgroups of the other classes in the dispatch order, so that the two kinds run side by side from the first
  // wave on -- as two launches the second waits for the first's tail, which on frames of a few Mpx is most of it
  // (1080p: blocks 39 -> ? us; the two-stream form pays ~40 us of fork / join events instead).
  // The launch is sized for a frame of nothing but DCT8 (dct8_wgs); k_prepare's count says how many of those
  // workgroups have work (need8).  On genuine content -- a few per cent DCT8 -- thousands of empty workgroups between
  // the persistent ones cost more than the DCT8 blocks themselves: then only the ones with work alternate and the
  // surplus sits at the end of the grid, leaving at once (two-phase, real-content shares: 8K 257 -> 200 us, 4K 61 ->
  // 51 us).  With a quarter or more of the bound in use (the d1 mix: 45 %) the old placement stays: there the empty
  // workgroups between the later persistent ones HELP (4K 39 vs 45 us; profiles/r03_dct8_workgroup_roles.txt).
  const uint32_t np = big_wgs + r_wgs;
  const uint32_t i = blockIdx.x - special_wgs;
  const uint32_t need8 = (wl.count[kClsDct8 * kCounterPad] + Dct8Geom<CT>::kPerWg - 1) / Dct8Geom<CT>::kPerWg;
  const uint32_t d8 = need8 * 4 < dct8_wgs ? need8 : dct8_wgs;
  const uint32_t pairs = np < d8 ? np : d8;
  bool is_dct8;
  uint32_t idx;
  if (i < 2 * pairs) {
    is_dct8 = (i & 1u) != 0;
    idx = i >> 1;
  } else {
    is_dct8 = d8 > np;
    idx = i - pairs;
    if (idx >= (is_dct8 ? d8 : np)) return;
  }
  if (is_dct8) {
    Dct8Rows<CT>(f, wl.list[kClsDct8], wl.count[kClsDct8 * kCounterPad], idx);
    return;
  }
  if (idx < big_wgs) {
    if (threadIdx.x >= 192) return;
    if constexpr (sizeof(CT) == 2) {
      FamilyALoop16(f, wl, smem, idx, big_wgs);
    } else {
      UnitDispatch(kFamilyA, wl,
                   [&](int index, const WorkItem* __restrict__ list, uint32_t first, uint32_t n) {
                     switch (index) {
                       case 0: MediumUnit<64, 64, 18, CT>(f, list, first, n, smem); break;
                       case 1: MediumUnit<64, 32, 19, CT>(f, list, first, n, smem); break;
                       default: MediumUnit<32, 64, 20, CT>(f, list, first, n, smem); break;
                     }
                   },
                   idx, big_wgs);
    }
    return;
  }
  UnitDispatch(kFamilyR, wl,
               [&](int index, const WorkItem* __restrict__ list, uint32_t first, uint32_t n) {
                 switch (index) {
                   case 0: RowLaneUnit<32, 32, 5, CT, true>(f, list, first, n, smem); break;
                   case 1: RowLaneUnit<32, 16, 10, CT, true>(f, list, first, n, smem); break;
                   case 2: RowLaneUnit<16, 32, 11, CT, true>(f, list, first, n, smem); break;
                   case 3: RowLaneUnit<32, 8, 8, CT, true>(f, list, first, n, smem); break;
                   case 4: RowLaneUnit<8, 32, 9, CT, true>(f, list, first, n, smem); break;
                   case 5: RowLaneUnit<16, 16, 4, CT, true>(f, list, first, n, smem); break;
                   case 6: RowLaneUnit<16, 8, 6, CT, true>(f, list, first, n, smem); break;
                   default: RowLaneUnit<8, 16, 7, CT, true>(f, list, first, n, smem); break;
                 }
               },
               idx - big_wgs, r_wgs, (f.mfma32 != nullptr ? 1u : 0u) | (f.mfma16 != nullptr ? 1u << 5 : 0u));
}

// --------------------------------------------------------------- launchers
template <typename CT>
static void LaunchBlocksT(const DevFrame& f, const WorkLists& wl, uint32_t cells, const float* wc,
                          const float* resample, hipStream_t* streams, int nstreams, const FilterParams* emit) {
  const uint32_t units = cells / 64;
  const uint32_t grid_l = cells / 128 < 512u ? (cells / 128 ? cells / 128 : 1) : 512u;
  constexpr uint32_t kDct8PerWg = Dct8Geom<CT>::kPerWg;
  // caps: residency of the kernel (workgroups per CU by LDS / registers) x 256 CUs x 2 generations
  const uint32_t grid_a = units + 3 < 1536u ? units + 3 : 1536u;
  const uint32_t grid_r16 = units + 3 < 4096u ? units + 3 : 4096u;
  const uint32_t grid_r32 = units / 2 + 5 < 3072u ? units / 2 + 5 : 3072u;  // units of 128 blocks
  // With two streams the latency-bound family A (a few hundred long 64x64 / 64x32 units and the
  // special 8x8 kinds, LDS-heavy, low occupancy) runs beside the LDS-free bandwidth-bound k_dct8.
  hipStream_t s0 = streams[0], s1 = streams[1 % nstreams];
  // used_acs (when the caller knows it) says which families have work at all
  auto any = [&](std::initializer_list<int> strategies) {
    if (f.used_acs == 0) return true;
    for (int st : strategies)
      if (f.used_acs & (1u << st)) return true;
    return false;
  };
  const bool need_r16 = any({6, 7}) || (!f.mfma16 && any({4}));
  const bool need_r32 = any({8, 9, 10, 11}) || (!f.mfma32 && any({5}));
  const bool merged_r = need_r16 && need_r32;
  const bool have_big = any({18, 19, 20});
  bool specials_in_r = false, dct8_in_r = false;
  uint32_t grid_specials = 0, grid_dct8 = 0;
  if (have_big && !merged_r)
    hipLaunchKernelGGL((k_transform_a<CT>), dim3(grid_a), dim3(192), 0, s1, f, wl);
  {
    // worst cases: all cells special (3 tasks per 64 blocks, 4 tasks per workgroup) or all DCT8
    const bool specials = any({1, 2, 3, 12, 13, 14, 15, 16, 17});
    const uint32_t bound_s = specials ? (cells / 64 + kNumSpecial) * 3 / 4 + 1 : 0;
    // fused == 2 (a stripe): only the DCT8 cells of the two block rows its neighbours pull are on the list
    const uint32_t cells_8 = f.fused == 0 ? cells : (f.fused == 2 ? 2u * f.xsb : 0u);
    const uint32_t bound_8 = (any({0}) && cells_8) ? (cells_8 + kDct8PerWg - 1) / kDct8PerWg : 0;
    const uint32_t grid_8 = (bound_s > bound_8 ? bound_s : bound_8) + (specials ? kNumSpecial : 0);
    // merged_r: the single-block classes ride in k_transform_r's launch (specials at its head, DCT8 workgroups
    // alternating with the other classes' persistent ones)
    specials_in_r = merged_r && specials;
    dct8_in_r = merged_r && bound_8 != 0;
    grid_specials = bound_s + kNumSpecial;
    grid_dct8 = bound_8;
    if (grid_8 && !merged_r) hipLaunchKernelGGL((k_transform_8<CT>), dim3(grid_8), dim3(256), 0, s0, f, wl);
  }
  if (merged_r) {  // -10 us per 8K d1.0 frame against two launches, -15 us more with family A inside
    uint32_t big_cap = 512u;
    const int big_env = jxlhip_env::Get().big_wgs.load(std::memory_order_relaxed);  // experiments: workgroups of the 64-point family
    if (big_env >= 1 && big_env <= 4096) big_cap = (uint32_t)big_env;  // (anything else: the built-in cap)
    const uint32_t big_wgs = have_big ? (grid_a < big_cap ? grid_a : big_cap) : 0u;
    const uint32_t special_wgs = specials_in_r ? grid_specials : 0u;
    const uint32_t dct8_wgs = dct8_in_r ? grid_dct8 : 0u;
    // (round 6, again: the 16-point classes in a launch of their own -- 93 VGPRs, no LDS, five waves per SIMD, where inside
    // this launch they run at three: all-DCT16X16 frames 159 against 115 us -- cost 9 us more per frame on both the d1 mix
    // and genuine-content shares, profiles/r06_transform_overread.txt; the single launch stays)
    hipLaunchKernelGGL((k_transform_r<CT>), dim3(special_wgs + big_wgs + grid_r16 + dct8_wgs), dim3(256), 0, s0, f, wl,
                       big_wgs, special_wgs, grid_r16, dct8_wgs);
  } else {
    if (need_r16) hipLaunchKernelGGL((k_transform_r16<CT>), dim3(grid_r16), dim3(256), 0, s0, f, wl);
    if (need_r32) hipLaunchKernelGGL((k_transform_r32<CT>), dim3(grid_r32), dim3(256), 0, s0, f, wl);
  }
  if (f.mfma32 && any({5})) LaunchMfma32(f, wl, cells, s1, emit);
  if (f.mfma16 && any({4})) LaunchMfma16(f, wl, cells, s1);
  if (cells >= 256 && any({21, 22, 23, 24, 25, 26}))
    hipLaunchKernelGGL(k_large<CT>, dim3(grid_l), dim3(256), 0, s1, f, wl.list[kClsLarge],
                       wl.count + kClsLarge * kCounterPad, wc, resample);
}

void LaunchPrepare(const DevFrame& f, const WorkLists& wl, int with_sigma, float epf_quant_mul,
                   const SharpLut& lut, hipStream_t st) {
  // lists for the band's group rows; sigma additionally for the group row just
  // outside the STRIPE when the band touches its first / last row (the EPF
  // stages evaluate halo rows there; inside the stripe the neighbouring bands
  // provide their own sigma before any filter launch needs it)
  uint32_t lo = f.band_g0, hi = f.band_g1;
  if (with_sigma) {
    if (lo == f.group_y0 && lo > 0) lo--;
    if (hi == f.group_y0 + f.group_rows && hi < f.ysg) hi++;
  }
  hipLaunchKernelGGL(k_prepare, dim3(f.xsg * (hi - lo)), dim3(1024), 0, st, f, wl, lo,
                     with_sigma, epf_quant_mul, lut);
}

void LaunchBlocks(const DevFrame& f, const WorkLists& wl, uint32_t cells, const float* wc,
                  const float* resample, hipStream_t* streams, int nstreams, const FilterParams* emit) {
  if (f.coeff_type == JXLHIP_COEFF_I16)
    LaunchBlocksT<int16_t>(f, wl, cells, wc, resample, streams, nstreams, emit);
  else
    LaunchBlocksT<int32_t>(f, wl, cells, wc, resample, streams, nstreams, emit);
}

}  // namespace jxlhip
