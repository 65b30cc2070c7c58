// kernels_mfma.hip -- the 32x32 and 16x16 inverse DCTs as two dense matrix products on the matrix cores
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate -- an fmaf chain, so the arithmetic
// stays fp32 like the rest of the path).  BASELINE configs[4] ("MFMA large-DCT path") / north_star: "MFMA only for the 16x16 /
// 32x32 DCT tiles where it is a true dense matmul".  Replaces, for DCT32X32 varblocks, what RowLaneUnit<32, 32>
// (kernels_blocks.hip) does with butterflies and register transposes: ComputeScaledIDCT<32, 32>
// (lib/jxl/dct-inl.h:376-397) = IDCT1D along both axes (dct-inl.h:191-232, scales dct_scales.h:237-369).
//
// One wave = one varblock, channel after channel.  With B[k][n] = the n-th output of the 32-point IDCT1D
// of the k-th unit vector (the exact linear map IdctReg<32> computes, tabulated on the host in double),
// F[u][v] the coefficient of vertical frequency u / horizontal frequency v, the pixels are
//     P = B^T F B,     and the stream stores M = F^T (rows = horizontal frequency: dec_transforms-inl.h:456-459),
// so                 P^T = B^T (M B):
//   product 1  Q = M B      A operand = M: lane (m = lane % 32, h = lane / 32) holds M[m][16 h .. 16 h + 15] --
//                           32 contiguous bytes of the int16 stream: the wave's two 16-byte loads per lane and
//                           channel fetch the whole 2 KB block, dequantised in place (the K index of an MFMA
//                           chain may be visited in any order as long as both operands agree: step kk of lane
//                           half h is k = 16 h + kk; the B operand is the constant table arranged to match)
//   product 2  P^T = B^T Q  Q leaves product 1 in the accumulator layout (row = 4 h + i % 4 + 8 (i / 4) in
//                           register i, column = lane % 32), which IS a B-operand layout up to that same
//                           freedom in K: register i feeds step i unchanged, no data movement between the
//                           products; the A operand is the second constant table
//   result     register i of lane (n, h) = P^T[x][y = n], x = 4 h + i % 4 + 8 (i / 4): four runs of four
//              consecutive pixels of image row y -> four 16-byte stores into the block-major tiles.
// 32 MFMAs per channel (64 cycles each on one SIMD) = 2 cycles per pixel and channel; the VALU only dequantises.
//
// DCT16X16 (k_transform_mfma16, below) is the same scheme on v_mfma_f32_16x16x4_f32: lane (m = lane % 16,
// h = lane / 16) holds M[m][4 h .. 4 h + 3] (8 or 16 contiguous bytes of the stream: one load per lane and channel,
// the wave fetches the whole block), step kk of lane quarter h is k = 4 h + kk; product 1's accumulator (register i
// of lane (n, h) = Q[4 h + i][n]) is product 2's B operand as it stands, and ONE constant table serves both
// products (B operand of product 1 = A operand of product 2 = B[4 h + step][lane % 16]).  The result is four
// consecutive pixels of image row m per lane: one 16-byte store.  8 MFMAs (32 cycles each) per channel = 1 cycle
// per pixel and channel.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "blocks_common.h"
#include "emit.h"

namespace jxlhip {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

// B[k][n] on the host: the recursion of IdctReg<N> (dev_common.h) in double on unit vectors
void IdctHost(double* v, int n) {
  if (n == 1) return;
  if (n == 2) {
    const double a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
    return;
  }
  const int h = n / 2;
  double e[128], o[128];
  for (int i = 0; i < h; i++) {
    e[i] = v[2 * i];
    o[i] = v[2 * i + 1];
  }
  IdctHost(e, h);
  for (int i = h - 1; i > 0; i--) o[i] = o[i] + o[i - 1];
  o[0] = o[0] * 1.41421356237309504880;
  IdctHost(o, h);
  for (int i = 0; i < h; i++) {
    const double mul = 1.0 / (2.0 * cos((i + 0.5) * M_PI / n));  // W_N[i], dct_scales.h:234-236
    v[i] = e[i] + mul * o[i];
    v[n - 1 - i] = e[i] - mul * o[i];
  }
}

// What the next varblock needs from memory, fetched while the current one is on the matrix cores
template <typename CT>
struct Staged {
  static constexpr int kVec = 16 * (int)sizeof(CT) / 16;  // 16-byte loads per lane and channel
  WorkItem it;
  uint4 raw[3][kVec];
  float dc;  // lane l < 48: DC value (channel l / 16, row l / 4 % 4, column l % 4) of the varblock's 4x4 patch
};

// EMIT: a frame made of DCT32X32 varblocks only, without loop filter, written as linear float RGB (BASELINE
// configs[4]): the class kernel applies the opsin inverse itself and stores the pixels -- no XYB planes, no second
// kernel.  Product 2 then runs with its operands swapped (A = product 1's accumulator, B = the constant table: the
// same table, the accumulator layout is an A-operand layout as well), which leaves lane = pixel COLUMN and
// register i = pixel row 4 h + i % 4 + 8 (i / 4): one 12-byte store per register, 32 lanes = 384 contiguous bytes
// of an output row.
template <typename CT, bool EMIT>
__global__ __launch_bounds__(256, 2) void k_transform_mfma32(DevFrame f, const WorkItem* __restrict__ list,
                                                             const uint32_t* __restrict__ count,
                                                             const float* __restrict__ bc, FilterParams P) {
  constexpr int kVec = Staged<CT>::kVec;
  __shared__ float patch_lds[4][48];
  const uint32_t n = *count;
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  float* patch = patch_lds[threadIdx.x >> 6];
  // the two constant operand tables, one register per MFMA step
  float b1[16], a2[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    b1[i] = bc[i * 64 + lane];
    a2[i] = bc[1024 + i * 64 + lane];
  }
  // the class's dequant matrices in LDS (rows padded to 36 floats: the lanes' 64-byte reads then cover all
  // banks): as global loads they would sit in the same in-order vmcnt queue as the prefetch below and
  // every wait for them would drain it
  __shared__ __attribute__((aligned(16))) float tab_lds[3 * 32 * 36];
  for (int i = threadIdx.x; i < 3 * 1024; i += 256)
    tab_lds[(i >> 5) * 36 + (i & 31)] = f.dequant[DequantOffset(5) + i];
  __syncthreads();
  const uint32_t stride = gridDim.x * 4;
  uint32_t vb = blockIdx.x * 4 + (threadIdx.x >> 6);
  auto item = [&](uint32_t i) { return list[i < n ? i : n - 1]; };
  auto fetch = [&](const WorkItem it, Staged<CT>& s) {
    s.it = it;
    const size_t coef = (size_t)it.off * 64u + (size_t)m * 32 + 16 * h;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint4* p = (const uint4*)((const CT*)f.coeffs[c] + coef);
#pragma unroll
      for (int i = 0; i < kVec; i++) s.raw[c][i] = p[i];
    }
    const int l = lane < 48 ? lane : 47;
    s.dc = f.dc[l >> 4][(size_t)((it.pos >> 16) + ((l >> 2) & 3)) * f.xsb + (it.pos & 0xffffu) + (l & 3)];
  };
  if (vb >= n) return;
  Staged<CT> cur;
  fetch(item(vb), cur);
  WorkItem it_next = item(vb + stride);
  for (; vb < n; vb += stride) {
    Staged<CT> nxt;
    fetch(it_next, nxt);  // (a repeat of the last varblock past the end of the list: loaded, never used)
    it_next = item(vb + 2 * stride);
    const BlockHdr hd = MakeHdr(f, cur.it);
    int tab_at = m * 36 + 16 * h;
    asm volatile("" : "+v"(tab_at));  // re-read per varblock: hoisted out of the loop they are 48 registers
    const float* tab = tab_lds + tab_at;
    // lowest frequencies from the 4x4 DC patch (LowestFrequenciesFromDC, dec_transforms-inl.h:691-818): the
    // lanes of matrix rows 0..3, first half, hold the corner M[x][0..3]
    if (lane < 48) patch[lane] = cur.dc;
    float llf[3][4];
    const bool llf_lane = h == 0 && m < 4;
    if (llf_lane) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float dp[4][4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
          float v[4];
#pragma unroll
          for (int y = 0; y < 4; y++) v[y] = patch[c * 16 + y * 4 + x];
          DctReg<4>(v);
#pragma unroll
          for (int y = 0; y < 4; y++) dp[y][x] = 0.25f * v[y];
        }
#pragma unroll
        for (int y = 0; y < 4; y++) {
          float v[4];
#pragma unroll
          for (int x = 0; x < 4; x++) v[x] = dp[y][x];
          DctReg<4>(v);
          const float ry = kResampleUpHost[4 + y];
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const float val = 0.25f * v[x];
            if (m == x) llf[c][y] = val * kResampleUpHost[4 + x] * ry;
          }
        }
      }
    }
    auto unpack = [&](const uint4* r, int32_t* q) {
      if constexpr (sizeof(CT) == 2) {
#pragma unroll
        for (int i = 0; i < kVec; i++) {
          const uint32_t w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            q[i * 8 + 2 * k] = (int32_t)(int16_t)(w[k] & 0xffffu);
            q[i * 8 + 2 * k + 1] = (int32_t)w[k] >> 16;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < kVec; i++) {
          q[i * 4] = (int32_t)r[i].x;
          q[i * 4 + 1] = (int32_t)r[i].y;
          q[i * 4 + 2] = (int32_t)r[i].z;
          q[i * 4 + 3] = (int32_t)r[i].w;
        }
      }
    };
    v16f px[3];  // EMIT: the three channels' pixels
    float vy[16];
    {
      int32_t q[16];
      unpack(cur.raw[1], q);
#pragma unroll
      for (int k = 0; k < 16; k++) vy[k] = AdjustQuantBias(q[k], f.biases[1], f.biases[3]) * (tab[32 * 36 + k] * hd.sy);
    }
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
      float v[16];
      if (c == 1) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = vy[k];
      } else {
        const float sc = c == 0 ? hd.sx : hd.sb;
        const float cc = c == 0 ? hd.x_cc : hd.b_cc;
        int32_t q[16];
        unpack(cur.raw[c], q);
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const float d = AdjustQuantBias(q[k], f.biases[c], f.biases[3]) * (tab[c * 32 * 36 + k] * sc);
          v[k] = __builtin_fmaf(cc, vy[k], d);
        }
      }
      if (llf_lane) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = llf[c][k];
      }
      // the steps that involve the LLF corner last: it is the longest dependency chain of the block
      v16f q = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kk = 15; kk >= 0; kk--) q = __builtin_amdgcn_mfma_f32_32x32x2f32(v[kk], b1[kk], q, 0, 0, 0);
      v16f p = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (EMIT) {
#pragma unroll
        for (int i = 0; i < 16; i++) p = __builtin_amdgcn_mfma_f32_32x32x2f32(q[i], a2[i], p, 0, 0, 0);
        px[c] = p;
      } else {
#pragma unroll
        for (int i = 0; i < 16; i++) p = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[i], q[i], p, 0, 0, 0);
        // lane (y = m, h): P[y][8 t + 4 h .. + 3] in registers 4 t .. 4 t + 3
#pragma unroll
        for (int t = 0; t < 4; t++) {
          float* dst = TilePtr(f, c, hd.aby + (m >> 3), hd.abx + t) + (m & 7) * 8 + 4 * h;
          *(float4*)dst = make_float4(p[4 * t], p[4 * t + 1], p[4 * t + 2], p[4 * t + 3]);
        }
      }
    }
    if constexpr (EMIT) {
      // lane (x = m, h), register i: pixel (row 4 h + i % 4 + 8 (i / 4), column x) of the varblock
      const int x = (int)hd.abx * 8 + m;
      const int y0 = (int)hd.aby * 8 + 4 * h;
      char* col = (char*)P.out + (size_t)x * 12;
      typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int y = y0 + (i & 3) + 8 * (i >> 2);
        float rgb[3];
        XybToRgb(px[0][i], px[1][i], px[2][i], P, rgb);
        if (x < (int)f.xsize && y >= (int)f.y0 && y < (int)f.y1)
          __builtin_nontemporal_store(f3u{rgb[0], rgb[1], rgb[2]}, (f3u*)(col + (size_t)(y - (int)f.y0) * P.out_stride));
      }
    }
    cur = nxt;
  }
}

typedef float v4f __attribute__((ext_vector_type(4)));

template <typename CT>
struct Staged16 {
  typedef typename std::conditional<sizeof(CT) == 2, uint2, uint4>::type Raw;  // M[m][4 h .. 4 h + 3]
  WorkItem it;
  Raw raw[3];
  float dc;  // lane l < 12: DC value (channel l / 4, row l / 2 % 2, column l % 2) of the varblock's 2x2 patch
};

template <typename CT>
__global__ __launch_bounds__(256, 4) void k_transform_mfma16(DevFrame f, const WorkItem* __restrict__ list,
                                                             const uint32_t* __restrict__ count,
                                                             const float* __restrict__ bc) {
  typedef typename Staged16<CT>::Raw Raw;
  // the class's dequant matrices in LDS, rows padded to 20 floats (a 16-lane group's 16-byte reads then cover all
  // banks); as global loads they would share the in-order vmcnt queue with the prefetch below
  __shared__ __attribute__((aligned(16))) float tab_lds[3 * 16 * 20];
  for (int i = threadIdx.x; i < 3 * 256; i += 256) tab_lds[(i >> 4) * 20 + (i & 15)] = f.dequant[DequantOffset(4) + i];
  __syncthreads();
  const uint32_t n = *count;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, h = lane >> 4;
  float b1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) b1[i] = bc[i * 64 + lane];
  const uint32_t stride = gridDim.x * 4;
  uint32_t vb = blockIdx.x * 4 + (threadIdx.x >> 6);
  auto item = [&](uint32_t i) { return list[i < n ? i : n - 1]; };
  auto fetch = [&](const WorkItem it, Staged16<CT>& s) {
    s.it = it;
    const size_t coef = (size_t)it.off * 64u + (size_t)m * 16 + 4 * h;
#pragma unroll
    for (int c = 0; c < 3; c++) s.raw[c] = *(const Raw*)((const CT*)f.coeffs[c] + coef);
    const int l = lane < 12 ? lane : 11;
    s.dc = f.dc[l >> 2][(size_t)((it.pos >> 16) + ((l >> 1) & 1)) * f.xsb + (it.pos & 0xffffu) + (l & 1)];
  };
  if (vb >= n) return;
  Staged16<CT> cur;
  fetch(item(vb), cur);
  WorkItem it_next = item(vb + stride);
  const float rx = m == 0 ? kResampleUpHost[2] : kResampleUpHost[3];
  const bool llf_lane = h == 0 && m < 2;
  for (; vb < n; vb += stride) {
    Staged16<CT> nxt;
    fetch(it_next, nxt);  // (a repeat of the last varblock past the end of the list: loaded, never used)
    it_next = item(vb + 2 * stride);
    const BlockHdr hd = MakeHdr(f, cur.it);
    int tab_at = m * 20 + 4 * h;
    asm volatile("" : "+v"(tab_at));
    const float* tab = tab_lds + tab_at;
    // lowest frequencies from the 2x2 DC patch (LowestFrequenciesFromDC, dec_transforms-inl.h:691-818), the twelve
    // DC values broadcast from the lanes that loaded them: lane (m = x < 2, h = 0) holds the corner M[x][0..1]
    float llf[3][2];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float d[2][2];
#pragma unroll
      for (int y = 0; y < 2; y++)
#pragma unroll
        for (int x = 0; x < 2; x++)
          d[y][x] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur.dc), c * 4 + y * 2 + x));
      float dp[2][2];
#pragma unroll
      for (int x = 0; x < 2; x++) {
        float v[2] = {d[0][x], d[1][x]};
        DctReg<2>(v);
        dp[0][x] = 0.5f * v[0];
        dp[1][x] = 0.5f * v[1];
      }
#pragma unroll
      for (int y = 0; y < 2; y++) {
        float v[2] = {dp[y][0], dp[y][1]};
        DctReg<2>(v);
        const float val = 0.5f * (m == 0 ? v[0] : v[1]);
        llf[c][y] = val * rx * kResampleUpHost[2 + y];
      }
    }
    auto unpack = [&](const Raw& r, int32_t* q) {
      if constexpr (sizeof(CT) == 2) {
        q[0] = (int32_t)(int16_t)(r.x & 0xffffu);
        q[1] = (int32_t)r.x >> 16;
        q[2] = (int32_t)(int16_t)(r.y & 0xffffu);
        q[3] = (int32_t)r.y >> 16;
      } else {
        q[0] = (int32_t)r.x;
        q[1] = (int32_t)r.y;
        q[2] = (int32_t)r.z;
        q[3] = (int32_t)r.w;
      }
    };
    float vy[4];
    {
      int32_t q[4];
      unpack(cur.raw[1], q);
      const float4 t = *(const float4*)(tab + 16 * 20);
      const float tk[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; k++) vy[k] = AdjustQuantBias(q[k], f.biases[1], f.biases[3]) * (tk[k] * hd.sy);
    }
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
      float v[4];
      if (c == 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = vy[k];
      } else {
        const float sc = c == 0 ? hd.sx : hd.sb;
        const float cc = c == 0 ? hd.x_cc : hd.b_cc;
        int32_t q[4];
        unpack(cur.raw[c], q);
        const float4 t = *(const float4*)(tab + c * 16 * 20);
        const float tk[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float d = AdjustQuantBias(q[k], f.biases[c], f.biases[3]) * (tk[k] * sc);
          v[k] = __builtin_fmaf(cc, vy[k], d);
        }
      }
      if (llf_lane) {
        v[0] = llf[c][0];
        v[1] = llf[c][1];
      }
      v4f q = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 3; kk >= 0; kk--) q = __builtin_amdgcn_mfma_f32_16x16x4f32(v[kk], b1[kk], q, 0, 0, 0);
      v4f p = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; i++) p = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[i], q[i], p, 0, 0, 0);
      // lane (y = m, h): pixels (row y, columns 4 h .. 4 h + 3)
      float* dst = TilePtr(f, c, hd.aby + (m >> 3), hd.abx + (h >> 1)) + (m & 7) * 8 + 4 * (h & 1);
      *(float4*)dst = make_float4(p[0], p[1], p[2], p[3]);
    }
    cur = nxt;
  }
}

}  // namespace

// bc16[step * 64 + lane]: B[4 h + step][n] of the 16-point IDCT, lane = (n = lane % 16, h = lane / 16) -- the B
// operand of product 1 and the A operand of product 2
void MfmaDct16Constants(float* host /* 256 floats */) {
  for (int k = 0; k < 16; k++) {
    double v[16] = {0};
    v[k] = 1.0;
    IdctHost(v, 16);
    for (int n = 0; n < 16; n++) host[(k & 3) * 64 + (k >> 2) * 16 + n] = (float)v[n];
  }
}

void LaunchMfma16(const DevFrame& f, const WorkLists& wl, uint32_t cells, hipStream_t st) {
  const int cls = kClsMedium0 + 2;  // DCT16X16 (kMediumStrategy[2] == 4)
  uint32_t grid = cells / 4 / 4 + 1;  // varblocks / 4 waves
  if (grid > 4096u) grid = 4096u;
  if (f.coeff_type == JXLHIP_COEFF_I16)
    hipLaunchKernelGGL((k_transform_mfma16<int16_t>), dim3(grid), dim3(256), 0, st, f, wl.list[cls],
                       wl.count + cls * kCounterPad, f.mfma16);
  else
    hipLaunchKernelGGL((k_transform_mfma16<int32_t>), dim3(grid), dim3(256), 0, st, f, wl.list[cls],
                       wl.count + cls * kCounterPad, f.mfma16);
}

// bc[0 .. 1023]: B operand of product 1, step kk: lane (n, h) -> B[16 h + kk][n]
// bc[1024 ..  ]: A operand of product 2, step i : lane (x, h) -> B[4 h + i % 4 + 8 (i / 4)][x]
void MfmaDct32Constants(float* host /* 2048 floats */) {
  double B[32][32];
  for (int k = 0; k < 32; k++) {
    double v[32] = {0};
    v[k] = 1.0;
    IdctHost(v, 32);
    for (int n = 0; n < 32; n++) B[k][n] = v[n];
  }
  for (int i = 0; i < 16; i++)
    for (int lane = 0; lane < 64; lane++) {
      const int n = lane & 31, h = lane >> 5;
      host[i * 64 + lane] = (float)B[16 * h + i][n];
      host[1024 + i * 64 + lane] = (float)B[4 * h + (i % 4) + 8 * (i / 4)][n];
    }
}

void LaunchMfma32(const DevFrame& f, const WorkLists& wl, uint32_t cells, hipStream_t st, const FilterParams* emit) {
  const float* bc = f.mfma32;
  const int cls = kClsMedium0 + 7;  // DCT32X32 (kMediumStrategy[7] == 5)
  uint32_t grid = cells / 16 / 4 + 1;  // varblocks / 4 waves
  if (grid > 2048u) grid = 2048u;
  const FilterParams none{};
#define JXLHIP_MFMA(CT, E)                                                                                  \
  hipLaunchKernelGGL((k_transform_mfma32<CT, E>), dim3(grid), dim3(256), 0, st, f, wl.list[cls],            \
                     wl.count + cls * kCounterPad, bc, E ? *emit : none)
  if (f.coeff_type == JXLHIP_COEFF_I16) {
    if (emit) JXLHIP_MFMA(int16_t, true);
    else JXLHIP_MFMA(int16_t, false);
  } else {
    if (emit) JXLHIP_MFMA(int32_t, true);
    else JXLHIP_MFMA(int32_t, false);
  }
#undef JXLHIP_MFMA
}

}  // namespace jxlhip
