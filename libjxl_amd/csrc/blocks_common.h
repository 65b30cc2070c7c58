// blocks_common.h -- pieces of the transform kernels shared by kernels_blocks.hip and the fused
// kernel (kernels_fused.hip): the per-varblock header and the row-per-lane DCT8 building blocks.
#ifndef JXLHIP_BLOCKS_COMMON_H_
#define JXLHIP_BLOCKS_COMMON_H_

#include "dev_common.h"
#include "kernels.h"

namespace jxlhip {

// ------------------------------------------------------------ block header
// threadIdx.x through an opaque (volatile, empty) asm: the unit functions below run inside the
// persistent loop of UnitDispatch, and without this every class's lane-dependent address
// arithmetic is loop-invariant and gets hoisted out of the loop -- into registers that stay
// live across ALL classes of the family (measured: 308 VGPRs).
__device__ __forceinline__ int Tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

struct BlockHdr {
  uint32_t abx, aby;
  size_t coef;  // element offset into coeffs[c]
  float sx, sy, sb, x_cc, b_cc;
};

__device__ __forceinline__ BlockHdr MakeHdr(const DevFrame& f, const WorkItem it) {
  BlockHdr h;
  h.abx = it.pos & 0xffffu;
  h.aby = it.pos >> 16;
  h.coef = (size_t)it.off * 64u;
  const int quant = (int)(it.qc & 0xffffu);
  const float s = f.inv_global_scale / (float)quant;  // dec_group.cc:164
  h.sx = s * f.x_dm;
  h.sy = s;
  h.sb = s * f.b_dm;
  h.x_cc = f.cfl_base_x + (float)(int8_t)((it.qc >> 16) & 0xffu) * f.color_scale;
  h.b_cc = f.cfl_base_b + (float)(int8_t)(it.qc >> 24) * f.color_scale;
  return h;
}

// ------------------------------------------------------------------ k_dct8
// DCT8 alone is ~45 % of a d1.0 frame and gets a kernel of its own with no LDS and few
// registers.  EIGHT LANES share a block:
//   lane = block-of-the-step (bits 0-2) | matrix row j (bits 3-5)
// Lane (b, j) loads row j of the stored 8x8 coefficient matrix of all three channels (one
// 16-byte load each: the block's 8 lanes fetch its whole 128-byte line), dequantises it with
// chroma-from-luma, runs the 8-point IDCT along the row, the wave transposes the 8x8 matrix
// across the 8 lanes in registers (v_permlane32_swap for lane bit 5, v_permlane16_swap for
// bit 4, DPP row_ror:8 + selects for bit 3), a second IDCT, and the lane holds pixel row j:
// two 16-byte stores, the block's 8 lanes writing its 256-byte tile.
// The two 1-D passes run in the opposite order of IDCT2D (dct-inl.h:254-291), which needs the
// transpose once instead of three times; the 1-D transform itself keeps the reference's
// operation order, the result differs from the other order by rounding only.
__device__ __forceinline__ void SwapHalves32(float& a, float& b) {  // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void SwapRows16(float& a, float& b) {  // odd rows of a <-> even rows of b
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// element (row j, register a) -> (row a, register j) for the lane mapping above
__device__ __forceinline__ void Transpose8Lanes(float* w, bool bit3) {
#pragma unroll
  for (int k = 0; k < 4; k++) SwapHalves32(w[k], w[k + 4]);
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (!(k & 2)) SwapRows16(w[k], w[k + 2]);
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const float send = bit3 ? w[k] : w[k + 1];
    const float recv =
        __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));
    w[k] = bit3 ? recv : w[k];
    w[k + 1] = bit3 ? w[k + 1] : recv;
  }
}

template <typename CT>
struct Dct8Row {  // one matrix row of one channel as loaded
  static constexpr int kVec = sizeof(CT) == 2 ? 1 : 2;
  uint4 v[kVec];
  __device__ __forceinline__ void Load(const void* base, size_t elem) {
    const uint4* p = (const uint4*)((const CT*)base + elem);
#pragma unroll
    for (int i = 0; i < kVec; i++) v[i] = p[i];
  }
  __device__ __forceinline__ void Unpack(int32_t* q) const {
    if constexpr (sizeof(CT) == 2) {
      const uint32_t w[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        q[2 * i] = (int32_t)(int16_t)(w[i] & 0xffffu);
        q[2 * i + 1] = (int32_t)w[i] >> 16;
      }
    } else {
      const uint32_t w[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = (int32_t)w[i];
    }
  }
};


}  // namespace jxlhip
#endif
