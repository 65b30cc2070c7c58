// kernels_fused.hip -- the loop-filter row march fed straight from the coefficient stream.
//
// Round 1's two phases move the frame's XYB planes through HBM twice (12 B/px written by the
// transform kernels, 12 B/px read by the filter kernel: 0.8 of the 1.5 GB an 8K frame moved).  Here
// the filter wave decodes the varblocks of its own 128-column window itself, one block row (8 pixel
// rows) at a time, into a 12 KB per-wave LDS slab and marches over the slab -- those pixels never
// exist in HBM.  What a wave can decode alone without holding more than 8 rows is the single-block
// class that dominates a d1.0 frame, DCT8 (45 % of the area, row-per-lane: 8 lanes per block,
// register transposes, as k_transform_8); every other varblock is still decoded by the class-sorted
// kernels of phase 1 into the block-major planes, and the wave copies those cells' tiles into its
// slab with LDS-DMA loads (global_load_lds_dword: HBM -> LDS without passing through registers).
//
//   wave window : 16 whole block columns [112 k - 8, 112 k + 120): 8 halo columns either side of the
//                 112 output columns (the march needs <= 4), so that the window is made of whole
//                 varblock cells; neighbouring windows share two block columns, decoded by both
//   per group of 8 rows (r = 0 mod 8):
//     Fill      : cell info of the 16 cells (k_prepare left coefficient offset + quant / CfL word of
//                 every DCT8 block in DevFrame::cell_info; other cells say "from the planes");
//                 LDS-DMA of the plane cells (24 instructions per 8 cells); DCT8 cells in steps of 8
//                 blocks: dequant + CfL + IDCT x transpose x IDCT, two ds_write_b128 per channel
//     8 x Step  : the row march of filters_march.h with SRC_LDS (ds_read_b64 per channel and row)
//   mirroring   : columns through the lane's (mirrored) slab address; rows by the slab row index
//                 (a group of 8 rows always lies in one block row: frames with 1..3 rows in their
//                 last block row go to the two-phase path)
//
// The reference semantics are those of the two-phase path (simple_render_pipeline.cc:129-164,
// loop_filter.h:26-29); the parity tests run both.
#include <stdlib.h>

#include <type_traits>

#include "blocks_common.h"
#include "env_switches.h"
#include "filters_march.h"
// Two translation units (this file compiles for minutes): kernels_fused.hip itself (part 2: k_fused_pc and the entry
// points) and kernels_fused_epf0.hip (part 3: k_fused_pc0, the epf_iters = 3 form), which includes this file.
#ifndef JXLHIP_FUSED_PART
#define JXLHIP_FUSED_PART 2
#endif
#if JXLHIP_FUSED_PART == 3
#include "epf0_march.h"
#endif
// The producing wave runs at a raised wave priority (s_setprio): the marching wave waits for it at every block row's
// barrier, and at equal priority the SIMD's arbiter lets the (longer, never-waiting) marches of OTHER windows take the
// issue slots a producer needs to finish its block row (8K d1.0: k_fused_pc 203.5 -> 196 us, profiles/r04_setprio.txt).
static constexpr int kProducerPrio = 3;

namespace jxlhip {

namespace {

typedef __attribute__((address_space(3))) float LdsF;
typedef __attribute__((address_space(3))) uint32_t LdsU;

static constexpr int kFusedHalo = 8;                      // window columns in front of the first output column
static constexpr int kFusedUse = kSlabCols - 2 * kFusedHalo;  // 112 output columns per wave

// The fill steps read their frame parameters (coefficient / DC / table / plane pointers, quantizer
// scalars) from the KERNARG SEGMENT at the point of use instead of keeping them in SGPRs for the whole
// march: the march alone wants the ~100 SGPRs a wave has, and what does not fit is spilled to VGPR
// lanes and read back with v_readlane_b32 -- 14 VALU issues per row step in the first version of this
// kernel.  (DevFrame is the kernel's first by-value argument: offset 0 of the segment.)  The pointer
// is laundered through an empty asm so that the loads are not hoisted out of the row loop again.
typedef const DevFrame __attribute__((address_space(4))) * FrameArgs;
__device__ __forceinline__ FrameArgs Fresh(FrameArgs q) {
  asm volatile("" : "+s"(q));
  return q;
}

// What a wave knows about the block row it fills next: the cell info of its 16 cells (lanes 0..15)
// and, once that load has returned, which cells it decodes itself / copies from the planes.
struct NextRow {
  int nb;        // block row, -1 = none
  uint2 ci;      // lanes 0..15: cell info
  uint32_t m8;   // wave-uniform: DCT8 cells
  uint32_t mp;   // wave-uniform: cells whose tiles come from the planes
};

__device__ __forceinline__ void NextRowRequest(FrameArgs fa, NextRow& n, int nb, int bc0) {
  const FrameArgs f = Fresh(fa);
  const int lane = threadIdx.x & 63;
  const int c16 = bc0 + (lane & 15);
  n.nb = nb;
  n.ci = make_uint2(kCellFromPlanes, 0u);
  if (nb >= 0 && lane < 16 && c16 >= 0 && c16 < (int)f->xsb) n.ci = f->cell_info[(size_t)nb * f->xsb + c16];
}
__device__ __forceinline__ void NextRowMasks(FrameArgs fa, NextRow& n, int bc0) {
  const FrameArgs f = Fresh(fa);
  const int lane = threadIdx.x & 63;
  const int c16 = bc0 + (lane & 15);
  const bool valid_cell = n.nb >= 0 && lane < 16 && c16 >= 0 && c16 < (int)f->xsb;
  const bool is_dct8 = valid_cell && n.ci.x != kCellFromPlanes;
  n.m8 = (uint32_t)__ballot(is_dct8) & 0xffffu;
  n.mp = (uint32_t)__ballot(valid_cell && !is_dct8) & 0xffffu;
}

// Plane cells of block row n.nb, slab rows 2k and 2k+1, all three channels: three LDS-DMA
// instructions of 1 KB (global_load_lds_dwordx4: lane l brings 16 bytes = columns 4 (l & 31) ..
// of row 2k + (l >> 5); the LDS destination of a wave instruction is contiguous, which two
// consecutive 128-float slab rows are).  Called when rows 2k, 2k+1 of the CURRENT block row have
// been consumed: the copy overlaps the march over the remaining rows.
__device__ __forceinline__ void DmaPlaneRows(FrameArgs fa, LdsF* slab, const NextRow& n, int bc0, int k) {
  if (n.mp == 0) return;  // wave-uniform
  const FrameArgs f = Fresh(fa);
  const int lane = threadIdx.x & 63;
  const int cell = (lane & 31) >> 1;
  if ((n.mp >> cell) & 1u) {
    const int row = 2 * k + (lane >> 5);
    const size_t at = ((size_t)(n.nb - (f->plane_y0 >> 3)) * f->tile_stride + (size_t)(bc0 + cell)) * 64u + row * 8 + (lane & 1) * 4;
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
      __builtin_amdgcn_global_load_lds(f->xyb[ch] + at, slab + ch * kSlabPlaneFloats + 2 * k * kSlabCols, 16, 0, 0);
  }
}

// The rest of a block row's fill: the last two plane rows and the DCT8 cells -- the row-per-lane scheme
// of k_transform_8 (kernels_blocks.hip), 8 blocks per step: dequant + CfL, IDCT, register transpose,
// IDCT, two ds_write_b128 per channel.  nb: block row (inside the frame), bc0: block column of slab
// column 0 (-1 at the left edge; cells outside the frame are skipped: no lane reads their columns).
template <typename CT>
__device__ __forceinline__ void FinishSlab(FrameArgs fa, LdsF* slab, LdsU* list, const NextRow& n, int bc0, int first_plane_k) {
  const FrameArgs f = Fresh(fa);
  const int lane = threadIdx.x & 63;
  const int nb = n.nb;
  // every row of the previous block row has been read (this wave's own ds_write stay in order behind
  // the reads; the DMA writes come through the vector memory path: wait for the reads explicitly)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int k = first_plane_k; k < 4; k++) DmaPlaneRows(fa, slab, n, bc0, k);
  const uint32_t m8 = n.m8;
                           // ~50 % busy; a wave cannot overlap its own fill with its own march, and registers + LDS
                           // cap the SIMD at three waves)
  if (m8) {  // wave-uniform
    const bool is_dct8 = lane < 16 && ((m8 >> lane) & 1u);
    if (is_dct8) {
      const uint32_t rank = __builtin_popcount(m8 & ((1u << lane) - 1u));
      list[rank * 4 + 0] = (uint32_t)lane;
      list[rank * 4 + 1] = n.ci.x;
      list[rank * 4 + 2] = n.ci.y;
    }
    const int n8 = __builtin_popcount(m8);
    const int j = lane >> 3;  // matrix row (input), pixel row (output)
    const bool bit3 = (lane & 8) != 0;
    for (int first = 0; first < n8; first += 8) {
      const int b = first + (lane & 7);
      const bool valid = b < n8;
      const int bb = valid ? b : n8 - 1;
      const int cell = (int)list[bb * 4 + 0];
      WorkItem it;
      it.pos = ((uint32_t)nb << 16) | (uint32_t)(bc0 + cell);
      it.off = list[bb * 4 + 1];
      it.qc = list[bb * 4 + 2];
      it.pad = 0;
      const size_t elem = (size_t)it.off * 64u + (size_t)j * 8u;
      const size_t dc_at = (size_t)nb * f->xsb + (size_t)(bc0 + cell);
      Dct8Row<CT> rows[3];
      float dcv[3];
      float tab[3][8];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        rows[c].Load(f->coeffs[c], elem);
        dcv[c] = f->dc[c][dc_at];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float4 t0 = *(const float4*)(f->dequant + c * 64 + j * 8);
        const float4 t1 = *(const float4*)(f->dequant + c * 64 + j * 8 + 4);
        tab[c][0] = t0.x, tab[c][1] = t0.y, tab[c][2] = t0.z, tab[c][3] = t0.w;
        tab[c][4] = t1.x, tab[c][5] = t1.y, tab[c][6] = t1.z, tab[c][7] = t1.w;
      }
      BlockHdr h;  // MakeHdr (blocks_common.h) on the kernarg copy of the frame
      {
        const int quant = (int)(it.qc & 0xffffu);
        const float sq = f->inv_global_scale / (float)quant;  // dec_group.cc:164
        h.sx = sq * f->x_dm;
        h.sy = sq;
        h.sb = sq * f->b_dm;
        h.x_cc = f->cfl_base_x + (float)(int8_t)((it.qc >> 16) & 0xffu) * f->color_scale;
        h.b_cc = f->cfl_base_b + (float)(int8_t)(it.qc >> 24) * f->color_scale;
      }
      const float bias0 = f->biases[0], bias1 = f->biases[1], bias2 = f->biases[2], bias3 = f->biases[3];
      int32_t q[8];
      float vy[8];
      rows[1].Unpack(q);
#pragma unroll
      for (int k = 0; k < 8; k++) vy[k] = AdjustQuantBias(q[k], bias1, bias3) * (tab[1][k] * h.sy);
#pragma unroll
      for (int ci3 = 0; ci3 < 3; ci3++) {
        const int c = ci3 == 0 ? 1 : (ci3 == 1 ? 0 : 2);
        float v[8];
        if (c == 1) {
#pragma unroll
          for (int k = 0; k < 8; k++) v[k] = vy[k];
        } else {
          const float sc = c == 0 ? h.sx : h.sb;
          const float cc = c == 0 ? h.x_cc : h.b_cc;
          rows[c].Unpack(q);
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const float d = AdjustQuantBias(q[k], c == 0 ? bias0 : bias2, bias3) * (tab[c][k] * sc);
            v[k] = __builtin_fmaf(cc, vy[k], d);
          }
        }
        if (j == 0) v[0] = dcv[c];
        IdctReg<8>(v);
        Transpose8Lanes(v, bit3);
        IdctReg<8>(v);
        if (valid) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          typedef f4v __attribute__((address_space(3))) * P4;
          LdsF* dst = slab + c * kSlabPlaneFloats + j * kSlabCols + cell * 8;
          *(P4)dst = f4v{v[0], v[1], v[2], v[3]};
          *(P4)(dst + 4) = f4v{v[4], v[5], v[6], v[7]};
        }
      }
    }
  }
  // the LDS-DMA loads count in vmcnt; this wave's ds_write / ds_read stay in order by themselves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// block row the 8 rows of the group starting at row r come from (also when they are mirror rows: see the header)
__device__ __forceinline__ int GroupBlockRow(int r, int nb_last) {
  const int nb = r < 0 ? 0 : (r >> 3);
  return nb > nb_last ? nb_last : nb;
}

// ------------------------------------------------------------------------------------------------
// k_fused_pc: the same window march with the two halves of the work on two WAVES of a workgroup.
//
// On gfx9 (gfx950 included) a wave has ONE counter, vmcnt, for its vector-memory loads AND stores, and it
// retires in issue order: waiting for a load that was issued behind N output stores waits for the write
// acknowledgements of those N stores first.  The single-wave kernel above waits like that once or twice per
// block row (the inv_sigma load, the vmcnt(0) that ends a fill) -- every group of 8 rows pays the latency of its
// own output stores on top of its loads, and the fill can only start when the march has let go of the slab.
// Here
//   wave 0 (march)  : reads its rows and its inv_sigma values from LDS, computes, stores pixels.  It issues NO
//                     vector-memory load, so it never waits on vmcnt: the stores just queue.
//   wave 1 (produce): fills block row i+1 into the other half of a double-buffered slab (cell info, LDS-DMA of
//                     the plane cells, in-wave DCT8 decode, inv_sigma row) while wave 0 marches over block row i.
//                     It issues no store, so its vmcnt waits cover loads only.
// One s_barrier per block row joins the two (fill(i) done / march(i-1) done).  Workgroup = 128 threads = one
// window; six workgroups per CU (three waves per SIMD by registers, 24.7 KB of LDS each).
static constexpr int kPcWaves = 3;   // waves per SIMD the kernels below are compiled for (<= 168 VGPRs)
static constexpr int kPcPerCu = 6;   // windows resident per CU
template <int NB>
struct __attribute__((aligned(16))) StripLdsT {
  float slab[NB][3 * kSlabPlaneFloats];  // [buffer][channel][row 0..7][column 0..127]
  float sigma[NB][16];                   // [buffer][cell]: inv_sigma of the block row's 16 cells (columns clamped into the frame)
  uint32_t list[NB - 1][16 * 4];         // per producer: the DCT8 cells of the block row being filled
};
typedef StripLdsT<2> StripLds;

// groups of 8 rows a window chunk [y_begin, y_end) walks over: [head (HX rows of the block row above)] + whole
// groups + [tail (HX rows of the block row below)]; group i starts at image row r_first + 8 i
template <int HX>
__device__ __forceinline__ int PcGroups(int y_begin, int y_end) {
  const int whole = (y_end - y_begin + 7) >> 3;
  const bool tail = HX > 0 && y_begin + 8 * whole <= y_end + HX - 1;
  return (HX > 0 ? 1 : 0) + whole + (tail ? 1 : 0);
}

__device__ __forceinline__ void PcBarrierProducer() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void PcBarrierMarch() {  // no vmcnt: the output stores stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int HX, typename CT>
__device__ __forceinline__ void ProducePC(FrameArgs fa, StripLds* w, int bc0, int y_begin, int y_end, int nb_last) {
  const int lane = threadIdx.x & 63;
  const int r_first = HX ? y_begin - 8 : y_begin;
  const int G = PcGroups<HX>(y_begin, y_end);
  auto sigma_request = [&](int nb) -> float {
    const FrameArgs f = Fresh(fa);
    const int xsb = (int)f->xsb;
    int col = bc0 + (lane & 15);
    col = col < 0 ? 0 : (col >= xsb ? xsb - 1 : col);
    return lane < 16 ? f->inv_sigma[(size_t)nb * xsb + col] : 0.0f;
  };
  NextRow nx;
  int nb = GroupBlockRow(r_first, nb_last);
  NextRowRequest(fa, nx, nb, bc0);
  float sg = sigma_request(nb);
  for (int i = 0; i < G; i++) {
    NextRowMasks(fa, nx, bc0);  // needs the cell info: the first wait of this fill
    const NextRow cur = nx;
    const float sg_cur = sg;
    if (i + 1 < G) {  // the next block row's cell info / sigma travel while this one is decoded
      nb = GroupBlockRow(r_first + 8 * (i + 1), nb_last);
      NextRowRequest(fa, nx, nb, bc0);
      sg = sigma_request(nb);
    }
    LdsF* slab = (LdsF*)w->slab[(i & 1)];
    FinishSlab<CT>(fa, slab, (LdsU*)w->list[0], cur, bc0, 0);  // four plane row pairs by LDS-DMA + the DCT8 cells; ends on vmcnt(0)
    if (lane < 16) ((LdsF*)w->sigma[i & 1])[lane] = sg_cur;
    PcBarrierProducer();
  }
}

// The producer as a software pipeline (16-bit coefficients): the coefficient rows and DC values of block row
// g+1 are requested -- into a second set of registers -- BEFORE block row g is decoded, and block row g+2's cell
// info with them, so that a fill is the decode arithmetic plus LDS writes and the vmcnt(0) in front of the
// barrier finds loads that have had the whole decode to arrive.  (ProducePC above starts every fill with two
// dependent round trips: cell info, then coefficients.)
//
// These loads are issued through inline asm on purpose: for a load it knows, the compiler places the s_waitcnt
// itself, and across this loop's control flow it falls back to vmcnt(0) in front of the first use -- which would
// wait for the prefetch that was just issued.  An asm load's result is "ready" as far as the compiler is concerned;
// the ONLY wait is the vmcnt(0) of PcBarrierProducer, and every loaded register is first used behind it (the two
// register sets alternate through a loop unrolled by two: no copies).
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u4v AsmLoad4(const void* p) {
  u4v r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ u2v AsmLoad2(const void* p) {
  u2v r;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ uint32_t AsmLoad1(const void* p) {
  uint32_t r;
  asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}

struct PcStepRegs {  // one decode step = up to 8 DCT8 cells, lane = (cell of the step: bits 0-2, matrix row: bits 3-5)
  u4v rows[3];       // the lane's row of 8 coefficients, per channel
  uint32_t dcv[3];   // DC of the lane's block (float bits)
  uint32_t qc;
  int cell;          // window cell 0..15 of the lane's block
};
struct PcGroupRegs {
  PcStepRegs st[2];
  int n8;       // DCT8 cells of the block row (wave-uniform)
  uint32_t mp;  // cells copied from the planes (wave-uniform)
  int nb;       // block row
};
struct PcNext {  // cell info + inv_sigma of a block row, as requested (valid behind the next barrier)
  u2v ci;
  uint32_t sg;
  int nb;
};

// Frame constants of the producing wave, read ONCE from the kernarg segment (round 5).  Rounds 2-4 re-read every field
// at its point of use (Fresh above): right for the single-wave kernel, whose march wants every SGPR -- but the
// producing wave of k_fused_pc runs no march, and each re-read was an s_load + s_waitcnt lgkmcnt(0) in the middle of
// the fill (19 scalar round trips per block row).  The wave's ~35 SGPRs of constants stay resident instead, and every
// load is SGPR base + 32-bit lane offset (no 64-bit VALU address arithmetic: 32 v_lshl_add_u64 per decode step before).
struct PcK {
  const char* coef[3];
  const char* dc[3];
  const char* cell_info;
  const char* inv_sigma;
  const float* xyb[3];
  int xsb;
  uint32_t tile_stride;
  int plane_tile_row0;
  float inv_global_scale, x_dm, b_dm, cfl_base_x, cfl_base_b, color_scale;
  float bias[4];
};
__device__ __forceinline__ PcK MakePcK(FrameArgs f) {
  PcK k;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    k.coef[c] = (const char*)f->coeffs[c];
    k.dc[c] = (const char*)f->dc[c];
    k.xyb[c] = f->xyb[c];
  }
  k.cell_info = (const char*)f->cell_info;
  k.inv_sigma = (const char*)f->inv_sigma;
  k.xsb = (int)f->xsb;
  k.tile_stride = f->tile_stride;
  k.plane_tile_row0 = f->plane_y0 >> 3;
  k.inv_global_scale = f->inv_global_scale;
  k.x_dm = f->x_dm;
  k.b_dm = f->b_dm;
  k.cfl_base_x = f->cfl_base_x;
  k.cfl_base_b = f->cfl_base_b;
  k.color_scale = f->color_scale;
#pragma unroll
  for (int i = 0; i < 4; i++) k.bias[i] = f->biases[i];
  return k;
}
// loads at SGPR base + unsigned 32-bit lane offset; no compiler-placed wait (see above)
__device__ __forceinline__ u4v AsmLoad4S(const char* base, uint32_t off) {
  u4v r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
  return r;
}
__device__ __forceinline__ u2v AsmLoad2S(const char* base, uint32_t off) {
  u2v r;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
  return r;
}
__device__ __forceinline__ uint32_t AsmLoad1S(const char* base, uint32_t off) {
  uint32_t r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
  return r;
}

// "These registers are defined HERE": placed right behind the wait (s_waitcnt vmcnt(0) / the barrier) that the data of
// the asm loads above has landed at.  An asm load's result looks ready to the compiler from the load on, and nothing but
// a data dependency keeps it from scheduling a plain VALU use of the register above the (volatile, but register-free)
// wait -- it did exactly that with the first build of this round (a v_cmp of the cell info in front of the prologue's
// s_waitcnt: a memory fault).  Volatile asm statements keep their order, so every use below depends on the wait.
__device__ __forceinline__ void PcLanded(PcNext& n) { asm volatile("" : "+v"(n.ci), "+v"(n.sg)); }
__device__ __forceinline__ void PcLanded(PcGroupRegs& R) {
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int c = 0; c < 3; c++) asm volatile("" : "+v"(R.st[s].rows[c]), "+v"(R.st[s].dcv[c]));
}

__device__ __forceinline__ void PcRequest(const PcK& K, PcNext& n, int nb, int bc0) {
  const int lane = threadIdx.x & 63;
  int col = bc0 + (lane & 15);
  col = col < 0 ? 0 : (col >= K.xsb ? K.xsb - 1 : col);  // lanes 16..63 and cells outside the frame: any valid address
  const uint32_t cell = (uint32_t)(nb * K.xsb + col);  // (whole-frame cell index: < 2^26)
  n.nb = nb;
  n.ci = AsmLoad2S(K.cell_info, cell * 8u);
  n.sg = AsmLoad1S(K.inv_sigma, cell * 4u);
}

// n's registers are valid (a barrier has passed since PcRequest): masks, the DCT8 list, and the loads of the block
// row's coefficient rows / DC values into R
__device__ __forceinline__ void PcIssue(const PcK& K, LdsU* list, const PcNext& n, int bc0, PcGroupRegs& R) {
  const int lane = threadIdx.x & 63;
  const int c16 = bc0 + (lane & 15);
  const bool valid_cell = lane < 16 && c16 >= 0 && c16 < K.xsb;
  const bool is_dct8 = valid_cell && n.ci.x != kCellFromPlanes;
  const uint32_t m8 = (uint32_t)__ballot(is_dct8) & 0xffffu;
  R.mp = (uint32_t)__ballot(valid_cell && !is_dct8) & 0xffffu;
  R.n8 = __builtin_popcount(m8);
  R.nb = n.nb;
  if (m8 == 0) return;  // wave-uniform
  if (is_dct8) {
    const uint32_t rank = __builtin_popcount(m8 & ((1u << lane) - 1u));
    list[rank * 4 + 0] = (uint32_t)lane;
    list[rank * 4 + 1] = n.ci.x;
    list[rank * 4 + 2] = n.ci.y;
  }
  const int j = lane >> 3;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    if (s * 8 < R.n8) {  // wave-uniform
      const int b = s * 8 + (lane & 7);
      const int bb = b < R.n8 ? b : R.n8 - 1;
      const int cell = (int)list[bb * 4 + 0];
      const uint32_t off = list[bb * 4 + 1];
      R.st[s].qc = list[bb * 4 + 2];
      R.st[s].cell = cell;
      // 16-bit coefficients: 128 bytes per block, 16 per matrix row (the offset stays below 2^32: a channel's
      // coefficient buffer is frame pixels x 2 bytes)
      const uint32_t elem_bytes = off * 128u + (uint32_t)j * 16u;
      const uint32_t dc_bytes = (uint32_t)(n.nb * K.xsb + bc0 + cell) * 4u;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        R.st[s].rows[c] = AsmLoad4S(K.coef[c], elem_bytes);
        R.st[s].dcv[c] = AsmLoad1S(K.dc[c], dc_bytes);
      }
    }
  }
}

// Plane cells of block row nb, the four slab row pairs, all three channels: twelve LDS-DMA instructions of 1 KB
// (DmaPlaneRows above, with the producer's resident constants)
__device__ __forceinline__ void PcDmaPlanes(const PcK& K, LdsF* slab, uint32_t mp, int nb, int bc0) {
  if (mp == 0) return;  // wave-uniform
  const int lane = threadIdx.x & 63;
  const int cell = (lane & 31) >> 1;
  if ((mp >> cell) & 1u) {
    const uint32_t tile = (uint32_t)(nb - K.plane_tile_row0) * K.tile_stride + (uint32_t)(bc0 + cell);
    const uint32_t at0 = tile * 64u + (uint32_t)(lane >> 5) * 8u + (uint32_t)(lane & 1) * 4u;  // floats; < 2^30 (FusedSupported)
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int ch = 0; ch < 3; ch++)
        __builtin_amdgcn_global_load_lds(K.xyb[ch] + (at0 + 16u * k), slab + ch * kSlabPlaneFloats + 2 * k * kSlabCols, 16, 0, 0);
  }
}

__device__ __forceinline__ void PcDecode(const PcK& K, LdsF* slab, const PcGroupRegs& R, const float (&tab)[3][8]) {
  const int lane = threadIdx.x & 63;
  const int j = lane >> 3;
  const bool bit3 = (lane & 8) != 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    if (s * 8 < R.n8) {  // wave-uniform
      const PcStepRegs& T = R.st[s];
      const bool valid = s * 8 + (lane & 7) < R.n8;
      float sx, sy, sb, x_cc, b_cc;
      {
        const int quant = (int)(T.qc & 0xffffu);
        const float sq = K.inv_global_scale / (float)quant;  // dec_group.cc:164
        sx = sq * K.x_dm;
        sy = sq;
        sb = sq * K.b_dm;
        x_cc = K.cfl_base_x + (float)(int8_t)((T.qc >> 16) & 0xffu) * K.color_scale;
        b_cc = K.cfl_base_b + (float)(int8_t)(T.qc >> 24) * K.color_scale;
      }
      const float bias0 = K.bias[0], bias1 = K.bias[1], bias2 = K.bias[2], bias3 = K.bias[3];
      auto unpack = [](const u4v r, int32_t* q) {
        const uint32_t wv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          q[2 * i] = (int32_t)(int16_t)(wv[i] & 0xffffu);
          q[2 * i + 1] = (int32_t)wv[i] >> 16;
        }
      };
      int32_t q[8];
      float vy[8];
      unpack(T.rows[1], q);
#pragma unroll
      for (int k = 0; k < 8; k++) vy[k] = AdjustQuantBias(q[k], bias1, bias3) * (tab[1][k] * sy);
      typedef float f4v __attribute__((ext_vector_type(4)));
      typedef f4v __attribute__((address_space(3))) * P4;
#pragma unroll
      for (int ci3 = 0; ci3 < 3; ci3++) {
        const int c = ci3 == 0 ? 1 : (ci3 == 1 ? 0 : 2);
        float v[8];
        if (c == 1) {
#pragma unroll
          for (int k = 0; k < 8; k++) v[k] = vy[k];
        } else {
          const float sc = c == 0 ? sx : sb;
          const float cc = c == 0 ? x_cc : b_cc;
          unpack(T.rows[c], q);
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const float d = AdjustQuantBias(q[k], c == 0 ? bias0 : bias2, bias3) * (tab[c][k] * sc);
            v[k] = __builtin_fmaf(cc, vy[k], d);
          }
        }
        if (j == 0) v[0] = __uint_as_float(T.dcv[c]);
        IdctReg<8>(v);
        Transpose8Lanes(v, bit3);
        IdctReg<8>(v);
        LdsF* dst = slab + c * kSlabPlaneFloats + j * kSlabCols + T.cell * 8;
        if (valid) {
          *(P4)dst = f4v{v[0], v[1], v[2], v[3]};
          *(P4)(dst + 4) = f4v{v[4], v[5], v[6], v[7]};
        }
      }
    }
  }
}

template <int HX>
__device__ __forceinline__ void ProducePC2(FrameArgs fa, StripLds* w, int bc0, int y_begin, int y_end, int nb_last) {
  const int lane = threadIdx.x & 63;
  const int r_first = HX ? y_begin - 8 : y_begin;
  const int G = PcGroups<HX>(y_begin, y_end);
  const PcK K = MakePcK(fa);
  // this lane's 8 entries of the three DCT8 dequant matrices, once per wave (DequantLane, dec_group.cc:115-153)
  float tab[3][8];
  {
    const int j = lane >> 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float4 t0 = *(const float4*)(fa->dequant + c * 64 + j * 8);
      const float4 t1 = *(const float4*)(fa->dequant + c * 64 + j * 8 + 4);
      tab[c][0] = t0.x, tab[c][1] = t0.y, tab[c][2] = t0.z, tab[c][3] = t0.w;
      tab[c][4] = t1.x, tab[c][5] = t1.y, tab[c][6] = t1.z, tab[c][7] = t1.w;
    }
  }
  LdsU* list = (LdsU*)w->list[0];
  auto group_nb = [&](int g) { return GroupBlockRow(r_first + 8 * (g < G ? g : G - 1), nb_last); };
  PcGroupRegs A, B;
  PcNext n0, n1;
  uint32_t sg_a = 0, sg_b = 0;
  // prologue: block row 0's loads and block row 1's cell info, then everything has landed
  PcRequest(K, n0, group_nb(0), bc0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PcLanded(n0);
  PcIssue(K, list, n0, bc0, A);
  sg_a = n0.sg;
  PcRequest(K, n1, group_nb(1), bc0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PcLanded(A);
  PcLanded(n1);
  // one fill: CUR holds block row g (loaded behind an earlier barrier), NXT receives block row g+1,
  // nn = cell info of block row g+1 (valid), refilled with block row g+2's
  auto body = [&](int g, PcGroupRegs& CUR, PcGroupRegs& NXT, uint32_t& sg_cur, uint32_t& sg_nxt, PcNext& nn) {
    LdsF* slab = (LdsF*)w->slab[(g & 1)];  // free: the march left it before the previous barrier
    // the plane cells of block row g first: of everything this fill waits for at its barrier, these copies were the last
    // to be issued (behind the list round trip of PcIssue) -- now they have the whole decode to land
    PcDmaPlanes(K, slab, CUR.mp, CUR.nb, bc0);
    PcIssue(K, list, nn, bc0, NXT);     // block row g+1 (the last block row again behind the end: harmless)
    sg_nxt = nn.sg;
    PcRequest(K, nn, group_nb(g + 2), bc0);
    PcDecode(K, slab, CUR, tab);
    if (lane < 16) ((LdsU*)w->sigma[g & 1])[lane] = sg_cur;
    PcBarrierProducer();
    PcLanded(NXT);
    PcLanded(nn);
  };
  for (int g = 0; g < G; g += 2) {
    body(g, A, B, sg_a, sg_b, n1);
    if (g + 1 < G) body(g + 1, B, A, sg_b, sg_a, n1);
  }
}

// INTERIOR (wave-uniform, chosen by the kernel): the chunk starts and ends on block rows and touches neither the frame's
// top nor its bottom -- every row step then knows its place in the block row, whether it writes, and that no row is a
// mirror row at COMPILE time (filters_march.h, StepKnown): the scalar bookkeeping of the generic step, a third of the
// marching wave's instruction issues, is gone.  The first whole group is peeled: its first HX steps still complete rows
// of the chunk above (not written here), every later step of the chunk writes.
template <int GAB, int EPF, int OUTK, int FMT, bool EDGE, bool INTERIOR, int NB = 2>
__device__ __forceinline__ void MarchPC(const DevFrame& f, const FilterParams& P, Lane& L, StripLdsT<NB>* w, int bc0,
                                        int y_begin, int y_end) {
  constexpr int HX = MarchGeom<GAB, EPF>::HX;
  constexpr int KI = INTERIOR ? (int)kStepInterior : 0;                  // a step that writes nothing
  constexpr int KE = INTERIOR ? (int)(kStepInterior | kStepEmit) : 0;   // a step that writes its row
  constexpr int KF = INTERIOR ? (int)kStepFirst : 0;
  const int H = (int)f.ysize;
  const int r_first = HX ? y_begin - 8 : y_begin;
  const int r_last = y_end + HX - 1;
  const int nb_last = (H - 1) >> 3;
  const int G = PcGroups<HX>(y_begin, y_end);
  State s;
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) s.x[c][k] = v2f{0.0f, 0.0f};
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
  float sigma_last = -1.0f;
  const XybConsts KC = MakeXybConsts(P);
  const size_t out_row_bytes = OUTK == JXLHIP_OUT_XYB_PLANAR ? P.out_stride * 4 : P.out_stride;
  char* out_row = (char*)P.out + (ptrdiff_t)(r_first - HX - (int)f.y0) * (ptrdiff_t)out_row_bytes;
  const float __attribute__((address_space(3)))* const slab0 = L.slab;
  // the lane's cell inside the window, for the inv_sigma row the producer leaves in LDS
  const LdsF* sig0 = (const LdsF*)w->sigma[0] + ((int)(L.sx4 >> 2) - bc0);
  int i = 0;
  auto enter_group = [&](int g) -> float {  // after the barrier that publishes buffer g % NB
    const int b = NB == 2 ? (g & 1) : g % NB;
    L.slab = slab0 + b * (3 * kSlabPlaneFloats);
    return EPF ? sig0[b * 16] : 0.0f;
  };
#define JXLHIP_PSTEPK(K, KN)                                                                                          \
  Step<GAB, EPF, OUTK, FMT, K, EDGE, 0, SRC_LDS, KN>(s, r + K, f, P, L, 0, y_begin, y_end, inv_sigma_blk, \
                                                                 inv_sigma_blk2, out_row, KC, slab_y0, sigma_pre,     \
                                                                 sigma_prev);                                         \
  out_row += out_row_bytes
// a whole group of 8 rows starting at image row r (a multiple of 8): steps 0 .. HX-1 take KLOW, the others KHIGH
#define JXLHIP_PGROUP(KLOW, KHIGH)                                                       \
  {                                                                                      \
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;                                   \
    const float sigma_prev = sigma_last;                                                 \
    const float sigma_pre = enter_group(i);                                              \
    sigma_last = sigma_pre;                                                              \
    {                                                                                    \
      const int row0 = INTERIOR ? 0 : Mirror1(r, H) - slab_y0;                           \
      _Pragma("unroll") for (int c = 0; c < 3; c++) s.x[c][0] = LdsPair<EDGE>(L, c, row0); \
    }                                                                                    \
    JXLHIP_PSTEPK(0, (0 < HX ? (KLOW) : (KHIGH)));                                       \
    JXLHIP_PSTEPK(1, (1 < HX ? (KLOW) : (KHIGH)));                                       \
    JXLHIP_PSTEPK(2, (2 < HX ? (KLOW) : (KHIGH)));                                       \
    JXLHIP_PSTEPK(3, (3 < HX ? (KLOW) : (KHIGH)));                                       \
    JXLHIP_PSTEPK(4, (KHIGH));                                                           \
    JXLHIP_PSTEPK(5, (KHIGH));                                                           \
    JXLHIP_PSTEPK(6, (KHIGH));                                                           \
    JXLHIP_PSTEPK(7, (KHIGH));                                                           \
    i++;                                                                                 \
  }
  PcBarrierMarch();  // fill(0)
  if constexpr (HX > 0) {  // the last HX rows of the block row above
    const int r = r_first;
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;
    const float sigma_prev = sigma_last;
    const float sigma_pre = enter_group(i);
    sigma_last = sigma_pre;
    {
      const int row0 = INTERIOR ? 8 - HX : Mirror1(r + 8 - HX, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][8 - HX] = LdsPair<EDGE>(L, c, row0);
    }
    out_row += (8 - HX) * out_row_bytes;
    if constexpr (HX >= 4) { JXLHIP_PSTEPK(4, KI); }
    if constexpr (HX >= 3) { JXLHIP_PSTEPK(5, KI); }
    if constexpr (HX >= 2) { JXLHIP_PSTEPK(6, KI); }
    JXLHIP_PSTEPK(7, KI);
    i++;
    PcBarrierMarch();  // a whole group always follows
  }
  int r = HX ? y_begin : r_first;
  if constexpr (INTERIOR) {
    JXLHIP_PGROUP(KI | KF, KE | KF)
    if (i < G) PcBarrierMarch();
    for (r += 8; r < y_end; r += 8) {
      JXLHIP_PGROUP(KE, KE)
      if (i < G) PcBarrierMarch();
    }
  } else {
    for (; r_last - r >= HX; r += 8) {
      JXLHIP_PGROUP(0, 0)
      if (i < G) PcBarrierMarch();
    }
  }
  if (HX > 0 && (INTERIOR || r <= r_last)) {  // the first HX rows of the block row below
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;
    const float sigma_prev = sigma_last;
    const float sigma_pre = enter_group(i);
    sigma_last = sigma_pre;
    {
      const int row0 = INTERIOR ? 0 : Mirror1(r, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][0] = LdsPair<EDGE>(L, c, row0);
    }
    JXLHIP_PSTEPK(0, KE);
    if constexpr (HX >= 2) { JXLHIP_PSTEPK(1, KE); }
    if constexpr (HX >= 3) { JXLHIP_PSTEPK(2, KE); }
    if constexpr (HX >= 4) { JXLHIP_PSTEPK(3, KE); }
  }
#undef JXLHIP_PGROUP
#undef JXLHIP_PSTEPK
}

// blockIdx.x is dispatched round-robin over the 8 XCDs: logical workgroup = (xcd, slot) -> xcd * per + slot, so
// that an XCD's L2 sees neighbouring windows of the same rows (they share two block columns of coefficients and
// plane tiles); the grid is padded to a multiple of 8
template <int GAB, int EPF, int OUTK, int FMT, typename CT>
__global__ __launch_bounds__(128, kPcWaves) void k_fused_pc(DevFrame f, FilterParams P, int RH, int strips, int nwg) {
  __shared__ StripLds lds;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(threadIdx.x >> 6);  // 0 marches, 1 produces
  const float __attribute__((address_space(3)))* dither_lds = nullptr;
  if constexpr (OUTK == JXLHIP_OUT_PACKED) {
    __shared__ float s_dither[1024];
    if (P.fmt.sample_type == JXLHIP_SAMPLE_U8) {  // uniform
      for (int i = threadIdx.x; i < 1024; i += 128) s_dither[i] = P.dither[i];
      __syncthreads();
    }
    dither_lds = (const float __attribute__((address_space(3)))*)s_dither;
  }
  const int per = (int)gridDim.x >> 3;
  const int logical = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (logical >= nwg) return;
  const int strip = logical % strips, chunk = logical / strips;
  const int W = (int)f.xsize;
  const int x_first = strip * kFusedUse;
  const int y_begin = (int)f.fy0 + chunk * RH;
  const int y_end = min(y_begin + RH, (int)f.fy1);
  if (x_first >= W || y_begin >= y_end) return;  // (both waves)
  const int x0 = x_first - kFusedHalo;
  const int bc0 = x0 >> 3;
  const FrameArgs fa = (FrameArgs)__builtin_amdgcn_kernarg_segment_ptr();
  if (wave == 1) {
    __builtin_amdgcn_s_setprio(kProducerPrio);
    if constexpr (sizeof(CT) == 2) ProducePC2<MarchGeom<GAB, EPF>::HX>(fa, &lds, bc0, y_begin, y_end, ((int)f.ysize - 1) >> 3);
    else
      ProducePC<MarchGeom<GAB, EPF>::HX, CT>(fa, &lds, bc0, y_begin, y_end, ((int)f.ysize - 1) >> 3);
    return;
  }
  Lane L;
  L.gx = x0 + 2 * lane;
  L.dither = dither_lds;
  const int m0 = MirrorF(L.gx, W), m1 = MirrorF(L.gx + 1, W);
  int base = (m0 & ~1) - x0;
  base = base < 0 ? 0 : (base > kSlabCols - 2 ? kSlabCols - 2 : base);
  L.sel0 = m0 & 1;
  L.sel1 = m1 & 1;
  L.byte_off = 0;
  L.slab = (const float __attribute__((address_space(3)))*)lds.slab[0] + base;
  const bool edge = x0 < 0 || x0 + kSlabCols > W;
  const bool lane_in = lane >= kFusedHalo / 2 && lane < 64 - kFusedHalo / 2;
  L.out0 = lane_in && L.gx < W;
  L.out1 = lane_in && L.gx + 1 < W;
  const int gxc = L.gx < 0 ? 0 : (L.gx >= W ? W - 1 : L.gx);
  L.sx4 = (uint32_t)(gxc >> 3) * 4u;
  L.out_off = (uint32_t)(L.gx < 0 ? 0 : L.gx) * (OUTK == JXLHIP_OUT_LINEAR_RGB_F32 ? 12u : 4u);
  const int ix = gxc & 7;
  L.mul = v2f{ix == 0 ? P.bsm[1] : P.sm[1], ix == 6 ? P.bsm[1] : P.sm[1]};
  L.mul2 = v2f{ix == 0 ? P.bsm[2] : P.sm[2], ix == 6 ? P.bsm[2] : P.sm[2]};
  L.fix_left = L.gx == -2;
  L.fix_right_even = L.gx == W;
  L.fix_right_odd = L.gx == W - 1;
  // a chunk of whole block rows that needs no mirror row: the march with its row bookkeeping resolved at compile time
  constexpr int HXk = MarchGeom<GAB, EPF>::HX;
  const bool interior = (y_begin & 7) == 0 && ((y_end - y_begin) & 7) == 0 && y_begin >= 8 &&
                        y_end + 8 <= (int)f.ysize;
  (void)HXk;
  if (interior) {
    if (edge) MarchPC<GAB, EPF, OUTK, FMT, true, true>(f, P, L, &lds, bc0, y_begin, y_end);
    else MarchPC<GAB, EPF, OUTK, FMT, false, true>(f, P, L, &lds, bc0, y_begin, y_end);
  } else {
    if (edge) MarchPC<GAB, EPF, OUTK, FMT, true, false>(f, P, L, &lds, bc0, y_begin, y_end);
    else MarchPC<GAB, EPF, OUTK, FMT, false, false>(f, P, L, &lds, bc0, y_begin, y_end);
  }
}

// rows per window chunk: a multiple of 8 that fills whole generations of resident workgroups (6 per CU)
int FusedRowsPC(unsigned strips, unsigned rows, unsigned per_cu = kPcPerCu) {
  const int forced = jxlhip_env::Get().fused_pc_rh.load(std::memory_order_relaxed);  // experiments / tests: rows per window chunk
  if (forced > 0) return (forced + 7) & ~7;
  const unsigned resident = DeviceCus() * per_cu;
  int best = 64;
  double best_cost = 1e30;
  for (int rh = 16; rh <= 1024; rh += 8) {
    const unsigned wgs = strips * ((rows + rh - 1) / rh);
    const unsigned gens = (wgs + resident - 1) / resident;
    const double cost = (double)gens * (rh + 6 + 16);  // 2 x HX marched rows + two more block-row fills per chunk
    if (cost < best_cost) {
      best_cost = cost;
      best = rh;
    }
  }
  return best;
}

template <int GAB, int EPF, int OUTK, int FMT = -1>
void LaunchFusedPcT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  const unsigned strips = (f.xsize + kFusedUse - 1) / kFusedUse;
  const int RH = FusedRowsPC(strips, f.fy1 - f.fy0);
  const unsigned nwg = strips * ((f.fy1 - f.fy0 + RH - 1) / RH);
  const dim3 grid((nwg + 7) & ~7u);
  if (f.coeff_type == JXLHIP_COEFF_I16)
    hipLaunchKernelGGL((k_fused_pc<GAB, EPF, OUTK, FMT, int16_t>), grid, dim3(128), 0, st, f, p, RH, (int)strips, (int)nwg);
  else
    hipLaunchKernelGGL((k_fused_pc<GAB, EPF, OUTK, FMT, int32_t>), grid, dim3(128), 0, st, f, p, RH, (int)strips, (int)nwg);
}

}  // namespace

#if JXLHIP_FUSED_PART == 2
// Frames the fused kernel takes (decided before k_prepare: it routes the DCT8 blocks).
bool FusedSupported(const DevFrame& f, int gab, int epf_iters, int output_kind) {
  (void)gab;
  if (epf_iters > 2) return false;                 // EPF0: k_epf0 + the EPF1 + EPF2 march (kernels_epf0.hip)
  // Packed outputs stay two-phase: their emit code doubles the march's instruction count, and k_fused_pc has the march
  // on half of a workgroup's waves (8K d1.0: sRGB RGBA8 0.50 ms fused against 0.42 ms two-phase,
  // profiles/r03_packed_paths.txt).  (Rounds 2-5 also shipped a single-wave fused kernel for the general packed
  // formats; removed in round 6 with the other never-default forms.)
  if (output_kind == JXLHIP_OUT_PACKED) return false;
  if (f.xsize < 16 || f.ysize < 16) return false;  // multiply mirrored columns / rows
  const uint32_t tail = f.ysize & 7u;
  if (tail >= 1 && tail <= 3) return false;        // mirror rows below the frame leave the last block row
  if ((f.fy0 & 7u) != 0) return false;
  if ((uint64_t)f.plane_tile_rows * f.tile_stride * 256u >= (1ull << 32)) return false;
  // the producing wave addresses coefficients as buffer base + 32-bit byte offset (PcIssue)
  if ((uint64_t)f.xsg * f.ysg * f.coef_stride64 * 64u * (f.coeff_type == JXLHIP_COEFF_I16 ? 2u : 4u) >= (1ull << 32)) return false;
  return true;
}
#endif  // JXLHIP_FUSED_PART == 2

#if JXLHIP_FUSED_PART == 3
// ------------------------------------------------------------------------------------------------
// k_fused_pc0: epf_iters = 3.  [Gaborish] + EPF0 marched from the producer's slab -- the DCT8 cells decoded in the
// producing wave, every other cell LDS-DMA'd from the planes, exactly as k_fused_pc's producer does it -- into the second
// plane set (row-major), from which the EPF1 + EPF2 march (k_filters_fast<0, 2>, SRC_LINEAR) produces the pixels as
// before.  What it saves over k_epf0: the DCT8 share of the frame never visits the first plane set (one write and one
// read of 12 bytes per pixel).  The march is epf0_march.h's Step0 with its rows and its inv_sigma from LDS.
static constexpr int kPc0PartLds = 0;  // running plus-sum parts of the march kept in LDS: none (4 fits three waves per
                                        // SIMD and measured slower, profiles/r04_epf3_fused.txt)
template <int GAB, bool EDGE>
__device__ __forceinline__ void MarchPC0(const DevFrame& f, const FilterParams& P, Lane& L, StripLds* w, int bc0, int y_begin,
                                         int y_end, float* const (&dst)[3], LdsF* part_lds) {
  constexpr int HX = GAB + 3;
  constexpr int PART_LDS = GAB ? kPc0PartLds : 0;
  const int H = (int)f.ysize;
  const int r_first = y_begin - 8;
  const int r_last = y_end + HX - 1;
  const int nb_last = (H - 1) >> 3;
  const int G = PcGroups<HX>(y_begin, y_end);
  State0 s;
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s.x[c][k] = v2f{0.0f, 0.0f};
      s.g[c][k] = v2f{0.0f, 0.0f};
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
  const float __attribute__((address_space(3)))* const slab0 = L.slab;
  const LdsF* sig0 = (const LdsF*)w->sigma[0] + ((int)(L.sx4 >> 2) - bc0);
  int i = 0;
  auto enter_group = [&](int g) -> float {  // after the barrier that publishes buffer g & 1
    const int b = g & 1;
    L.slab = slab0 + b * (3 * kSlabPlaneFloats);
    return sig0[b * 16];
  };
  if constexpr (PART_LDS > 0) {
    typedef v2f __attribute__((address_space(3))) * P2;
#pragma unroll
    for (int k = 0; k < PART_LDS; k++) *(P2)(part_lds + k * 128) = v2f{0.0f, 0.0f};
  }
#define JXLHIP_PSTEP0(K) \
  Step0<GAB, K, EDGE, SRC_LDS, PART_LDS>(s, r + K, f, P, L, 0, y_begin, y_end, inv_sigma_blk, dst, slab_y0, sigma_grp, part_lds)
  PcBarrierMarch();  // fill(0)
  {  // the last HX rows of the block row above
    const int r = r_first;
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;
    const float sigma_grp = enter_group(i);
    {
      const int row0 = Mirror1(r + 8 - HX, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][8 - HX] = LdsPair<EDGE>(L, c, row0);
    }
    if constexpr (HX >= 4) { JXLHIP_PSTEP0(4); }
    JXLHIP_PSTEP0(5);
    JXLHIP_PSTEP0(6);
    JXLHIP_PSTEP0(7);
    i++;
    PcBarrierMarch();  // a whole group always follows
  }
  int r = y_begin;
  for (; r_last - r >= HX; r += 8) {
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;
    const float sigma_grp = enter_group(i);
    {
      const int row0 = Mirror1(r, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][0] = LdsPair<EDGE>(L, c, row0);
    }
    JXLHIP_PSTEP0(0);
    JXLHIP_PSTEP0(1);
    JXLHIP_PSTEP0(2);
    JXLHIP_PSTEP0(3);
    JXLHIP_PSTEP0(4);
    JXLHIP_PSTEP0(5);
    JXLHIP_PSTEP0(6);
    JXLHIP_PSTEP0(7);
    i++;
    if (i < G) PcBarrierMarch();
  }
  if (r <= r_last) {  // the first HX rows of the block row below
    const int slab_y0 = GroupBlockRow(r, nb_last) * 8;
    const float sigma_grp = enter_group(i);
    {
      const int row0 = Mirror1(r, H) - slab_y0;
#pragma unroll
      for (int c = 0; c < 3; c++) s.x[c][0] = LdsPair<EDGE>(L, c, row0);
    }
    JXLHIP_PSTEP0(0);
    JXLHIP_PSTEP0(1);
    JXLHIP_PSTEP0(2);
    if constexpr (HX >= 4) { JXLHIP_PSTEP0(3); }
  }
#undef JXLHIP_PSTEP0
}

// (the EPF0 window with Gaborish in front wants 175 VGPRs, seven more than three waves per SIMD leave -- and a spilling
// build must not ship: the producing wave's asm loads, libjxl_amd/build.py.  Keeping four of the
// march's six running plus-sum parts into LDS, 2 KB per window -- what six windows per CU leave of the 160 KB beside
// their slabs -- and fits three waves; measured slower, see the macro.)
template <int GAB, typename CT>
__global__ __launch_bounds__(128, GAB != 0 ? 2 : kPcWaves) void k_fused_pc0(DevFrame f, FilterParams P, int RH, int strips, int nwg, int oy0, int oy1,
                                                                     float* d0, float* d1, float* d2) {
  __shared__ StripLds lds;
  __shared__ float part_store[(GAB != 0 && kPc0PartLds > 0) ? kPc0PartLds * 128 : 2];
  const int lane = threadIdx.x & 63;
  const int wave = (int)(threadIdx.x >> 6);  // 0 marches, 1 produces
  const int per = (int)gridDim.x >> 3;
  const int logical = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (logical >= nwg) return;
  const int strip = logical % strips, chunk = logical / strips;
  const int W = (int)f.xsize;
  const int x_first = strip * kFusedUse;
  const int y_begin = oy0 + chunk * RH;
  const int y_end = min(y_begin + RH, oy1);
  if (x_first >= W || y_begin >= y_end) return;  // (both waves)
  const int x0 = x_first - kFusedHalo;
  const int bc0 = x0 >> 3;
  const FrameArgs fa = (FrameArgs)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int HX = GAB + 3;
  if (wave == 1) {
    __builtin_amdgcn_s_setprio(kProducerPrio);
    if constexpr (sizeof(CT) == 2) ProducePC2<HX>(fa, &lds, bc0, y_begin, y_end, ((int)f.ysize - 1) >> 3);
    else ProducePC<HX, CT>(fa, &lds, bc0, y_begin, y_end, ((int)f.ysize - 1) >> 3);
    return;
  }
  Lane L;
  L.gx = x0 + 2 * lane;
  L.dither = nullptr;
  const int m0 = MirrorF(L.gx, W), m1 = MirrorF(L.gx + 1, W);
  int base = (m0 & ~1) - x0;
  base = base < 0 ? 0 : (base > kSlabCols - 2 ? kSlabCols - 2 : base);
  L.sel0 = m0 & 1;
  L.sel1 = m1 & 1;
  L.byte_off = 0;
  L.slab = (const float __attribute__((address_space(3)))*)lds.slab[0] + base;
  const bool edge = x0 < 0 || x0 + kSlabCols > W;
  const bool lane_in = lane >= kFusedHalo / 2 && lane < 64 - kFusedHalo / 2;
  L.out0 = lane_in && L.gx < W;
  L.out1 = lane_in && L.gx + 1 < W;
  const int gxc = L.gx < 0 ? 0 : (L.gx >= W ? W - 1 : L.gx);
  L.sx4 = (uint32_t)(gxc >> 3) * 4u;
  L.out_off = (uint32_t)gxc * 4u;
  const int ix = gxc & 7;
  L.mul = v2f{ix == 0 ? P.bsm[0] : P.sm[0], ix == 6 ? P.bsm[0] : P.sm[0]};
  L.mul2 = L.mul;
  L.fix_left = L.fix_right_even = L.fix_right_odd = false;
  float* const dst[3] = {d0, d1, d2};
  LdsF* const part_lds = (LdsF*)part_store + 2 * lane;
  if (edge) MarchPC0<GAB, true>(f, P, L, &lds, bc0, y_begin, y_end, dst, part_lds);
  else MarchPC0<GAB, false>(f, P, L, &lds, bc0, y_begin, y_end, dst, part_lds);
}
#endif  // JXLHIP_FUSED_PART == 3

#if JXLHIP_FUSED_PART == 2
bool LaunchFused(const DevFrame& f, const FilterParams& p, int gab, int epf_iters, int output_kind, hipStream_t st) {
  if (!FusedSupported(f, gab, epf_iters, output_kind)) return false;
#define JXLHIP_FUSED_PCX(G, E)                                    \
  if (gab == G && epf_iters == E) {                               \
    if (output_kind == 0) LaunchFusedPcT<G, E, 0>(f, p, st);      \
    else LaunchFusedPcT<G, E, 1>(f, p, st);                       \
    return true;                                                  \
  }
  JXLHIP_FUSED_PCX(1, 1)
  JXLHIP_FUSED_PCX(0, 0)
  JXLHIP_FUSED_PCX(0, 1)
  JXLHIP_FUSED_PCX(1, 0)
  JXLHIP_FUSED_PCX(0, 2)
  JXLHIP_FUSED_PCX(1, 2)
#undef JXLHIP_FUSED_PCX
  return false;
}
#elif JXLHIP_FUSED_PART == 3
// [Gaborish] + EPF0 of a frame whose phase 1 ran in fused mode (DevFrame::fused = 1: the DCT8 cells are not in the
// planes), for the rows the following EPF1 + EPF2 march of rows [f.fy0, f.fy1) reads, into dst (row-major second plane
// set, as LaunchEpf0).  false: geometry / configuration not covered (the caller then must not have skipped the DCT8 cells).
bool FusedEpf0Supported(const DevFrame& f, int gab) {
  (void)gab;
  if (f.xsize < 16 || f.ysize < 16) return false;
  const uint32_t tail = f.ysize & 7u;
  if (tail >= 1 && tail <= 3) return false;  // (as FusedSupported: mirror rows below the frame leave the last block row)
  if ((f.fy0 & 7u) != 0 || f.fy0 != 0 || f.fy1 != f.ysize) return false;  // whole frames
  if ((uint64_t)f.plane_tile_rows * f.tile_stride * 256u >= (1ull << 32)) return false;
  return true;
}
bool LaunchFusedEpf0(const DevFrame& f, const FilterParams& p, int gab, float* const dst[3], hipStream_t st) {
  if (!FusedEpf0Supported(f, gab)) return false;
  const int oy0 = 0, oy1 = (int)f.ysize;
  const unsigned strips = (f.xsize + kFusedUse - 1) / kFusedUse;
  // (with Gaborish the march wants 175 VGPRs: two waves per SIMD = four windows per CU)
  const int RH = FusedRowsPC(strips, oy1 - oy0, gab != 0 ? 4 : kPcPerCu);
  const unsigned nwg = strips * ((oy1 - oy0 + RH - 1) / RH);
  const dim3 grid((nwg + 7) & ~7u);
#define JXLHIP_PC0(G, CT) \
  hipLaunchKernelGGL((k_fused_pc0<G, CT>), grid, dim3(128), 0, st, f, p, RH, (int)strips, (int)nwg, oy0, oy1, dst[0], dst[1], dst[2])
  if (f.coeff_type == JXLHIP_COEFF_I16) {
    if (gab) JXLHIP_PC0(1, int16_t);
    else JXLHIP_PC0(0, int16_t);
  } else {
    if (gab) JXLHIP_PC0(1, int32_t);
    else JXLHIP_PC0(0, int32_t);
  }
#undef JXLHIP_PC0
  return true;
}
#endif

}  // namespace jxlhip
