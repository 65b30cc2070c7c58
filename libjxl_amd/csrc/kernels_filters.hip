// kernels_filters.hip -- phase 2 of the VarDCT back-end on gfx950: the render
// pipeline stages Gaborish -> EPF0 -> EPF1 -> EPF2 -> XYB->linear RGB (+ float
// output packing) fused into ONE kernel.  Each workgroup owns a TW x TH output
// tile; the tile plus the accumulated stage borders is staged in LDS and every
// stage runs LDS -> LDS, the last one straight to the output buffer.
//
// Replaces (behaviour, not code): lib/jxl/render_pipeline/stage_gaborish.cc:33-99,
// stage_epf.cc:47-494, stage_xyb.cc:42-98 + lib/jxl/dec_xyb-inl.h:38-86, and the
// executor lib/jxl/render_pipeline/simple_render_pipeline.cc:79-297 whose border
// rule is reproduced exactly: every stage sees its input mirrored at the TRUE
// image edge (lib/jxl/image_ops.h:184-196).
#include "dev_common.h"
#include "emit.h"
#include "kernels.h"

namespace jxlhip {

__device__ __forceinline__ int Mirror(int x, int n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}

enum StageId : int { kGab = 0, kEpf0 = 1, kEpf1 = 2, kEpf2 = 3 };

// Stage list of PassesDecoderState::PreparePipeline (dec_cache.cc:151-170).
template <int GAB, int EPF>
struct Chain {
  static constexpr int kNum = GAB + EPF;
  static constexpr int Stage(int i) {
    if (GAB) {
      if (i == 0) return kGab;
      i--;
    }
    if (EPF == 3) return kEpf0 + i;
    return kEpf1 + i;  // EPF == 1: epf1 ; EPF == 2: epf1, epf2
  }
  static constexpr int Border(int id) { return id == kGab ? 1 : id == kEpf0 ? 3 : id == kEpf1 ? 2 : 1; }
  // halo still needed in front of stage i (i == kNum -> 0)
  static constexpr int Halo(int i) {
    int h = 0;
    for (int j = i; j < kNum; j++) h += Border(Stage(j));
    return h;
  }
};

__device__ __forceinline__ float EpfWeight(float sad, float inv_sigma) {
  const float v = __builtin_fmaf(sad, inv_sigma, 1.0f);
  return v < 0.0f ? 0.0f : v;
}

// One pixel of stage ID.  p0/p1/p2: pointers to the centre sample in the three
// LDS planes (row stride iw).  Returns the three outputs in o[].
template <int ID>
__device__ __forceinline__ void StagePixel(const float* p0, const float* p1, const float* p2,
                                           int iw, int gx, int gy, const DevFrame& f,
                                           const FilterParams& P, float* o) {
  const float* pl[3] = {p0, p1, p2};
  if constexpr (ID == kGab) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float* q = pl[c];
      const float t = q[-iw], tl = q[-iw - 1], tr = q[-iw + 1];
      const float m = q[0], l = q[-1], r = q[1];
      const float b = q[iw], bl = q[iw - 1], br = q[iw + 1];
      const float sum1 = (l + r) + (t + b);
      const float sum2 = (tl + tr) + (bl + br);
      o[c] = __builtin_fmaf(sum2, P.gab_w[c][2],
                            __builtin_fmaf(sum1, P.gab_w[c][1], m * P.gab_w[c][0]));
    }
  } else {
    constexpr int which = ID - kEpf0;
    const float kMinSigma = -3.90524291751269967465540850526868f;  // epf.h:22
    const float is = f.inv_sigma[(size_t)(gy >> 3) * f.xsb + (gx >> 3)];
    float X = p0[0], Y = p1[0], B = p2[0];
    if (is < kMinSigma) {
      o[0] = X;
      o[1] = Y;
      o[2] = B;
      return;
    }
    const int ix = gx & 7, iy = gy & 7;
    const bool border = (iy == 0) | (iy == 7) | (ix == 0) | (ix == 7);
    const float inv_sigma = is * (border ? P.bsm[which] : P.sm[which]);
    float wsum = 1.0f;
    if constexpr (ID == kEpf0) {
      constexpr int kOff[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1},
                                   {0, 1},  {0, 2},   {1, -1}, {1, 0},  {1, 1},  {2, 0}};
      constexpr int kPlus[5][2] = {{0, 0}, {-1, 0}, {0, -1}, {1, 0}, {0, 1}};
      float sads[12];
#pragma unroll
      for (int i = 0; i < 12; i++) sads[i] = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* q = pl[c];
        const float scale = P.ch_scale[c];
        float r[5];
#pragma unroll
        for (int k = 0; k < 5; k++) r[k] = q[kPlus[k][0] * iw + kPlus[k][1]];
#pragma unroll
        for (int i = 0; i < 12; i++) {
          float sad = 0.0f;
#pragma unroll
          for (int k = 0; k < 5; k++) {
            const float c11 = q[(kOff[i][0] + kPlus[k][0]) * iw + kOff[i][1] + kPlus[k][1]];
            sad = sad + __builtin_fabsf(r[k] - c11);
          }
          sads[i] = __builtin_fmaf(sad, scale, sads[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const float w = EpfWeight(sads[i], inv_sigma);
        const int d = kOff[i][0] * iw + kOff[i][1];
        wsum = wsum + w;
        X = __builtin_fmaf(w, p0[d], X);
        Y = __builtin_fmaf(w, p1[d], Y);
        B = __builtin_fmaf(w, p2[d], B);
      }
    } else if constexpr (ID == kEpf1) {
      float sad0 = 0, sad1 = 0, sad2 = 0, sad3 = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* q = pl[c];
        // pXY: X = column (2 = centre), Y = row (2 = centre)
        const float p20 = q[-2 * iw], p21 = q[-iw];
        float sad0c = __builtin_fabsf(p20 - p21);
        const float p11 = q[-iw - 1];
        float sad1c = __builtin_fabsf(p11 - p21);
        const float p31 = q[-iw + 1];
        float sad2c = __builtin_fabsf(p31 - p21);
        const float p02 = q[-2], p12 = q[-1];
        sad1c = sad1c + __builtin_fabsf(p02 - p12);
        sad0c = sad0c + __builtin_fabsf(p11 - p12);
        const float p22 = q[0];
        float t = __builtin_fabsf(p12 - p22);
        sad1c = sad1c + t;
        sad2c = sad2c + t;
        t = __builtin_fabsf(p22 - p21);
        float sad3c = t;
        sad0c = sad0c + t;
        const float p32 = q[1];
        sad0c = sad0c + __builtin_fabsf(p31 - p32);
        t = __builtin_fabsf(p22 - p32);
        sad1c = sad1c + t;
        sad2c = sad2c + t;
        const float p42 = q[2];
        sad2c = sad2c + __builtin_fabsf(p42 - p32);
        const float p13 = q[iw - 1];
        sad3c = sad3c + __builtin_fabsf(p13 - p12);
        const float p23 = q[iw];
        t = __builtin_fabsf(p22 - p23);
        sad0c = sad0c + t;
        sad3c = sad3c + t;
        sad1c = sad1c + __builtin_fabsf(p13 - p23);
        const float p33 = q[iw + 1];
        sad2c = sad2c + __builtin_fabsf(p33 - p23);
        sad3c = sad3c + __builtin_fabsf(p33 - p32);
        const float p24 = q[2 * iw];
        sad3c = sad3c + __builtin_fabsf(p24 - p23);
        const float scale = P.ch_scale[c];
        sad0 = __builtin_fmaf(sad0c, scale, sad0);
        sad1 = __builtin_fmaf(sad1c, scale, sad1);
        sad2 = __builtin_fmaf(sad2c, scale, sad2);
        sad3 = __builtin_fmaf(sad3c, scale, sad3);
      }
      const float sads[4] = {sad0, sad1, sad2, sad3};
      const int offs[4] = {-iw, -1, 1, iw};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float w = EpfWeight(sads[i], inv_sigma);
        wsum = wsum + w;
        X = __builtin_fmaf(w, p0[offs[i]], X);
        Y = __builtin_fmaf(w, p1[offs[i]], Y);
        B = __builtin_fmaf(w, p2[offs[i]], B);
      }
    } else {
      const int offs[4] = {-iw, -1, 1, iw};
      const float rx = X, ry = Y, rb = B;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float cx = p0[offs[i]], cy = p1[offs[i]], cb = p2[offs[i]];
        float sad = __builtin_fabsf(cx - rx) * P.ch_scale[0];
        sad = __builtin_fmaf(__builtin_fabsf(cy - ry), P.ch_scale[1], sad);
        sad = __builtin_fmaf(__builtin_fabsf(cb - rb), P.ch_scale[2], sad);
        const float w = EpfWeight(sad, inv_sigma);
        wsum = wsum + w;
        X = __builtin_fmaf(w, cx, X);
        Y = __builtin_fmaf(w, cy, Y);
        B = __builtin_fmaf(w, cb, B);
      }
    }
    const float inv_w = __builtin_amdgcn_rcpf(wsum);
    o[0] = X * inv_w;
    o[1] = Y * inv_w;
    o[2] = B * inv_w;
  }
}

// XybToRgb (dec_xyb-inl.h:38-86) + output packing; gy is the image row
template <int OUTK>
__device__ __forceinline__ void EmitPixel(const float* v, int gx, int gy, const DevFrame& f,
                                          const FilterParams& P) {
  const int gy_rel = gy - (int)f.y0;
  if constexpr (OUTK == JXLHIP_OUT_LINEAR_RGB_F32) {
    float* dst = (float*)((char*)P.out + (size_t)gy_rel * P.out_stride) + 3 * (size_t)gx;
    float rgb[3];
    XybToRgb(v[0], v[1], v[2], P, rgb);
    dst[0] = rgb[0];
    dst[1] = rgb[1];
    dst[2] = rgb[2];
  } else if constexpr (OUTK == JXLHIP_OUT_PACKED) {
    float rgb[3];
    XybToRgb(v[0], v[1], v[2], P, rgb);
    StorePackedPixel<FmtSel<-1>>(P, P.dither, (char*)P.out + (size_t)gy_rel * P.out_stride, gx, gy, rgb);
  } else {
    float* dst = (float*)P.out + (size_t)gy_rel * P.out_stride + gx;
    dst[0] = v[0];
    dst[P.out_plane_stride] = v[1];
    dst[2 * P.out_plane_stride] = v[2];
  }
}

template <typename CH, int I, int OUTK, int TW, int TH, int NT>
__device__ __forceinline__ void RunStages(const DevFrame& f, const FilterParams& P, float* in,
                                          float* out, int tx0, int ty0, bool edge_tile) {
  if constexpr (I < CH::kNum) {
    constexpr int ID = CH::Stage(I);
    constexpr int IH = CH::Halo(I), OH = CH::Halo(I + 1);
    constexpr int IW = TW + 2 * IH, IHt = TH + 2 * IH;
    constexpr int OW = TW + 2 * OH, OHt = TH + 2 * OH;
    constexpr int OFF = IH - OH;
    constexpr bool kLast = (I == CH::kNum - 1);
    const int W = (int)f.xsize, H = (int)f.ysize;
    for (int i = threadIdx.x; i < OW * OHt; i += NT) {
      const int ry = i / OW, rx = i % OW;
      const int gx = tx0 - OH + rx, gy = ty0 - OH + ry;
      if (gx < 0 || gx >= W || gy < 0 || gy >= H) continue;
      if (kLast && gy >= (int)f.fy1) continue;
      const int ci = (ry + OFF) * IW + rx + OFF;
      float o[3];
      StagePixel<ID>(in + ci, in + IW * IHt + ci, in + 2 * IW * IHt + ci, IW, gx, gy, f, P, o);
      if constexpr (kLast) {
        EmitPixel<OUTK>(o, gx, gy, f, P);
      } else {
        out[i] = o[0];
        out[OW * OHt + i] = o[1];
        out[2 * OW * OHt + i] = o[2];
      }
    }
    if constexpr (!kLast) {
      __syncthreads();
      if (edge_tile) {
        // the next stage reads this one's output mirrored at the image edge
        for (int i = threadIdx.x; i < OW * OHt; i += NT) {
          const int ry = i / OW, rx = i % OW;
          const int gx = tx0 - OH + rx, gy = ty0 - OH + ry;
          if (gx >= 0 && gx < W && gy >= 0 && gy < H) continue;
          const int sx = Mirror(gx, W) - (tx0 - OH), sy = Mirror(gy, H) - (ty0 - OH);
          if (sx < 0 || sx >= OW || sy < 0 || sy >= OHt) continue;
          const int s = sy * OW + sx;
          out[i] = out[s];
          out[OW * OHt + i] = out[OW * OHt + s];
          out[2 * OW * OHt + i] = out[2 * OW * OHt + s];
        }
        __syncthreads();
      }
      RunStages<CH, I + 1, OUTK, TW, TH, NT>(f, P, out, in, tx0, ty0, edge_tile);
    }
  }
}

template <int GAB, int EPF, int OUTK, int TW, int TH, int NT>
__global__ __launch_bounds__(NT) void k_filters(DevFrame f, FilterParams P) {
  using CH = Chain<GAB, EPF>;
  constexpr int HT = CH::Halo(0);
  constexpr int AW = TW + 2 * HT, AH = TH + 2 * HT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem;
  float* Bf = smem + 3 * AW * AH;
  const int tx0 = blockIdx.x * TW;
  const int ty0 = (int)f.fy0 + blockIdx.y * TH;
  const int W = (int)f.xsize, H = (int)f.ysize;
  for (int i = threadIdx.x; i < AW * AH; i += NT) {
    const int ry = i / AW, rx = i % AW;
    const int mx = Mirror(tx0 - HT + rx, W);
    const int my = Mirror(ty0 - HT + ry, H);
    const int prow = my - f.plane_y0;
    float v0 = 0, v1 = 0, v2 = 0;
    if (prow >= 0 && prow < (int)f.plane_tile_rows * 8) {
      const size_t o = PlaneOffset(f, my, mx);
      v0 = f.xyb[0][o];
      v1 = f.xyb[1][o];
      v2 = f.xyb[2][o];
    }
    A[i] = v0;
    A[AW * AH + i] = v1;
    A[2 * AW * AH + i] = v2;
  }
  __syncthreads();
  const bool edge_tile = tx0 - HT < 0 || tx0 + TW + HT > W || ty0 - HT < 0 || ty0 + TH + HT > H;
  RunStages<CH, 0, OUTK, TW, TH, NT>(f, P, A, Bf, tx0, ty0, edge_tile);
}

// No loop filter at all: pointwise XYB -> output.
template <int OUTK>
__global__ __launch_bounds__(256) void k_xyb_only(DevFrame f, FilterParams P) {
  const int gx = blockIdx.x * 256 + threadIdx.x;
  const int gy = (int)f.fy0 + blockIdx.y;
  if (gx >= (int)f.xsize || gy >= (int)f.fy1) return;
  const size_t o = PlaneOffset(f, gy, gx);
  const float v[3] = {f.xyb[0][o], f.xyb[1][o], f.xyb[2][o]};
  EmitPixel<OUTK>(v, gx, gy, f, P);
}

template <int GAB, int EPF, int OUTK>
static void LaunchFiltersT(const DevFrame& f, const FilterParams& p, hipStream_t st) {
  constexpr int TW = 64, TH = 32, NT = 256;
  using CH = Chain<GAB, EPF>;
  constexpr int HT = CH::Halo(0), H1 = CH::Halo(1);
  constexpr size_t lds =
      sizeof(float) * 3 * ((TW + 2 * HT) * (TH + 2 * HT) + (TW + 2 * H1) * (TH + 2 * H1));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_filters<GAB, EPF, OUTK, TW, TH, NT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((f.xsize + TW - 1) / TW, (f.fy1 - f.fy0 + TH - 1) / TH);
  hipLaunchKernelGGL((k_filters<GAB, EPF, OUTK, TW, TH, NT>), grid, dim3(NT), lds, st, f, p);
}

int LaunchFilters(const DevFrame& f, const FilterParams& p, int gab, int epf_iters,
                  int output_kind, hipStream_t st) {
  if (gab < 0 || gab > 1 || epf_iters < 0 || epf_iters > 3 || output_kind < 0 || output_kind > 2)
    return -1;
  if (f.fy1 <= f.fy0) return 0;
  if (gab == 0 && epf_iters == 0) {
    const dim3 grid((f.xsize + 255) / 256, f.fy1 - f.fy0);
    if (output_kind == 0)
      hipLaunchKernelGGL(k_xyb_only<0>, grid, dim3(256), 0, st, f, p);
    else if (output_kind == 1)
      hipLaunchKernelGGL(k_xyb_only<1>, grid, dim3(256), 0, st, f, p);
    else
      hipLaunchKernelGGL(k_xyb_only<2>, grid, dim3(256), 0, st, f, p);
    return 0;
  }
#define JXLHIP_CASE(G, E)                                    \
  if (gab == G && epf_iters == E) {                          \
    if (output_kind == 0) LaunchFiltersT<G, E, 0>(f, p, st); \
    else if (output_kind == 1) LaunchFiltersT<G, E, 1>(f, p, st); \
    else LaunchFiltersT<G, E, 2>(f, p, st);                  \
    return 0;                                                \
  }
  JXLHIP_CASE(1, 0)
  JXLHIP_CASE(0, 1)
  JXLHIP_CASE(1, 1)
  JXLHIP_CASE(0, 2)
  JXLHIP_CASE(1, 2)
  JXLHIP_CASE(0, 3)
  JXLHIP_CASE(1, 3)
#undef JXLHIP_CASE
  return -1;
}

}  // namespace jxlhip
