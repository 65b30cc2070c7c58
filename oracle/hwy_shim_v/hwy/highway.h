// TEST / MEASUREMENT INFRASTRUCTURE ONLY (part of oracle/): a from-scratch MULTI-LANE stand-in for Google Highway's
// <hwy/highway.h>, used for ONE purpose -- bench.py's cpu_baseline: the translation units of libjxl's decode hot path
// (dec_group.cc with the inverse transforms, the Gaborish / EPF / XYB / write stages) compiled IN PLACE from
// /root/reference against 256-bit vectors (8 float lanes, what Highway's AVX2 target gives them), so that the CPU
// figure beside the GPU one is libjxl's SIMD code path and not its code on one lane (oracle/hwy_shim, the bit-exact
// CHECKER, stays single-lane).  oracle/build_ref.py variant "v8" compiles only those units with this header in front
// of the single-lane one; every other unit is the single-lane object.
//
// Vectors are GCC vector-extension values (`T __attribute__((vector_size))`): element-wise operations are single
// expressions the compiler maps to AVX2 instructions with -mavx2 -mfma, lane crossings are __builtin_shuffle with
// constant index vectors.  Semantics follow Highway's documentation of each operation for 128-bit-block targets
// (InterleaveLower / Upper, Shuffle*, Broadcast, TableLookupBytes, LoadDup128 work per 128-bit block; Concat*,
// LowerHalf / UpperHalf, Combine, Reverse on the whole vector).  Like the single-lane shim: MulAdd is a single-rounding
// FMA, ApproximateReciprocal(Sqrt) are exact.  Held to the single-lane checker within the reference's own executor
// tolerance (2e-4, render_pipeline_test.cc:321-327) by tests/test_reference_parity.py.
#ifndef ORACLE_HWY_SHIM_V_HIGHWAY_H_
#define ORACLE_HWY_SHIM_V_HIGHWAY_H_

#include <immintrin.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>

#include "hwy/base.h"
#include "hwy/cache_control.h"

// (base.h is shared with the single-lane shim: the vector-size dependent macros are this header's)
#undef HWY_ALIGN
#undef HWY_MAX_BYTES
#undef HWY_LANES
#define HWY_ALIGN alignas(32)
#define HWY_MAX_BYTES 32
#define HWY_LANES(T) (32 / sizeof(T))
#ifndef HWY_IF_NOT_FLOAT
#define HWY_IF_NOT_FLOAT(T) typename std::enable_if<!std::is_floating_point<T>::value>::type* = nullptr
#endif

// ---- target identification ----------------------------------------------
#define HWY_AVX3_SPR (1LL << 4)
#define HWY_AVX3_ZEN4 (1LL << 6)
#define HWY_AVX3_DL (1LL << 7)
#define HWY_AVX3 (1LL << 8)
#define HWY_AVX2 (1LL << 9)
#define HWY_SSE4 (1LL << 11)
#define HWY_SSSE3 (1LL << 12)
#define HWY_SSE2 (1LL << 14)
#define HWY_HIGHEST_TARGET_BIT_X86 14
#define HWY_SVE2_128 (1LL << 15)
#define HWY_SVE_256 (1LL << 16)
#define HWY_SVE2 (1LL << 17)
#define HWY_SVE (1LL << 18)
#define HWY_NEON_BF16 (1LL << 19)
#define HWY_NEON (1LL << 20)
#define HWY_NEON_WITHOUT_AES (1LL << 21)
#define HWY_RVV (1LL << 34)
#define HWY_PPC10 (1LL << 45)
#define HWY_PPC9 (1LL << 46)
#define HWY_PPC8 (1LL << 47)
#define HWY_Z15 (1LL << 48)
#define HWY_Z14 (1LL << 49)
#define HWY_WASM_EMU256 (1LL << 55)
#define HWY_WASM (1LL << 56)
#define HWY_EMU128 (1LL << 61)
#define HWY_SCALAR (1LL << 62)

#define HWY_TARGET HWY_AVX2
#define HWY_STATIC_TARGET HWY_AVX2
#define HWY_TARGETS HWY_AVX2
#define HWY_NAMESPACE N_AVX2
#define HWY_ONCE 1
#define HWY_IDE 0

#define HWY_CAP_GE256 1
#define HWY_CAP_GE512 0
#define HWY_CAP_INTEGER64 1
#define HWY_CAP_FLOAT16 0
#define HWY_CAP_FLOAT64 1
#define HWY_HAVE_SCALABLE 0
#define HWY_HAVE_INTEGER64 1
#define HWY_HAVE_FLOAT16 0
#define HWY_HAVE_FLOAT64 1
#define HWY_MEM_OPS_MIGHT_FAULT 1
#define HWY_NATIVE_FMA 1

#define HWY_BEFORE_NAMESPACE() static_assert(true, "hwy shim")
#define HWY_AFTER_NAMESPACE() static_assert(true, "hwy shim")
#define HWY_EXPORT(FUNC) static_assert(true, "hwy shim")
#define HWY_EXPORT_T(TABLE, FUNC) static_assert(true, "hwy shim")
#define HWY_STATIC_DISPATCH(FUNC) N_AVX2::FUNC
#define HWY_DYNAMIC_DISPATCH(FUNC) N_AVX2::FUNC
#define HWY_DYNAMIC_POINTER(FUNC) (&N_AVX2::FUNC)
#define HWY_DYNAMIC_DISPATCH_T(TABLE) N_AVX2::TABLE
#define HWY_EXPORT_AND_DYNAMIC_DISPATCH_T(FUNC) N_AVX2::FUNC

#define HWY_SHIM_VECTOR_BYTES 32
#define HWY_FULL(T) hwy::N_AVX2::Simd<T, HWY_SHIM_VECTOR_BYTES / sizeof(T), 0>
#define HWY_CAPPED(T, N) \
  hwy::N_AVX2::Simd<T, ((N) < HWY_SHIM_VECTOR_BYTES / sizeof(T) ? (N) : HWY_SHIM_VECTOR_BYTES / sizeof(T)), 0>
#define HWY_FULL1(T) HWY_FULL(T)
#define HWY_FULL2(T, LMUL) HWY_FULL(T)

namespace hwy {

// Software binary16 <-> binary32 (round to nearest even), as in the single-lane shim.
static inline float F32FromF16Bits(uint16_t h) {
  const uint32_t sign = static_cast<uint32_t>(h >> 15) << 31;
  const uint32_t exp = (h >> 10) & 0x1F;
  const uint32_t man = h & 0x3FF;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {
      float f = static_cast<float>(man) * 5.9604644775390625e-8f;
      uint32_t fb;
      memcpy(&fb, &f, 4);
      bits = sign | fb;
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &bits, 4);
  return out;
}
static inline uint16_t F16BitsFromF32(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t absx = x & 0x7FFFFFFFu;
  if (absx >= 0x7F800000u) return static_cast<uint16_t>(sign | 0x7C00u | (absx > 0x7F800000u ? 0x200u | ((absx >> 13) & 0x3FF) : 0));
  if (absx >= 0x477FF000u) return static_cast<uint16_t>(sign | 0x7C00u);
  if (absx < 0x38800000u) {
    float a;
    memcpy(&a, &absx, 4);
    const uint32_t m = static_cast<uint32_t>(lrintf(a * 16777216.0f));
    return static_cast<uint16_t>(sign | m);
  }
  uint32_t mant = absx & 0x7FFFFFu;
  uint32_t exp = (absx >> 23) - 112;
  uint32_t half = (exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;
  return static_cast<uint16_t>(sign | half);
}
static inline float F32FromF16(float16_t h) { return F32FromF16Bits(h.bits); }
static inline float16_t F16FromF32(float f) {
  float16_t r;
  r.bits = F16BitsFromF32(f);
  return r;
}
static inline float F32FromBF16(bfloat16_t b) {
  uint32_t bits = static_cast<uint32_t>(b.bits) << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

namespace N_AVX2 {

// ---- descriptors ----------------------------------------------------------
template <typename Lane, size_t N, int kPow2>
struct Simd {
  constexpr Simd() = default;
  using T = Lane;
  static constexpr size_t kPrivateLanes = N;
  static constexpr int kPrivatePow2 = 0;
  template <typename NewT>
  using Rebind = Simd<NewT, N, 0>;
  template <typename NewT>
  using Repartition = Simd<NewT, (N * sizeof(Lane) / sizeof(NewT)) ? (N * sizeof(Lane) / sizeof(NewT)) : 1, 0>;
  using Half = Simd<Lane, (N / 2) ? (N / 2) : 1, 0>;
  using Twice = Simd<Lane, N * 2, 0>;
  constexpr size_t MaxLanes() const { return N; }
  constexpr size_t MaxBytes() const { return N * sizeof(Lane); }
  constexpr size_t MaxBlocks() const { return (N * sizeof(Lane) + 15) / 16; }
  constexpr int Pow2() const { return 0; }
};
template <typename T, int kPow2 = 0>
using ScalableTag = Simd<T, HWY_SHIM_VECTOR_BYTES / sizeof(T), 0>;
template <typename T, size_t kLimit, int kPow2 = 0>
using CappedTag = Simd<T, (kLimit < HWY_SHIM_VECTOR_BYTES / sizeof(T) ? kLimit : HWY_SHIM_VECTOR_BYTES / sizeof(T)), 0>;
template <typename T, size_t kNumLanes>
using FixedTag = Simd<T, kNumLanes, 0>;
template <typename T>
using Sisd = Simd<T, 1, 0>;
template <typename T>
using Full16 = Simd<T, 2 / sizeof(T) ? 2 / sizeof(T) : 1, 0>;
template <typename T>
using Full32 = Simd<T, 4 / sizeof(T) ? 4 / sizeof(T) : 1, 0>;
template <typename T>
using Full64 = Simd<T, 8 / sizeof(T), 0>;
template <typename T>
using Full128 = Simd<T, 16 / sizeof(T), 0>;

template <class D>
using TFromD = typename D::T;
template <class T, class D>
using Rebind = typename D::template Rebind<T>;
template <class T, class D>
using Repartition = typename D::template Repartition<T>;
template <class D>
using RebindToSigned = Rebind<MakeSigned<TFromD<D>>, D>;
template <class D>
using RebindToUnsigned = Rebind<MakeUnsigned<TFromD<D>>, D>;
template <class D>
using RebindToFloat = Rebind<MakeFloat<TFromD<D>>, D>;
template <class D>
using RepartitionToWide = Repartition<MakeWide<TFromD<D>>, D>;
template <class D>
using RepartitionToNarrow = Repartition<MakeNarrow<TFromD<D>>, D>;
template <class D>
using Half = typename D::Half;
template <class D>
using Twice = typename D::Twice;

template <class D>
constexpr size_t Lanes(D) {
  return D::kPrivateLanes;
}
template <class D>
constexpr size_t MaxLanes(D) {
  return D::kPrivateLanes;
}
#define HWY_MAX_LANES_D(D) (D::kPrivateLanes)

// ---- vector and mask --------------------------------------------------------
template <typename T, size_t N, bool kVector = (std::is_arithmetic<T>::value && N > 1)>
struct RawOf {
  typedef T type __attribute__((vector_size(N * sizeof(T))));
};
template <typename T, size_t N>
struct RawOf<T, N, false> {  // one lane, or a lane type the compiler has no vectors of (float16_t): an array
  struct type {
    T v[N];
    T& operator[](size_t i) { return v[i]; }
    const T& operator[](size_t i) const { return v[i]; }
  };
};
template <typename T, size_t N>
struct VecN {
  using PrivateT = T;
  static constexpr size_t kPrivateN = N;
  using Raw = typename RawOf<T, N>::type;
  Raw raw;
};
template <typename T, size_t N>
struct MaskN {  // lane i: all ones / all zero, as an unsigned integer of the lane's size
  using U = MakeUnsigned<T>;
  typename RawOf<U, N>::type raw;
};

// true: the lanes live in one GCC vector value (whole-vector expressions are single AVX2 instructions)
template <typename T, size_t N>
constexpr bool kIsVec = std::is_arithmetic<T>::value && N > 1;
template <typename T, size_t N>
using URawOf = typename RawOf<MakeUnsigned<T>, N>::type;
template <typename T, size_t N>
using SRawOf = typename RawOf<MakeSigned<T>, N>::type;

template <class D>
using VFromD = VecN<TFromD<D>, D::kPrivateLanes>;
template <class D>
using Vec = VFromD<D>;
template <class D>
using MFromD = MaskN<TFromD<D>, D::kPrivateLanes>;
template <class D>
using Mask = MFromD<D>;
template <class V>
using TFromV = typename V::PrivateT;
template <class V>
using DFromV = Simd<typename V::PrivateT, V::kPrivateN, 0>;
template <typename T>
using Vec256 = VecN<T, 32 / sizeof(T)>;
template <typename T>
using Vec128 = VecN<T, 16 / sizeof(T)>;
template <typename T>
using Vec64 = VecN<T, 8 / sizeof(T)>;

// lane-wise helpers: F(i) -> lane i
#define HWY_SHIM_FOR(i, N) for (size_t i = 0; i < (N); ++i)

template <typename T>
HWY_INLINE MakeUnsigned<T> ToBits(T t) {
  MakeUnsigned<T> u;
  memcpy(&u, &t, sizeof(T));
  return u;
}
template <typename T>
HWY_INLINE T FromBits(MakeUnsigned<T> u) {
  T t;
  memcpy(&t, &u, sizeof(T));
  return t;
}

// ---- init ---------------------------------------------------------------------------
template <class D, typename T2>
HWY_API VFromD<D> Set(D, T2 t) {
  VFromD<D> v;
  if constexpr (kIsVec<TFromD<D>, D::kPrivateLanes>) {
    v.raw = typename VFromD<D>::Raw{} + static_cast<TFromD<D>>(t);
  } else {
    HWY_SHIM_FOR(i, D::kPrivateLanes) v.raw[i] = static_cast<TFromD<D>>(t);
  }
  return v;
}
template <class D>
HWY_API VFromD<D> Zero(D d) {
  return Set(d, TFromD<D>(0));
}
template <class D>
HWY_API VFromD<D> Undefined(D d) {
  return Zero(d);
}
template <class D, typename T2>
HWY_API VFromD<D> Iota(D, T2 first) {
  VFromD<D> v;
  HWY_SHIM_FOR(i, D::kPrivateLanes) v.raw[i] = static_cast<TFromD<D>>(first + static_cast<T2>(i));
  return v;
}
template <class D>
HWY_API VFromD<D> SignBit(D d) {
  using T = TFromD<D>;
  using U = MakeUnsigned<T>;
  return Set(d, FromBits<T>(static_cast<U>(U(1) << (sizeof(T) * 8 - 1))));
}
template <typename T, size_t N>
HWY_API T GetLane(VecN<T, N> v) {
  return v.raw[0];
}
template <typename T, size_t N>
HWY_API T ExtractLane(VecN<T, N> v, size_t i) {
  return v.raw[i];
}
template <typename T, size_t N>
HWY_API VecN<T, N> InsertLane(VecN<T, N> v, size_t i, T t) {
  v.raw[i] = t;
  return v;
}
template <class D, typename FromT, size_t FromN>
HWY_API VFromD<D> BitCast(D, VecN<FromT, FromN> v) {
  static_assert(sizeof(VFromD<D>) == sizeof(v) || true, "");
  VFromD<D> out;
  memset(&out, 0, sizeof(out));
  memcpy(&out, &v, sizeof(out) < sizeof(v) ? sizeof(out) : sizeof(v));
  return out;
}
template <class D, typename FromT, size_t FromN>
HWY_API VFromD<D> ResizeBitCast(D d, VecN<FromT, FromN> v) {
  return BitCast(d, v);
}

// ---- memory -------------------------------------------------------------------------
template <class D>
HWY_API VFromD<D> LoadU(D, const TFromD<D>* HWY_RESTRICT p) {
  VFromD<D> v;
  memcpy(&v.raw, p, D::kPrivateLanes * sizeof(TFromD<D>));
  return v;
}
template <class D>
HWY_API VFromD<D> Load(D d, const TFromD<D>* HWY_RESTRICT p) {
  return LoadU(d, p);
}
template <class D>
HWY_API VFromD<D> LoadDup128(D, const TFromD<D>* HWY_RESTRICT p) {
  constexpr size_t kBlock = 16 / sizeof(TFromD<D>);
  VFromD<D> v;
  HWY_SHIM_FOR(i, D::kPrivateLanes) v.raw[i] = p[i % kBlock];
  return v;
}
template <class D>
HWY_API VFromD<D> LoadN(D d, const TFromD<D>* HWY_RESTRICT p, size_t n) {
  VFromD<D> v = Zero(d);
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (i < n) v.raw[i] = p[i];
  return v;
}
template <class D>
HWY_API VFromD<D> MaskedLoad(MFromD<D> m, D d, const TFromD<D>* HWY_RESTRICT p) {
  VFromD<D> v = Zero(d);
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (m.raw[i]) v.raw[i] = p[i];
  return v;
}
template <class D>
HWY_API void StoreU(VFromD<D> v, D, TFromD<D>* HWY_RESTRICT p) {
  memcpy(p, &v.raw, D::kPrivateLanes * sizeof(TFromD<D>));
}
template <class D>
HWY_API void Store(VFromD<D> v, D d, TFromD<D>* HWY_RESTRICT p) {
  StoreU(v, d, p);
}
template <class D>
HWY_API void Stream(VFromD<D> v, D d, TFromD<D>* HWY_RESTRICT p) {
  StoreU(v, d, p);
}
template <class D>
HWY_API void StoreN(VFromD<D> v, D, TFromD<D>* HWY_RESTRICT p, size_t n) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (i < n) p[i] = v.raw[i];
}
template <class D>
HWY_API void BlendedStore(VFromD<D> v, MFromD<D> m, D, TFromD<D>* HWY_RESTRICT p) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (m.raw[i]) p[i] = v.raw[i];
}
template <class D, typename TI, size_t NI>
HWY_API VFromD<D> GatherIndex(D, const TFromD<D>* HWY_RESTRICT base, VecN<TI, NI> index) {
  VFromD<D> v;
  HWY_SHIM_FOR(i, D::kPrivateLanes) v.raw[i] = base[index.raw[i]];
  return v;
}
template <class D>
HWY_API void StoreInterleaved2(VFromD<D> v0, VFromD<D> v1, D, TFromD<D>* HWY_RESTRICT p) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    p[2 * i] = v0.raw[i];
    p[2 * i + 1] = v1.raw[i];
  }
}
template <class D>
HWY_API void StoreInterleaved3(VFromD<D> v0, VFromD<D> v1, VFromD<D> v2, D, TFromD<D>* HWY_RESTRICT p) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    p[3 * i] = v0.raw[i];
    p[3 * i + 1] = v1.raw[i];
    p[3 * i + 2] = v2.raw[i];
  }
}
template <class D>
HWY_API void StoreInterleaved4(VFromD<D> v0, VFromD<D> v1, VFromD<D> v2, VFromD<D> v3, D, TFromD<D>* HWY_RESTRICT p) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    p[4 * i] = v0.raw[i];
    p[4 * i + 1] = v1.raw[i];
    p[4 * i + 2] = v2.raw[i];
    p[4 * i + 3] = v3.raw[i];
  }
}
template <class D>
HWY_API void LoadInterleaved2(D, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    v0.raw[i] = p[2 * i];
    v1.raw[i] = p[2 * i + 1];
  }
}
template <class D>
HWY_API void LoadInterleaved3(D, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1, VFromD<D>& v2) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    v0.raw[i] = p[3 * i];
    v1.raw[i] = p[3 * i + 1];
    v2.raw[i] = p[3 * i + 2];
  }
}
template <class D>
HWY_API void LoadInterleaved4(D, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1, VFromD<D>& v2, VFromD<D>& v3) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) {
    v0.raw[i] = p[4 * i];
    v1.raw[i] = p[4 * i + 1];
    v2.raw[i] = p[4 * i + 2];
    v3.raw[i] = p[4 * i + 3];
  }
}

// ---- arithmetic -----------------------------------------------------------------------
#define HWY_SHIM_BINOP(NAME, EXPR, VEXPR)                                        \
  template <typename T, size_t N>                                                \
  HWY_API VecN<T, N> NAME(VecN<T, N> a, VecN<T, N> b) {                          \
    VecN<T, N> r;                                                                \
    if constexpr (kIsVec<T, N>) {                                                \
      using Raw = typename VecN<T, N>::Raw;                                      \
      using URaw = URawOf<T, N>;                                                 \
      (void)sizeof(URaw);                                                        \
      r.raw = (Raw)(VEXPR);                                                      \
    } else {                                                                     \
      HWY_SHIM_FOR(i, N) {                                                       \
        const T x = a.raw[i], y = b.raw[i];                                      \
        r.raw[i] = static_cast<T>(EXPR);                                         \
      }                                                                          \
    }                                                                            \
    return r;                                                                    \
  }
// (integers wrap: whole-vector arithmetic runs in the unsigned domain)
template <typename T>
HWY_INLINE T WrapAdd(T x, T y) {
  if constexpr (std::is_floating_point<T>::value) return x + y;
  else return static_cast<T>(static_cast<MakeUnsigned<T>>(x) + static_cast<MakeUnsigned<T>>(y));
}
template <typename T>
HWY_INLINE T WrapSub(T x, T y) {
  if constexpr (std::is_floating_point<T>::value) return x - y;
  else return static_cast<T>(static_cast<MakeUnsigned<T>>(x) - static_cast<MakeUnsigned<T>>(y));
}
template <typename T>
HWY_INLINE T WrapMul(T x, T y) {
  if constexpr (std::is_floating_point<T>::value) return x * y;
  else return static_cast<T>(static_cast<MakeUnsigned<T>>(x) * static_cast<MakeUnsigned<T>>(y));
}
template <typename T, size_t N, class Raw>
HWY_INLINE auto ArithDomain(Raw v) {  // floats as they are, integers as unsigned vectors
  if constexpr (std::is_floating_point<T>::value) return v;
  else return (URawOf<T, N>)v;
}
HWY_SHIM_BINOP(Add, WrapAdd(x, y), (ArithDomain<T, N>(a.raw) + ArithDomain<T, N>(b.raw)))
HWY_SHIM_BINOP(Sub, WrapSub(x, y), (ArithDomain<T, N>(a.raw) - ArithDomain<T, N>(b.raw)))
HWY_SHIM_BINOP(Mul, WrapMul(x, y), (ArithDomain<T, N>(a.raw) * ArithDomain<T, N>(b.raw)))
HWY_SHIM_BINOP(Div, x / y, (a.raw / b.raw))
HWY_SHIM_BINOP(Min, (y < x ? y : x), (b.raw < a.raw ? b.raw : a.raw))
HWY_SHIM_BINOP(Max, (x < y ? y : x), (a.raw < b.raw ? b.raw : a.raw))
HWY_SHIM_BINOP(AbsDiff, (x < y ? WrapSub(y, x) : WrapSub(x, y)),
               (a.raw < b.raw ? (Raw)(ArithDomain<T, N>(b.raw) - ArithDomain<T, N>(a.raw))
                              : (Raw)(ArithDomain<T, N>(a.raw) - ArithDomain<T, N>(b.raw))))
template <typename T, size_t N>
HWY_API VecN<T, N> operator+(VecN<T, N> a, VecN<T, N> b) { return Add(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator-(VecN<T, N> a, VecN<T, N> b) { return Sub(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator*(VecN<T, N> a, VecN<T, N> b) { return Mul(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator/(VecN<T, N> a, VecN<T, N> b) { return Div(a, b); }

#define HWY_SHIM_UNOP(NAME, EXPR, VEXPR)                \
  template <typename T, size_t N>                       \
  HWY_API VecN<T, N> NAME(VecN<T, N> a) {               \
    VecN<T, N> r;                                       \
    if constexpr (kIsVec<T, N> && (VEXPR##_OK)) {       \
      using Raw = typename VecN<T, N>::Raw;             \
      r.raw = (Raw)(VEXPR(a));                          \
    } else {                                            \
      HWY_SHIM_FOR(i, N) {                              \
        const T x = a.raw[i];                           \
        r.raw[i] = static_cast<T>(EXPR);                \
      }                                                 \
    }                                                   \
    return r;                                           \
  }
template <typename T, size_t N>
HWY_INLINE auto VNeg(VecN<T, N> a) {
  return typename VecN<T, N>::Raw{} - a.raw;  // (floats: 0 - x differs from -x only in the sign of zero; Highway's Neg flips the sign bit)
}
template <typename T, size_t N>
HWY_INLINE auto VAbsF(VecN<T, N> a) {  // clear the sign bit
  using U = MakeUnsigned<T>;
  return (typename VecN<T, N>::Raw)((URawOf<T, N>)a.raw & static_cast<U>(~(U(1) << (sizeof(T) * 8 - 1))));
}
template <typename T, size_t N>
HWY_INLINE auto VRecip(VecN<T, N> a) {
  return (typename VecN<T, N>::Raw{} + T(1)) / a.raw;
}
#define VAbsF_OK std::is_floating_point<T>::value
#define VRecip_OK std::is_floating_point<T>::value
#define VNone_OK false
#define VNone(a) a.raw
template <typename T, size_t N>
HWY_API VecN<T, N> Neg(VecN<T, N> a) {
  VecN<T, N> r;
  if constexpr (kIsVec<T, N> && std::is_floating_point<T>::value) {
    using U = MakeUnsigned<T>;
    r.raw = (typename VecN<T, N>::Raw)((URawOf<T, N>)a.raw ^ static_cast<U>(U(1) << (sizeof(T) * 8 - 1)));
  } else if constexpr (kIsVec<T, N>) {
    r.raw = (typename VecN<T, N>::Raw)(URawOf<T, N>{} - (URawOf<T, N>)a.raw);
  } else {
    HWY_SHIM_FOR(i, N) r.raw[i] = WrapSub(T(0), static_cast<T>(a.raw[i]));
  }
  return r;
}
HWY_SHIM_UNOP(Abs, (x < T(0) ? WrapSub(T(0), x) : x), VAbsF)
HWY_SHIM_UNOP(ApproximateReciprocal, T(1) / x, VRecip)
HWY_SHIM_UNOP(ApproximateReciprocalSqrt, T(1) / std::sqrt(x), VNone)
template <typename T, size_t N>
HWY_API VecN<T, N> Sqrt(VecN<T, N> a) {
  VecN<T, N> r;
  if constexpr (std::is_same<T, float>::value && N == 8) {
    r.raw = (typename VecN<T, N>::Raw)_mm256_sqrt_ps((__m256)a.raw);
  } else if constexpr (std::is_same<T, float>::value && N == 4) {
    r.raw = (typename VecN<T, N>::Raw)_mm_sqrt_ps((__m128)a.raw);
  } else {
    HWY_SHIM_FOR(i, N) r.raw[i] = std::sqrt(static_cast<T>(a.raw[i]));
  }
  return r;
}
#define HWY_SHIM_ROUND(NAME, MODE, SCALAR)                                                      \
  template <typename T, size_t N>                                                               \
  HWY_API VecN<T, N> NAME(VecN<T, N> a) {                                                       \
    VecN<T, N> r;                                                                               \
    if constexpr (std::is_same<T, float>::value && N == 8) {                                    \
      r.raw = (typename VecN<T, N>::Raw)_mm256_round_ps((__m256)a.raw, (MODE) | _MM_FROUND_NO_EXC); \
    } else if constexpr (std::is_same<T, float>::value && N == 4) {                             \
      r.raw = (typename VecN<T, N>::Raw)_mm_round_ps((__m128)a.raw, (MODE) | _MM_FROUND_NO_EXC);  \
    } else {                                                                                    \
      HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(SCALAR(static_cast<T>(a.raw[i])));          \
    }                                                                                           \
    return r;                                                                                   \
  }
HWY_SHIM_ROUND(Round, _MM_FROUND_TO_NEAREST_INT, std::nearbyint)
HWY_SHIM_ROUND(Trunc, _MM_FROUND_TO_ZERO, std::trunc)
HWY_SHIM_ROUND(Floor, _MM_FROUND_TO_NEG_INF, std::floor)
HWY_SHIM_ROUND(Ceil, _MM_FROUND_TO_POS_INF, std::ceil)

#define HWY_SHIM_FMA(NAME, SA, SC, I256, I128)                                                         \
  template <size_t N>                                                                                  \
  HWY_API VecN<float, N> NAME(VecN<float, N> a, VecN<float, N> b, VecN<float, N> c) {                  \
    VecN<float, N> r;                                                                                  \
    if constexpr (N == 8) {                                                                            \
      r.raw = (typename VecN<float, N>::Raw)I256((__m256)a.raw, (__m256)b.raw, (__m256)c.raw);          \
    } else if constexpr (N == 4) {                                                                     \
      r.raw = (typename VecN<float, N>::Raw)I128((__m128)a.raw, (__m128)b.raw, (__m128)c.raw);          \
    } else {                                                                                           \
      HWY_SHIM_FOR(i, N) r.raw[i] = __builtin_fmaf(SA a.raw[i], b.raw[i], SC c.raw[i]);                \
    }                                                                                                  \
    return r;                                                                                          \
  }
HWY_SHIM_FMA(MulAdd, +, +, _mm256_fmadd_ps, _mm_fmadd_ps)
HWY_SHIM_FMA(NegMulAdd, -, +, _mm256_fnmadd_ps, _mm_fnmadd_ps)
HWY_SHIM_FMA(MulSub, +, -, _mm256_fmsub_ps, _mm_fmsub_ps)
HWY_SHIM_FMA(NegMulSub, -, -, _mm256_fnmsub_ps, _mm_fnmsub_ps)
template <size_t N>
HWY_API VecN<double, N> MulAdd(VecN<double, N> a, VecN<double, N> b, VecN<double, N> c) {
  VecN<double, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = __builtin_fma(a.raw[i], b.raw[i], c.raw[i]);
  return r;
}
template <size_t N>
HWY_API VecN<double, N> NegMulAdd(VecN<double, N> a, VecN<double, N> b, VecN<double, N> c) {
  VecN<double, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = __builtin_fma(-a.raw[i], b.raw[i], c.raw[i]);
  return r;
}
template <typename T, size_t N, HWY_IF_NOT_FLOAT(T)>
HWY_API VecN<T, N> MulAdd(VecN<T, N> a, VecN<T, N> b, VecN<T, N> c) {
  return Add(Mul(a, b), c);
}

template <typename T, size_t N>
HWY_API VecN<T, N> SaturatedAdd(VecN<T, N> a, VecN<T, N> b) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) {
    const int64_t s = static_cast<int64_t>(a.raw[i]) + static_cast<int64_t>(b.raw[i]);
    const int64_t lo = std::numeric_limits<T>::min(), hi = std::numeric_limits<T>::max();
    r.raw[i] = static_cast<T>(s < lo ? lo : (s > hi ? hi : s));
  }
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> SaturatedSub(VecN<T, N> a, VecN<T, N> b) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) {
    const int64_t s = static_cast<int64_t>(a.raw[i]) - static_cast<int64_t>(b.raw[i]);
    const int64_t lo = std::numeric_limits<T>::min(), hi = std::numeric_limits<T>::max();
    r.raw[i] = static_cast<T>(s < lo ? lo : (s > hi ? hi : s));
  }
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> AverageRound(VecN<T, N> a, VecN<T, N> b) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>((static_cast<uint64_t>(a.raw[i]) + static_cast<uint64_t>(b.raw[i]) + 1) >> 1);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> MulHigh(VecN<T, N> a, VecN<T, N> b) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>((static_cast<int64_t>(a.raw[i]) * static_cast<int64_t>(b.raw[i])) >> (sizeof(T) * 8));
  return r;
}

// ---- shifts ------------------------------------------------------------------------------
template <int kBits, typename T, size_t N>
HWY_API VecN<T, N> ShiftLeft(VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(static_cast<MakeUnsigned<T>>(v.raw[i]) << kBits);
  return r;
}
template <int kBits, typename T, size_t N>
HWY_API VecN<T, N> ShiftRight(VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(v.raw[i] >> kBits);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> ShiftLeftSame(VecN<T, N> v, int bits) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(static_cast<MakeUnsigned<T>>(v.raw[i]) << bits);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> ShiftRightSame(VecN<T, N> v, int bits) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(v.raw[i] >> bits);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shl(VecN<T, N> v, VecN<T, N> bits) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(static_cast<MakeUnsigned<T>>(v.raw[i]) << bits.raw[i]);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shr(VecN<T, N> v, VecN<T, N> bits) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<T>(v.raw[i] >> bits.raw[i]);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> operator<<(VecN<T, N> v, VecN<T, N> bits) { return Shl(v, bits); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator>>(VecN<T, N> v, VecN<T, N> bits) { return Shr(v, bits); }

// ---- logical ---------------------------------------------------------------------------
#define HWY_SHIM_BITOP(NAME, EXPR, VEXPR)                                   \
  template <typename T, size_t N>                                           \
  HWY_API VecN<T, N> NAME(VecN<T, N> a, VecN<T, N> b) {                     \
    VecN<T, N> r;                                                           \
    if constexpr (kIsVec<T, N>) {                                           \
      const URawOf<T, N> x = (URawOf<T, N>)a.raw, y = (URawOf<T, N>)b.raw;  \
      r.raw = (typename VecN<T, N>::Raw)(VEXPR);                            \
    } else {                                                                \
      HWY_SHIM_FOR(i, N) {                                                  \
        const MakeUnsigned<T> x = ToBits(static_cast<T>(a.raw[i])), y = ToBits(static_cast<T>(b.raw[i])); \
        r.raw[i] = FromBits<T>(static_cast<MakeUnsigned<T>>(EXPR));         \
      }                                                                     \
    }                                                                       \
    return r;                                                               \
  }
HWY_SHIM_BITOP(And, x & y, x & y)
HWY_SHIM_BITOP(Or, x | y, x | y)
HWY_SHIM_BITOP(Xor, x ^ y, x ^ y)
HWY_SHIM_BITOP(AndNot, ~x & y, ~x & y)
template <typename T, size_t N>
HWY_API VecN<T, N> Not(VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = FromBits<T>(static_cast<MakeUnsigned<T>>(~ToBits(static_cast<T>(v.raw[i]))));
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Or3(VecN<T, N> a, VecN<T, N> b, VecN<T, N> c) { return Or(a, Or(b, c)); }
template <typename T, size_t N>
HWY_API VecN<T, N> Xor3(VecN<T, N> a, VecN<T, N> b, VecN<T, N> c) { return Xor(a, Xor(b, c)); }
template <typename T, size_t N>
HWY_API VecN<T, N> OrAnd(VecN<T, N> o, VecN<T, N> a1, VecN<T, N> a2) { return Or(o, And(a1, a2)); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator&(VecN<T, N> a, VecN<T, N> b) { return And(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator|(VecN<T, N> a, VecN<T, N> b) { return Or(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> operator^(VecN<T, N> a, VecN<T, N> b) { return Xor(a, b); }
template <typename T, size_t N>
HWY_API VecN<T, N> CopySign(VecN<T, N> magn, VecN<T, N> sign) {
  const auto msb = SignBit(DFromV<VecN<T, N>>());
  return Or(AndNot(msb, magn), And(msb, sign));
}
template <typename T, size_t N>
HWY_API VecN<T, N> CopySignToAbs(VecN<T, N> abs, VecN<T, N> sign) {
  return Or(abs, And(SignBit(DFromV<VecN<T, N>>()), sign));
}

// ---- masks -----------------------------------------------------------------------------
template <typename T, size_t N>
HWY_API MaskN<T, N> MaskFromVec(VecN<T, N> v) {
  MaskN<T, N> m;
  using U = MakeUnsigned<T>;
  if constexpr (kIsVec<T, N>) {
    m.raw = (URawOf<T, N>)(((SRawOf<T, N>)v.raw) >> (sizeof(T) * 8 - 1));  // arithmetic shift: the MSB everywhere
  } else {
    HWY_SHIM_FOR(i, N) m.raw[i] = (ToBits(static_cast<T>(v.raw[i])) >> (sizeof(T) * 8 - 1)) ? static_cast<U>(~U(0)) : U(0);
  }
  return m;
}
template <typename T, size_t N>
HWY_API VecN<T, N> VecFromMask(MaskN<T, N> m) {
  VecN<T, N> v;
  if constexpr (kIsVec<T, N>) {
    v.raw = (typename VecN<T, N>::Raw)m.raw;
  } else {
    HWY_SHIM_FOR(i, N) v.raw[i] = FromBits<T>(m.raw[i]);
  }
  return v;
}
template <class D>
HWY_API VFromD<D> VecFromMask(D, MFromD<D> m) {
  return VecFromMask(m);
}
template <class D, typename TFrom, size_t N>
HWY_API MFromD<D> RebindMask(D, MaskN<TFrom, N> m) {
  MFromD<D> r;
  using U = MakeUnsigned<TFromD<D>>;
  HWY_SHIM_FOR(i, N) r.raw[i] = m.raw[i] ? static_cast<U>(~U(0)) : U(0);
  return r;
}
template <class D>
HWY_API MFromD<D> FirstN(D, size_t n) {
  MFromD<D> r;
  using U = MakeUnsigned<TFromD<D>>;
  HWY_SHIM_FOR(i, D::kPrivateLanes) r.raw[i] = i < n ? static_cast<U>(~U(0)) : U(0);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> IfThenElse(MaskN<T, N> m, VecN<T, N> yes, VecN<T, N> no) {
  VecN<T, N> r;
  if constexpr (kIsVec<T, N>) {
    const URawOf<T, N> y = (URawOf<T, N>)yes.raw, n = (URawOf<T, N>)no.raw;
    r.raw = (typename VecN<T, N>::Raw)((y & m.raw) | (n & ~m.raw));
  } else {
    HWY_SHIM_FOR(i, N) r.raw[i] = m.raw[i] ? yes.raw[i] : no.raw[i];
  }
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> IfThenElseZero(MaskN<T, N> m, VecN<T, N> yes) {
  return IfThenElse(m, yes, Zero(DFromV<VecN<T, N>>()));
}
template <typename T, size_t N>
HWY_API VecN<T, N> IfThenZeroElse(MaskN<T, N> m, VecN<T, N> no) {
  return IfThenElse(m, Zero(DFromV<VecN<T, N>>()), no);
}
template <typename T, size_t N>
HWY_API VecN<T, N> IfVecThenElse(VecN<T, N> mask, VecN<T, N> yes, VecN<T, N> no) {
  return Or(And(mask, yes), AndNot(mask, no));
}
template <typename T, size_t N>
HWY_API VecN<T, N> IfNegativeThenElse(VecN<T, N> v, VecN<T, N> yes, VecN<T, N> no) {
  return IfThenElse(MaskFromVec(v), yes, no);
}
template <typename T, size_t N>
HWY_API VecN<T, N> ZeroIfNegative(VecN<T, N> v) {
  VecN<T, N> r;
  if constexpr (kIsVec<T, N>) {
    const typename VecN<T, N>::Raw zero{};
    r.raw = v.raw < zero ? zero : v.raw;
  } else {
    HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i] < T(0) ? T(0) : v.raw[i];
  }
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Clamp(VecN<T, N> v, VecN<T, N> lo, VecN<T, N> hi) {
  return Min(Max(lo, v), hi);
}
#define HWY_SHIM_MASKOP(NAME, EXPR)                                      \
  template <typename T, size_t N>                                        \
  HWY_API MaskN<T, N> NAME(MaskN<T, N> a, MaskN<T, N> b) {               \
    MaskN<T, N> r;                                                       \
    if constexpr (kIsVec<T, N>) {                                        \
      const URawOf<T, N> x = a.raw, y = b.raw;                           \
      r.raw = (EXPR);                                                    \
    } else {                                                             \
      HWY_SHIM_FOR(i, N) {                                               \
        const MakeUnsigned<T> x = a.raw[i], y = b.raw[i];                \
        r.raw[i] = static_cast<MakeUnsigned<T>>(EXPR);                   \
      }                                                                  \
    }                                                                    \
    return r;                                                            \
  }
HWY_SHIM_MASKOP(And, x & y)
HWY_SHIM_MASKOP(Or, x | y)
HWY_SHIM_MASKOP(Xor, x ^ y)
HWY_SHIM_MASKOP(AndNot, ~x & y)
template <typename T, size_t N>
HWY_API MaskN<T, N> Not(MaskN<T, N> m) {
  MaskN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<MakeUnsigned<T>>(~m.raw[i]);
  return r;
}
template <class D>
HWY_API bool AllTrue(D, MFromD<D> m) {
  bool all = true;
  HWY_SHIM_FOR(i, D::kPrivateLanes) all &= m.raw[i] != 0;
  return all;
}
template <class D>
HWY_API bool AllFalse(D, MFromD<D> m) {
  bool any = false;
  HWY_SHIM_FOR(i, D::kPrivateLanes) any |= m.raw[i] != 0;
  return !any;
}
template <class D>
HWY_API size_t CountTrue(D, MFromD<D> m) {
  size_t n = 0;
  HWY_SHIM_FOR(i, D::kPrivateLanes) n += m.raw[i] != 0;
  return n;
}
template <class D>
HWY_API intptr_t FindFirstTrue(D, MFromD<D> m) {
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (m.raw[i]) return static_cast<intptr_t>(i);
  return -1;
}

// ---- comparisons -------------------------------------------------------------------------
#define HWY_SHIM_CMP(NAME, OP)                                                            \
  template <typename T, size_t N>                                                         \
  HWY_API MaskN<T, N> NAME(VecN<T, N> a, VecN<T, N> b) {                                  \
    MaskN<T, N> m;                                                                        \
    using U = MakeUnsigned<T>;                                                            \
    if constexpr (kIsVec<T, N>) {                                                         \
      m.raw = (URawOf<T, N>)(a.raw OP b.raw);                                             \
    } else {                                                                              \
      HWY_SHIM_FOR(i, N) m.raw[i] = (a.raw[i] OP b.raw[i]) ? static_cast<U>(~U(0)) : U(0);  \
    }                                                                                     \
    return m;                                                                             \
  }
HWY_SHIM_CMP(Eq, ==)
HWY_SHIM_CMP(Ne, !=)
HWY_SHIM_CMP(Lt, <)
HWY_SHIM_CMP(Le, <=)
HWY_SHIM_CMP(Gt, >)
HWY_SHIM_CMP(Ge, >=)
template <typename T, size_t N>
HWY_API MaskN<T, N> operator==(VecN<T, N> a, VecN<T, N> b) { return Eq(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> operator!=(VecN<T, N> a, VecN<T, N> b) { return Ne(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> operator<(VecN<T, N> a, VecN<T, N> b) { return Lt(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> operator<=(VecN<T, N> a, VecN<T, N> b) { return Le(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> operator>(VecN<T, N> a, VecN<T, N> b) { return Gt(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> operator>=(VecN<T, N> a, VecN<T, N> b) { return Ge(a, b); }
template <typename T, size_t N>
HWY_API MaskN<T, N> TestBit(VecN<T, N> v, VecN<T, N> bit) {
  return Ne(And(v, bit), Zero(DFromV<VecN<T, N>>()));
}
template <typename T, size_t N>
HWY_API MaskN<T, N> IsNaN(VecN<T, N> v) {
  return Ne(v, v);
}

// ---- conversions --------------------------------------------------------------------------
template <typename ToT, typename FromT>
HWY_INLINE ToT ConvertLane(FromT f) {
  if constexpr (std::is_same<FromT, float16_t>::value) {
    return static_cast<ToT>(F32FromF16(f));
  } else if constexpr (std::is_same<ToT, float16_t>::value) {
    return F16FromF32(static_cast<float>(f));
  } else if constexpr (std::is_floating_point<FromT>::value && std::is_integral<ToT>::value) {
    // Highway: float -> int conversions saturate (x86: out of range -> INT_MIN; Highway fixes positive overflow)
    if (std::isnan(f)) return ToT(0);
    const double lo = static_cast<double>(std::numeric_limits<ToT>::min());
    const double hi = static_cast<double>(std::numeric_limits<ToT>::max());
    const double t = std::trunc(static_cast<double>(f));
    if (t <= lo) return std::numeric_limits<ToT>::min();
    if (t >= hi) return std::numeric_limits<ToT>::max();
    return static_cast<ToT>(t);
  } else {
    return static_cast<ToT>(f);
  }
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> ConvertTo(D, VecN<FromT, N> v) {
  VFromD<D> r;
  if constexpr (kIsVec<FromT, N> && kIsVec<TFromD<D>, N> && std::is_integral<FromT>::value) {  // int -> float: exact semantics
    r.raw = __builtin_convertvector(v.raw, typename VFromD<D>::Raw);
  } else {
    HWY_SHIM_FOR(i, N) r.raw[i] = ConvertLane<TFromD<D>>(static_cast<FromT>(v.raw[i]));
  }
  return r;
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> PromoteTo(D, VecN<FromT, N> v) {
  VFromD<D> r;
  if constexpr (kIsVec<FromT, N> && kIsVec<TFromD<D>, N> && N == D::kPrivateLanes &&
                !(std::is_floating_point<FromT>::value && std::is_integral<TFromD<D>>::value)) {
    r.raw = __builtin_convertvector(v.raw, typename VFromD<D>::Raw);  // widening: every value is representable
  } else {
    HWY_SHIM_FOR(i, D::kPrivateLanes) r.raw[i] = ConvertLane<TFromD<D>>(static_cast<FromT>(v.raw[i]));
  }
  return r;
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> DemoteTo(D, VecN<FromT, N> v) {
  using ToT = TFromD<D>;
  VFromD<D> r;
  HWY_SHIM_FOR(i, N) {
    const FromT f = static_cast<FromT>(v.raw[i]);
    if constexpr (std::is_integral<FromT>::value && std::is_integral<ToT>::value) {  // saturating
      const int64_t lo = static_cast<int64_t>(std::numeric_limits<ToT>::min());
      const int64_t hi = static_cast<int64_t>(std::numeric_limits<ToT>::max());
      const int64_t x = static_cast<int64_t>(f);
      r.raw[i] = static_cast<ToT>(x < lo ? lo : (x > hi ? hi : x));
    } else {
      r.raw[i] = ConvertLane<ToT>(f);
    }
  }
  return r;
}
template <size_t N>
HWY_API VecN<uint8_t, N> U8FromU32(VecN<uint32_t, N> v) {
  VecN<uint8_t, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<uint8_t>(v.raw[i] & 0xFF);
  return r;
}
template <size_t N>
HWY_API VecN<int32_t, N> NearestInt(VecN<float, N> v) {
  VecN<int32_t, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = ConvertLane<int32_t>(std::nearbyint(static_cast<float>(v.raw[i])));
  return r;
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> TruncateTo(D, VecN<FromT, N> v) {
  VFromD<D> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<TFromD<D>>(v.raw[i]);
  return r;
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> PromoteLowerTo(D d, VecN<FromT, N> v) {
  return PromoteTo(d, v);  // (reads lanes 0 .. Lanes(d) - 1)
}
template <class D, typename FromT, size_t N>
HWY_API VFromD<D> PromoteUpperTo(D, VecN<FromT, N> v) {
  VFromD<D> r;
  HWY_SHIM_FOR(i, D::kPrivateLanes) r.raw[i] = ConvertLane<TFromD<D>>(static_cast<FromT>(v.raw[i + N / 2]));
  return r;
}

// ---- halves, blocks, lane crossings -----------------------------------------------------------
template <typename T, size_t N>
HWY_API VecN<T, (N / 2 ? N / 2 : 1)> LowerHalf(VecN<T, N> v) {
  VecN<T, (N / 2 ? N / 2 : 1)> r;
  HWY_SHIM_FOR(i, (N / 2 ? N / 2 : 1)) r.raw[i] = v.raw[i];
  return r;
}
template <class D, typename T, size_t N>
HWY_API VFromD<D> LowerHalf(D, VecN<T, N> v) {
  return LowerHalf(v);
}
template <class D, typename T, size_t N>
HWY_API VFromD<D> UpperHalf(D, VecN<T, N> v) {
  VFromD<D> r;
  HWY_SHIM_FOR(i, D::kPrivateLanes) r.raw[i] = v.raw[i + N / 2];
  return r;
}
template <class D, typename T, size_t NH>
HWY_API VFromD<D> Combine(D, VecN<T, NH> hi, VecN<T, NH> lo) {
  VFromD<D> r;
  HWY_SHIM_FOR(i, NH) {
    r.raw[i] = lo.raw[i];
    r.raw[i + NH] = hi.raw[i];
  }
  return r;
}
template <class D, typename T, size_t NH>
HWY_API VFromD<D> ZeroExtendVector(D d, VecN<T, NH> lo) {
  VFromD<D> r = Zero(d);
  HWY_SHIM_FOR(i, NH) r.raw[i] = lo.raw[i];
  return r;
}
// lanes per 128-bit block (at most the vector)
template <typename T, size_t N>
constexpr size_t BlockLanes() {
  return (16 / sizeof(T)) < N ? (16 / sizeof(T)) : N;
}
// r[i] = (a ++ b)[F(i)]: one __builtin_shuffle with an index vector the optimiser folds to a constant
template <class F, typename T, size_t N>
HWY_INLINE VecN<T, N> Shuffle2(VecN<T, N> a, VecN<T, N> b, F f) {
  VecN<T, N> r;
  if constexpr (kIsVec<T, N>) {
    SRawOf<T, N> idx;
    HWY_SHIM_FOR(i, N) idx[i] = static_cast<MakeSigned<T>>(f(i));
    r.raw = __builtin_shuffle(a.raw, b.raw, idx);
  } else {
    HWY_SHIM_FOR(i, N) {
      const size_t k = f(i);
      r.raw[i] = k < N ? a.raw[k] : b.raw[k - N];
    }
  }
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> InterleaveLower(VecN<T, N> a, VecN<T, N> b) {
  constexpr size_t B = BlockLanes<T, N>();
  return Shuffle2(a, b, [](size_t i) { return (i / B * B) + (i % B) / 2 + ((i & 1) ? N : 0); });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> InterleaveLower(D, VecN<T, N> a, VecN<T, N> b) {
  return InterleaveLower(a, b);
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> InterleaveUpper(D, VecN<T, N> a, VecN<T, N> b) {
  constexpr size_t B = BlockLanes<T, N>();
  return Shuffle2(a, b, [](size_t i) { return (i / B * B) + B / 2 + (i % B) / 2 + ((i & 1) ? N : 0); });
}
template <class DW, typename T, size_t N>
HWY_API VFromD<DW> ZipLower(DW dw, VecN<T, N> a, VecN<T, N> b) {
  return BitCast(dw, InterleaveLower(a, b));
}
template <class DW, typename T, size_t N>
HWY_API VFromD<DW> ZipUpper(DW dw, VecN<T, N> a, VecN<T, N> b) {
  return BitCast(dw, InterleaveUpper(DFromV<VecN<T, N>>(), a, b));
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatLowerLower(D, VecN<T, N> hi, VecN<T, N> lo) {
  return Shuffle2(lo, hi, [](size_t i) { return i < N / 2 ? (i) : (i - N / 2 + N); });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatUpperUpper(D, VecN<T, N> hi, VecN<T, N> lo) {
  return Shuffle2(lo, hi, [](size_t i) { return i < N / 2 ? (i + N / 2) : (i + N); });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatLowerUpper(D, VecN<T, N> hi, VecN<T, N> lo) {
  return Shuffle2(lo, hi, [](size_t i) { return i < N / 2 ? (i + N / 2) : (i - N / 2 + N); });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatUpperLower(D, VecN<T, N> hi, VecN<T, N> lo) {
  return Shuffle2(lo, hi, [](size_t i) { return i < N / 2 ? (i) : (i + N); });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatEven(D, VecN<T, N> hi, VecN<T, N> lo) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N / 2) {
    r.raw[i] = lo.raw[2 * i];
    r.raw[i + N / 2] = hi.raw[2 * i];
  }
  return r;
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> ConcatOdd(D, VecN<T, N> hi, VecN<T, N> lo) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N / 2) {
    r.raw[i] = lo.raw[2 * i + 1];
    r.raw[i + N / 2] = hi.raw[2 * i + 1];
  }
  return r;
}
template <int kLane, typename T, size_t N>
HWY_API VecN<T, N> Broadcast(VecN<T, N> v) {
  constexpr size_t B = BlockLanes<T, N>();
  return Shuffle2(v, v, [](size_t i) { return i / B * B + kLane; });
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> Reverse(D, VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[N - 1 - i];
  return r;
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> Reverse2(D, VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 1];
  return r;
}
template <class D, typename T, size_t N>
HWY_API VecN<T, N> Reverse4(D, VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 3];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle2301(VecN<T, N> v) {  // swap adjacent lanes
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 1];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle1032(VecN<T, N> v) {  // swap 64-bit halves of each block
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 2];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle01(VecN<T, N> v) {  // 64-bit lanes: swap within each block
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 1];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle0123(VecN<T, N> v) {  // reverse each block of four
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i ^ 3];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle2103(VecN<T, N> v) {  // rotate right by one lane within each block of four
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[(i & ~size_t(3)) + ((i + 3) & 3)];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> Shuffle0321(VecN<T, N> v) {  // rotate left by one lane within each block of four
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[(i & ~size_t(3)) + ((i + 1) & 3)];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> DupEven(VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i & ~size_t(1)];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> DupOdd(VecN<T, N> v) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[i | 1];
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> OddEven(VecN<T, N> odd, VecN<T, N> even) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = (i & 1) ? odd.raw[i] : even.raw[i];
  return r;
}
template <int kLanes, class D, typename T, size_t N>
HWY_API VecN<T, N> ShiftLeftLanes(D, VecN<T, N> v) {  // per block
  constexpr size_t B = BlockLanes<T, N>();
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = (i % B) >= size_t(kLanes) ? v.raw[i - kLanes] : T(0);
  return r;
}
template <int kLanes, class D, typename T, size_t N>
HWY_API VecN<T, N> ShiftRightLanes(D, VecN<T, N> v) {  // per block
  constexpr size_t B = BlockLanes<T, N>();
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = (i % B) + kLanes < B ? v.raw[i + kLanes] : T(0);
  return r;
}
template <typename T, size_t N, typename TI, size_t NI>
HWY_API VecN<TI, NI> TableLookupBytes(VecN<T, N> bytes, VecN<TI, NI> from) {  // per 128-bit block of `from`
  uint8_t tab[N * sizeof(T)], idx[NI * sizeof(TI)], out[NI * sizeof(TI)];
  memcpy(tab, &bytes.raw, sizeof(tab));
  memcpy(idx, &from.raw, sizeof(idx));
  for (size_t i = 0; i < sizeof(idx); i++) {
    const size_t blk = (i / 16 * 16) < sizeof(tab) ? (i / 16 * 16) : 0;
    out[i] = tab[blk + (idx[i] & 15)];
  }
  VecN<TI, NI> r;
  memcpy(&r.raw, out, sizeof(out));
  return r;
}
template <typename T, size_t N, typename TI, size_t NI>
HWY_API VecN<TI, NI> TableLookupBytesOr0(VecN<T, N> bytes, VecN<TI, NI> from) {
  uint8_t tab[N * sizeof(T)], idx[NI * sizeof(TI)], out[NI * sizeof(TI)];
  memcpy(tab, &bytes.raw, sizeof(tab));
  memcpy(idx, &from.raw, sizeof(idx));
  for (size_t i = 0; i < sizeof(idx); i++) {
    const size_t blk = (i / 16 * 16) < sizeof(tab) ? (i / 16 * 16) : 0;
    out[i] = (idx[i] & 0x80) ? 0 : tab[blk + (idx[i] & 15)];
  }
  VecN<TI, NI> r;
  memcpy(&r.raw, out, sizeof(out));
  return r;
}
template <typename T, size_t N>
struct IndicesN {
  MakeSigned<T> raw[N];
};
template <class D, typename TI>
HWY_API IndicesN<TFromD<D>, D::kPrivateLanes> SetTableIndices(D, const TI* idx) {
  IndicesN<TFromD<D>, D::kPrivateLanes> r;
  HWY_SHIM_FOR(i, D::kPrivateLanes) r.raw[i] = static_cast<MakeSigned<TFromD<D>>>(idx[i]);
  return r;
}
template <class D, typename TI, size_t N>
HWY_API IndicesN<TFromD<D>, D::kPrivateLanes> IndicesFromVec(D, VecN<TI, N> v) {
  IndicesN<TFromD<D>, D::kPrivateLanes> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = static_cast<MakeSigned<TFromD<D>>>(v.raw[i]);
  return r;
}
template <typename T, size_t N>
HWY_API VecN<T, N> TableLookupLanes(VecN<T, N> v, IndicesN<T, N> idx) {
  VecN<T, N> r;
  HWY_SHIM_FOR(i, N) r.raw[i] = v.raw[idx.raw[i]];
  return r;
}

// ---- reductions -----------------------------------------------------------------------------
template <class D>
HWY_API VFromD<D> SumOfLanes(D d, VFromD<D> v) {
  TFromD<D> s = 0;
  HWY_SHIM_FOR(i, D::kPrivateLanes) s = WrapAdd(s, static_cast<TFromD<D>>(v.raw[i]));
  return Set(d, s);
}
template <class D>
HWY_API TFromD<D> ReduceSum(D, VFromD<D> v) {
  TFromD<D> s = 0;
  HWY_SHIM_FOR(i, D::kPrivateLanes) s = WrapAdd(s, static_cast<TFromD<D>>(v.raw[i]));
  return s;
}
template <class D>
HWY_API VFromD<D> MinOfLanes(D d, VFromD<D> v) {
  TFromD<D> s = v.raw[0];
  HWY_SHIM_FOR(i, D::kPrivateLanes) s = v.raw[i] < s ? static_cast<TFromD<D>>(v.raw[i]) : s;
  return Set(d, s);
}
template <class D>
HWY_API VFromD<D> MaxOfLanes(D d, VFromD<D> v) {
  TFromD<D> s = v.raw[0];
  HWY_SHIM_FOR(i, D::kPrivateLanes) s = s < v.raw[i] ? static_cast<TFromD<D>>(v.raw[i]) : s;
  return Set(d, s);
}
template <class D>
HWY_API size_t CompressStore(VFromD<D> v, MFromD<D> m, D, TFromD<D>* HWY_RESTRICT p) {
  size_t n = 0;
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (m.raw[i]) p[n++] = v.raw[i];
  return n;
}
template <class D>
HWY_API size_t StoreMaskBits(D, MFromD<D> m, uint8_t* bits) {
  constexpr size_t kBytes = (D::kPrivateLanes + 7) / 8;
  for (size_t i = 0; i < kBytes; i++) bits[i] = 0;
  HWY_SHIM_FOR(i, D::kPrivateLanes) if (m.raw[i]) bits[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
  return kBytes;
}

}  // namespace N_AVX2
}  // namespace hwy

#endif  // ORACLE_HWY_SHIM_V_HIGHWAY_H_
