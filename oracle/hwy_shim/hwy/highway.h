// TEST INFRASTRUCTURE ONLY (part of oracle/): a from-scratch, single-lane
// stand-in for Google Highway's <hwy/highway.h>, sufficient to compile the
// libjxl reference decoder sources IN PLACE (see base.h, oracle/build_ref.py).
//
// Semantics follow Highway's HWY_SCALAR target: every vector has exactly one
// lane, so Lanes(d) == 1 for every descriptor and the cross-lane operations
// (interleave, concat, shuffles) that libjxl itself guards with
// `#if HWY_TARGET != HWY_SCALAR` do not exist.  Deliberate choices:
//   * MulAdd / NegMulAdd / MulSub are single-rounding FMAs (fmaf), as on every
//     FMA-capable SIMD target libjxl is deployed on (AVX2, AVX-512, NEON);
//   * ApproximateReciprocal(Sqrt) are exact (1/x), the upper bound on the
//     accuracy of the hardware estimates.
#ifndef ORACLE_HWY_SHIM_HIGHWAY_H_
#define ORACLE_HWY_SHIM_HIGHWAY_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>

#include "hwy/base.h"
#include "hwy/cache_control.h"

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

#define HWY_TARGET HWY_SCALAR
#define HWY_STATIC_TARGET HWY_SCALAR
#define HWY_TARGETS HWY_SCALAR
#define HWY_NAMESPACE N_SCALAR
#define HWY_ONCE 1
#define HWY_IDE 0

#define HWY_CAP_GE256 0
#define HWY_CAP_GE512 0
#define HWY_CAP_INTEGER64 1
#define HWY_CAP_FLOAT16 0
#define HWY_CAP_FLOAT64 1
#define HWY_HAVE_SCALABLE 0
#define HWY_HAVE_INTEGER64 1
#define HWY_HAVE_FLOAT16 0
#define HWY_HAVE_FLOAT64 1
#define HWY_MEM_OPS_MIGHT_FAULT 0
#define HWY_NATIVE_FMA 1

#define HWY_BEFORE_NAMESPACE() static_assert(true, "hwy shim")
#define HWY_AFTER_NAMESPACE() static_assert(true, "hwy shim")
#define HWY_EXPORT(FUNC) static_assert(true, "hwy shim")
#define HWY_EXPORT_T(TABLE, FUNC) static_assert(true, "hwy shim")
#define HWY_STATIC_DISPATCH(FUNC) N_SCALAR::FUNC
#define HWY_DYNAMIC_DISPATCH(FUNC) N_SCALAR::FUNC
#define HWY_DYNAMIC_POINTER(FUNC) (&N_SCALAR::FUNC)
#define HWY_DYNAMIC_DISPATCH_T(TABLE) N_SCALAR::TABLE
#define HWY_EXPORT_AND_DYNAMIC_DISPATCH_T(FUNC) N_SCALAR::FUNC

#define HWY_FULL(T) hwy::N_SCALAR::Simd<T, 1, 0>
#define HWY_CAPPED(T, N) hwy::N_SCALAR::Simd<T, 1, 0>
#define HWY_FULL1(T) hwy::N_SCALAR::Simd<T, 1, 0>
#define HWY_FULL2(T, LMUL) hwy::N_SCALAR::Simd<T, 1, 0>

namespace hwy {

// Software binary16 <-> binary32 (round to nearest even, IEEE semantics).
static inline float F32FromF16Bits(uint16_t h) {
  const uint32_t sign = static_cast<uint32_t>(h >> 15) << 31;
  const uint32_t exp = (h >> 10) & 0x1F;
  const uint32_t man = h & 0x3FF;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: value = man * 2^-24
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
  if (absx >= 0x7F800000u) {  // inf / nan
    return static_cast<uint16_t>(sign | 0x7C00u | (absx > 0x7F800000u ? 0x200u | ((absx >> 13) & 0x3FF) : 0));
  }
  if (absx >= 0x477FF000u) {  // rounds to >= 65520 -> inf
    return static_cast<uint16_t>(sign | 0x7C00u);
  }
  if (absx < 0x38800000u) {  // below the smallest normal half: subnormal
    // value * 2^24 rounded to nearest even integer
    float a;
    memcpy(&a, &absx, 4);
    const float scaled = a * 16777216.0f;
    const uint32_t m = static_cast<uint32_t>(lrintf(scaled));
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

namespace N_SCALAR {

// ---- descriptors ----------------------------------------------------------
template <typename Lane, size_t N, int kPow2>
struct Simd {
  constexpr Simd() = default;
  using T = Lane;
  static constexpr size_t kPrivateLanes = 1;
  static constexpr int kPrivatePow2 = 0;
  template <typename NewT>
  using Rebind = Simd<NewT, 1, 0>;
  template <typename NewT>
  using Repartition = Simd<NewT, 1, 0>;
  using Half = Simd<Lane, 1, 0>;
  using Twice = Simd<Lane, 1, 0>;
  constexpr size_t MaxLanes() const { return 1; }
  constexpr size_t MaxBytes() const { return sizeof(Lane); }
  constexpr size_t MaxBlocks() const { return 1; }
  constexpr int Pow2() const { return 0; }
};

template <typename T>
using Sisd = Simd<T, 1, 0>;
template <typename T, int kPow2 = 0>
using ScalableTag = Simd<T, 1, 0>;
template <typename T, size_t kLimit, int kPow2 = 0>
using CappedTag = Simd<T, 1, 0>;
template <typename T, size_t kNumLanes>
using FixedTag = Simd<T, 1, 0>;
template <typename T>
using Full16 = Simd<T, 1, 0>;
template <typename T>
using Full32 = Simd<T, 1, 0>;
template <typename T>
using Full64 = Simd<T, 1, 0>;
template <typename T>
using Full128 = Simd<T, 1, 0>;

template <class D>
using TFromD = typename D::T;
template <class T, class D>
using Rebind = Simd<T, 1, 0>;
template <class T, class D>
using Repartition = Simd<T, 1, 0>;
template <class D>
using RebindToSigned = Simd<MakeSigned<TFromD<D>>, 1, 0>;
template <class D>
using RebindToUnsigned = Simd<MakeUnsigned<TFromD<D>>, 1, 0>;
template <class D>
using RebindToFloat = Simd<MakeFloat<TFromD<D>>, 1, 0>;
template <class D>
using RepartitionToWide = Simd<MakeWide<TFromD<D>>, 1, 0>;
template <class D>
using RepartitionToNarrow = Simd<MakeNarrow<TFromD<D>>, 1, 0>;
template <class D>
using Half = D;
template <class D>
using Twice = D;

template <class D>
constexpr size_t Lanes(D) {
  return 1;
}
template <class D>
constexpr size_t MaxLanes(D) {
  return 1;
}
#define HWY_MAX_LANES_D(D) 1

// ---- vector and mask --------------------------------------------------------
template <typename T>
struct Vec1 {
  using PrivateT = T;
  static constexpr size_t kPrivateN = 1;
  Vec1() = default;
  Vec1(const Vec1&) = default;
  Vec1& operator=(const Vec1&) = default;
  explicit Vec1(T t) : raw(t) {}
  T raw;
};
template <typename T>
struct Mask1 {
  bool bit;
};

template <class D>
using VFromD = Vec1<TFromD<D>>;
template <class D>
using Vec = Vec1<TFromD<D>>;
template <class D>
using MFromD = Mask1<TFromD<D>>;
template <class D>
using Mask = Mask1<TFromD<D>>;
template <class V>
using TFromV = typename V::PrivateT;
template <class V>
using DFromV = Simd<typename V::PrivateT, 1, 0>;
template <typename T>
using Vec128 = Vec1<T>;
template <typename T>
using Vec64 = Vec1<T>;
template <typename T>
using Vec32 = Vec1<T>;

// ---- helpers --------------------------------------------------------------------
template <typename T>
struct BitsOf {
  using type = MakeUnsigned<T>;
};
template <typename T>
HWY_API typename BitsOf<T>::type ToBits(T t) {
  typename BitsOf<T>::type u;
  memcpy(&u, &t, sizeof(T));
  return u;
}
template <typename T>
HWY_API T FromBits(typename BitsOf<T>::type u) {
  T t;
  memcpy(&t, &u, sizeof(T));
  return t;
}

// ---- init ---------------------------------------------------------------------------
template <class D, typename T2>
HWY_API VFromD<D> Set(D, T2 t) {
  return VFromD<D>(static_cast<TFromD<D>>(t));
}
template <class D>
HWY_API VFromD<D> Zero(D) {
  VFromD<D> v;
  memset(&v.raw, 0, sizeof(v.raw));
  return v;
}
template <class D>
HWY_API VFromD<D> Undefined(D d) {
  return Zero(d);
}
template <class D, typename T2>
HWY_API VFromD<D> Iota(D, T2 first) {
  return VFromD<D>(static_cast<TFromD<D>>(first));
}
template <class D>
HWY_API VFromD<D> SignBit(D) {
  using T = TFromD<D>;
  using U = MakeUnsigned<T>;
  return VFromD<D>(FromBits<T>(static_cast<U>(U(1) << (sizeof(T) * 8 - 1))));
}
template <typename T>
HWY_API T GetLane(Vec1<T> v) {
  return v.raw;
}
template <typename T>
HWY_API T ExtractLane(Vec1<T> v, size_t) {
  return v.raw;
}
template <typename T>
HWY_API Vec1<T> InsertLane(Vec1<T>, size_t, T t) {
  return Vec1<T>(t);
}

template <class D, typename FromT>
HWY_API VFromD<D> BitCast(D, Vec1<FromT> v) {
  using T = TFromD<D>;
  T out;
  memset(&out, 0, sizeof(T));
  memcpy(&out, &v.raw, sizeof(T) < sizeof(FromT) ? sizeof(T) : sizeof(FromT));
  return VFromD<D>(out);
}
template <class D, typename FromT>
HWY_API VFromD<D> ResizeBitCast(D d, Vec1<FromT> v) {
  return BitCast(d, v);
}

// ---- memory -------------------------------------------------------------------------
template <class D>
HWY_API VFromD<D> Load(D, const TFromD<D>* HWY_RESTRICT p) {
  TFromD<D> t;
  memcpy(&t, p, sizeof(t));
  return VFromD<D>(t);
}
template <class D>
HWY_API VFromD<D> LoadU(D d, const TFromD<D>* HWY_RESTRICT p) {
  return Load(d, p);
}
template <class D>
HWY_API VFromD<D> LoadDup128(D d, const TFromD<D>* HWY_RESTRICT p) {
  return Load(d, p);
}
template <class D>
HWY_API VFromD<D> LoadN(D d, const TFromD<D>* HWY_RESTRICT p, size_t n) {
  return n ? Load(d, p) : Zero(d);
}
template <class D>
HWY_API VFromD<D> MaskedLoad(MFromD<D> m, D d, const TFromD<D>* HWY_RESTRICT p) {
  return m.bit ? Load(d, p) : Zero(d);
}
template <class D>
HWY_API void Store(VFromD<D> v, D, TFromD<D>* HWY_RESTRICT p) {
  memcpy(p, &v.raw, sizeof(v.raw));
}
template <class D>
HWY_API void StoreU(VFromD<D> v, D d, TFromD<D>* HWY_RESTRICT p) {
  Store(v, d, p);
}
template <class D>
HWY_API void Stream(VFromD<D> v, D d, TFromD<D>* HWY_RESTRICT p) {
  Store(v, d, p);
}
template <class D>
HWY_API void StoreN(VFromD<D> v, D d, TFromD<D>* HWY_RESTRICT p, size_t n) {
  if (n) Store(v, d, p);
}
template <class D>
HWY_API void BlendedStore(VFromD<D> v, MFromD<D> m, D d, TFromD<D>* HWY_RESTRICT p) {
  if (m.bit) Store(v, d, p);
}
template <class D, typename TI>
HWY_API VFromD<D> GatherIndex(D d, const TFromD<D>* HWY_RESTRICT base, Vec1<TI> index) {
  return Load(d, base + index.raw);
}
template <class D, typename TI>
HWY_API VFromD<D> GatherOffset(D d, const TFromD<D>* HWY_RESTRICT base, Vec1<TI> offset) {
  return Load(d, reinterpret_cast<const TFromD<D>*>(reinterpret_cast<const uint8_t*>(base) + offset.raw));
}
template <class D>
HWY_API void StoreInterleaved2(VFromD<D> v0, VFromD<D> v1, D d, TFromD<D>* HWY_RESTRICT p) {
  Store(v0, d, p);
  Store(v1, d, p + 1);
}
template <class D>
HWY_API void StoreInterleaved3(VFromD<D> v0, VFromD<D> v1, VFromD<D> v2, D d,
                               TFromD<D>* HWY_RESTRICT p) {
  Store(v0, d, p);
  Store(v1, d, p + 1);
  Store(v2, d, p + 2);
}
template <class D>
HWY_API void StoreInterleaved4(VFromD<D> v0, VFromD<D> v1, VFromD<D> v2, VFromD<D> v3, D d,
                               TFromD<D>* HWY_RESTRICT p) {
  Store(v0, d, p);
  Store(v1, d, p + 1);
  Store(v2, d, p + 2);
  Store(v3, d, p + 3);
}
template <class D>
HWY_API void LoadInterleaved2(D d, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1) {
  v0 = Load(d, p);
  v1 = Load(d, p + 1);
}
template <class D>
HWY_API void LoadInterleaved3(D d, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1,
                              VFromD<D>& v2) {
  v0 = Load(d, p);
  v1 = Load(d, p + 1);
  v2 = Load(d, p + 2);
}
template <class D>
HWY_API void LoadInterleaved4(D d, const TFromD<D>* HWY_RESTRICT p, VFromD<D>& v0, VFromD<D>& v1,
                              VFromD<D>& v2, VFromD<D>& v3) {
  v0 = Load(d, p);
  v1 = Load(d, p + 1);
  v2 = Load(d, p + 2);
  v3 = Load(d, p + 3);
}

// ---- logical ---------------------------------------------------------------------------
template <typename T>
HWY_API Vec1<T> Not(Vec1<T> v) {
  return Vec1<T>(FromBits<T>(static_cast<MakeUnsigned<T>>(~ToBits(v.raw))));
}
template <typename T>
HWY_API Vec1<T> And(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(FromBits<T>(static_cast<MakeUnsigned<T>>(ToBits(a.raw) & ToBits(b.raw))));
}
template <typename T>
HWY_API Vec1<T> AndNot(Vec1<T> not_a, Vec1<T> b) {
  return Vec1<T>(FromBits<T>(static_cast<MakeUnsigned<T>>(~ToBits(not_a.raw) & ToBits(b.raw))));
}
template <typename T>
HWY_API Vec1<T> Or(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(FromBits<T>(static_cast<MakeUnsigned<T>>(ToBits(a.raw) | ToBits(b.raw))));
}
template <typename T>
HWY_API Vec1<T> Xor(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(FromBits<T>(static_cast<MakeUnsigned<T>>(ToBits(a.raw) ^ ToBits(b.raw))));
}
template <typename T>
HWY_API Vec1<T> Or3(Vec1<T> a, Vec1<T> b, Vec1<T> c) {
  return Or(a, Or(b, c));
}
template <typename T>
HWY_API Vec1<T> Xor3(Vec1<T> a, Vec1<T> b, Vec1<T> c) {
  return Xor(a, Xor(b, c));
}
template <typename T>
HWY_API Vec1<T> OrAnd(Vec1<T> o, Vec1<T> a1, Vec1<T> a2) {
  return Or(o, And(a1, a2));
}
template <typename T>
HWY_API Vec1<T> CopySign(Vec1<T> magn, Vec1<T> sign) {
  const auto msb = SignBit(DFromV<Vec1<T>>());
  return Or(AndNot(msb, magn), And(msb, sign));
}
template <typename T>
HWY_API Vec1<T> CopySignToAbs(Vec1<T> abs, Vec1<T> sign) {
  return Or(abs, And(SignBit(DFromV<Vec1<T>>()), sign));
}
template <typename T>
HWY_API Vec1<T> operator&(Vec1<T> a, Vec1<T> b) {
  return And(a, b);
}
template <typename T>
HWY_API Vec1<T> operator|(Vec1<T> a, Vec1<T> b) {
  return Or(a, b);
}
template <typename T>
HWY_API Vec1<T> operator^(Vec1<T> a, Vec1<T> b) {
  return Xor(a, b);
}

// ---- masks -----------------------------------------------------------------------------
template <typename T>
HWY_API Mask1<T> MaskFromVec(Vec1<T> v) {
  // Highway: lane must be all-ones or all-zero; take the MSB like the SIMD targets
  return Mask1<T>{(ToBits(v.raw) >> (sizeof(T) * 8 - 1)) != 0};
}
template <typename T>
HWY_API Vec1<T> VecFromMask(Mask1<T> m) {
  using U = MakeUnsigned<T>;
  return Vec1<T>(FromBits<T>(m.bit ? static_cast<U>(~U(0)) : U(0)));
}
template <class D>
HWY_API VFromD<D> VecFromMask(D, MFromD<D> m) {
  return VecFromMask(m);
}
template <class D, typename TFrom>
HWY_API MFromD<D> RebindMask(D, Mask1<TFrom> m) {
  return MFromD<D>{m.bit};
}
template <class D>
HWY_API MFromD<D> FirstN(D, size_t n) {
  return MFromD<D>{n != 0};
}
template <typename T>
HWY_API Vec1<T> IfThenElse(Mask1<T> m, Vec1<T> yes, Vec1<T> no) {
  return m.bit ? yes : no;
}
template <typename T>
HWY_API Vec1<T> IfThenElseZero(Mask1<T> m, Vec1<T> yes) {
  return m.bit ? yes : Zero(DFromV<Vec1<T>>());
}
template <typename T>
HWY_API Vec1<T> IfThenZeroElse(Mask1<T> m, Vec1<T> no) {
  return m.bit ? Zero(DFromV<Vec1<T>>()) : no;
}
template <typename T>
HWY_API Vec1<T> IfVecThenElse(Vec1<T> mask, Vec1<T> yes, Vec1<T> no) {
  return Or(And(mask, yes), AndNot(mask, no));
}
template <typename T>
HWY_API Vec1<T> IfNegativeThenElse(Vec1<T> v, Vec1<T> yes, Vec1<T> no) {
  return MaskFromVec(v).bit ? yes : no;
}
template <typename T>
HWY_API Vec1<T> ZeroIfNegative(Vec1<T> v) {
  return v.raw < T(0) ? Vec1<T>(T(0)) : v;
}
template <typename T>
HWY_API Mask1<T> Not(Mask1<T> m) {
  return Mask1<T>{!m.bit};
}
template <typename T>
HWY_API Mask1<T> And(Mask1<T> a, Mask1<T> b) {
  return Mask1<T>{a.bit && b.bit};
}
template <typename T>
HWY_API Mask1<T> AndNot(Mask1<T> a, Mask1<T> b) {
  return Mask1<T>{!a.bit && b.bit};
}
template <typename T>
HWY_API Mask1<T> Or(Mask1<T> a, Mask1<T> b) {
  return Mask1<T>{a.bit || b.bit};
}
template <typename T>
HWY_API Mask1<T> Xor(Mask1<T> a, Mask1<T> b) {
  return Mask1<T>{a.bit != b.bit};
}
template <class D>
HWY_API bool AllFalse(D, MFromD<D> m) {
  return !m.bit;
}
template <class D>
HWY_API bool AllTrue(D, MFromD<D> m) {
  return m.bit;
}
template <class D>
HWY_API size_t CountTrue(D, MFromD<D> m) {
  return m.bit ? 1 : 0;
}
template <class D>
HWY_API intptr_t FindFirstTrue(D, MFromD<D> m) {
  return m.bit ? 0 : -1;
}
template <class D>
HWY_API size_t StoreMaskBits(D, MFromD<D> m, uint8_t* bits) {
  *bits = m.bit ? 1 : 0;
  return 1;
}

// ---- comparisons ---------------------------------------------------------------------------
#define ORACLE_HWY_CMP(NAME, OP)                 \
  template <typename T>                          \
  HWY_API Mask1<T> NAME(Vec1<T> a, Vec1<T> b) {  \
    return Mask1<T>{a.raw OP b.raw};             \
  }
ORACLE_HWY_CMP(Eq, ==)
ORACLE_HWY_CMP(Ne, !=)
ORACLE_HWY_CMP(Lt, <)
ORACLE_HWY_CMP(Le, <=)
ORACLE_HWY_CMP(Gt, >)
ORACLE_HWY_CMP(Ge, >=)
ORACLE_HWY_CMP(operator==, ==)
ORACLE_HWY_CMP(operator!=, !=)
ORACLE_HWY_CMP(operator<, <)
ORACLE_HWY_CMP(operator<=, <=)
ORACLE_HWY_CMP(operator>, >)
ORACLE_HWY_CMP(operator>=, >=)
#undef ORACLE_HWY_CMP
template <typename T>
HWY_API Mask1<T> TestBit(Vec1<T> v, Vec1<T> bit) {
  return Mask1<T>{(ToBits(v.raw) & ToBits(bit.raw)) != 0};
}
template <typename T>
HWY_API Mask1<T> IsNaN(Vec1<T> v) {
  return Mask1<T>{v.raw != v.raw};
}
template <typename T>
HWY_API Mask1<T> IsInf(Vec1<T> v) {
  return Mask1<T>{std::isinf(v.raw)};
}
template <typename T>
HWY_API Mask1<T> IsFinite(Vec1<T> v) {
  return Mask1<T>{std::isfinite(v.raw)};
}

// ---- arithmetic ---------------------------------------------------------------------------
namespace detail {
template <typename T, bool kFloat = std::is_floating_point<T>::value>
struct Arith {
  // integers wrap (computed in the unsigned domain)
  using U = MakeUnsigned<T>;
  static T Add(T a, T b) { return static_cast<T>(static_cast<U>(static_cast<U>(a) + static_cast<U>(b))); }
  static T Sub(T a, T b) { return static_cast<T>(static_cast<U>(static_cast<U>(a) - static_cast<U>(b))); }
  static T Mul(T a, T b) {
    return static_cast<T>(static_cast<U>(static_cast<uint64_t>(static_cast<U>(a)) * static_cast<uint64_t>(static_cast<U>(b))));
  }
  static T Neg(T a) { return static_cast<T>(static_cast<U>(U(0) - static_cast<U>(a))); }
};
template <typename T>
struct Arith<T, true> {
  static T Add(T a, T b) { return a + b; }
  static T Sub(T a, T b) { return a - b; }
  static T Mul(T a, T b) { return a * b; }
  static T Neg(T a) { return -a; }
};
}  // namespace detail

template <typename T>
HWY_API Vec1<T> Add(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(detail::Arith<T>::Add(a.raw, b.raw));
}
template <typename T>
HWY_API Vec1<T> Sub(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(detail::Arith<T>::Sub(a.raw, b.raw));
}
template <typename T>
HWY_API Vec1<T> Mul(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(detail::Arith<T>::Mul(a.raw, b.raw));
}
template <typename T>
HWY_API Vec1<T> Div(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(a.raw / b.raw);
}
template <typename T>
HWY_API Vec1<T> Neg(Vec1<T> a) {
  return Vec1<T>(detail::Arith<T>::Neg(a.raw));
}
template <typename T>
HWY_API Vec1<T> operator+(Vec1<T> a, Vec1<T> b) {
  return Add(a, b);
}
template <typename T>
HWY_API Vec1<T> operator-(Vec1<T> a, Vec1<T> b) {
  return Sub(a, b);
}
template <typename T>
HWY_API Vec1<T> operator*(Vec1<T> a, Vec1<T> b) {
  return Mul(a, b);
}
template <typename T>
HWY_API Vec1<T> operator/(Vec1<T> a, Vec1<T> b) {
  return Div(a, b);
}
template <typename T>
HWY_API Vec1<T> SaturatedAdd(Vec1<T> a, Vec1<T> b) {
  const int64_t s = static_cast<int64_t>(a.raw) + static_cast<int64_t>(b.raw);
  const int64_t lo = std::numeric_limits<T>::min(), hi = std::numeric_limits<T>::max();
  return Vec1<T>(static_cast<T>(s < lo ? lo : (s > hi ? hi : s)));
}
template <typename T>
HWY_API Vec1<T> SaturatedSub(Vec1<T> a, Vec1<T> b) {
  const int64_t s = static_cast<int64_t>(a.raw) - static_cast<int64_t>(b.raw);
  const int64_t lo = std::numeric_limits<T>::min(), hi = std::numeric_limits<T>::max();
  return Vec1<T>(static_cast<T>(s < lo ? lo : (s > hi ? hi : s)));
}
template <typename T>
HWY_API Vec1<T> AverageRound(Vec1<T> a, Vec1<T> b) {
  return Vec1<T>(static_cast<T>((static_cast<uint64_t>(a.raw) + b.raw + 1) >> 1));
}
template <typename T>
HWY_API Vec1<T> Abs(Vec1<T> a) {
  if (std::is_floating_point<T>::value) {
    return AndNot(SignBit(DFromV<Vec1<T>>()), a);
  }
  return Vec1<T>(a.raw < T(0) ? detail::Arith<T>::Neg(a.raw) : a.raw);
}
template <typename T>
HWY_API Vec1<T> AbsDiff(Vec1<T> a, Vec1<T> b) {
  return Abs(Sub(a, b));
}
template <typename T>
HWY_API Vec1<T> Min(Vec1<T> a, Vec1<T> b) {
  if (std::is_floating_point<T>::value) {
    if (a.raw != a.raw) return b;
    if (b.raw != b.raw) return a;
  }
  return b.raw < a.raw ? b : a;
}
template <typename T>
HWY_API Vec1<T> Max(Vec1<T> a, Vec1<T> b) {
  if (std::is_floating_point<T>::value) {
    if (a.raw != a.raw) return b;
    if (b.raw != b.raw) return a;
  }
  return a.raw < b.raw ? b : a;
}
template <typename T>
HWY_API Vec1<T> Clamp(Vec1<T> v, Vec1<T> lo, Vec1<T> hi) {
  return Min(Max(lo, v), hi);
}

HWY_API Vec1<float> MulAdd(Vec1<float> mul, Vec1<float> x, Vec1<float> add) {
  return Vec1<float>(fmaf(mul.raw, x.raw, add.raw));
}
HWY_API Vec1<double> MulAdd(Vec1<double> mul, Vec1<double> x, Vec1<double> add) {
  return Vec1<double>(fma(mul.raw, x.raw, add.raw));
}
HWY_API Vec1<float> NegMulAdd(Vec1<float> mul, Vec1<float> x, Vec1<float> add) {
  return Vec1<float>(fmaf(-mul.raw, x.raw, add.raw));
}
HWY_API Vec1<double> NegMulAdd(Vec1<double> mul, Vec1<double> x, Vec1<double> add) {
  return Vec1<double>(fma(-mul.raw, x.raw, add.raw));
}
HWY_API Vec1<float> MulSub(Vec1<float> mul, Vec1<float> x, Vec1<float> sub) {
  return Vec1<float>(fmaf(mul.raw, x.raw, -sub.raw));
}
HWY_API Vec1<double> MulSub(Vec1<double> mul, Vec1<double> x, Vec1<double> sub) {
  return Vec1<double>(fma(mul.raw, x.raw, -sub.raw));
}
HWY_API Vec1<float> NegMulSub(Vec1<float> mul, Vec1<float> x, Vec1<float> sub) {
  return Vec1<float>(fmaf(-mul.raw, x.raw, -sub.raw));
}
// integer MulAdd (used by a few modular-mode helpers)
template <typename T, typename = EnableIf<std::is_integral<T>::value>>
HWY_API Vec1<T> MulAdd(Vec1<T> mul, Vec1<T> x, Vec1<T> add) {
  return Add(Mul(mul, x), add);
}
template <typename T, typename = EnableIf<std::is_integral<T>::value>>
HWY_API Vec1<T> NegMulAdd(Vec1<T> mul, Vec1<T> x, Vec1<T> add) {
  return Sub(add, Mul(mul, x));
}

template <typename T>
HWY_API Vec1<T> Sqrt(Vec1<T> v) {
  return Vec1<T>(std::sqrt(v.raw));
}
HWY_API Vec1<float> ApproximateReciprocal(Vec1<float> v) { return Vec1<float>(1.0f / v.raw); }
HWY_API Vec1<float> ApproximateReciprocalSqrt(Vec1<float> v) {
  return Vec1<float>(1.0f / std::sqrt(v.raw));
}
template <typename T>
HWY_API Vec1<T> Floor(Vec1<T> v) {
  return Vec1<T>(std::floor(v.raw));
}
template <typename T>
HWY_API Vec1<T> Ceil(Vec1<T> v) {
  return Vec1<T>(std::ceil(v.raw));
}
template <typename T>
HWY_API Vec1<T> Trunc(Vec1<T> v) {
  return Vec1<T>(std::trunc(v.raw));
}
template <typename T>
HWY_API Vec1<T> Round(Vec1<T> v) {
  return Vec1<T>(std::nearbyint(v.raw));  // ties to even (default rounding mode)
}
HWY_API Vec1<int32_t> NearestInt(Vec1<float> v) {
  const float f = v.raw;
  if (f != f) return Vec1<int32_t>(0);
  if (f >= 2147483648.0f) return Vec1<int32_t>(std::numeric_limits<int32_t>::max());
  if (f <= -2147483648.0f) return Vec1<int32_t>(std::numeric_limits<int32_t>::min());
  return Vec1<int32_t>(static_cast<int32_t>(lrintf(f)));
}

// 16x16 -> high half, widening even/odd multiplies
HWY_API Vec1<int16_t> MulHigh(Vec1<int16_t> a, Vec1<int16_t> b) {
  return Vec1<int16_t>(static_cast<int16_t>((static_cast<int32_t>(a.raw) * b.raw) >> 16));
}
HWY_API Vec1<uint16_t> MulHigh(Vec1<uint16_t> a, Vec1<uint16_t> b) {
  return Vec1<uint16_t>(static_cast<uint16_t>((static_cast<uint32_t>(a.raw) * b.raw) >> 16));
}
HWY_API Vec1<int64_t> MulEven(Vec1<int32_t> a, Vec1<int32_t> b) {
  return Vec1<int64_t>(static_cast<int64_t>(a.raw) * b.raw);
}
HWY_API Vec1<uint64_t> MulEven(Vec1<uint32_t> a, Vec1<uint32_t> b) {
  return Vec1<uint64_t>(static_cast<uint64_t>(a.raw) * b.raw);
}
HWY_API Vec1<uint64_t> MulEven(Vec1<uint64_t> a, Vec1<uint64_t> b) {  // low 64 bits of the product
  return Vec1<uint64_t>(static_cast<uint64_t>(static_cast<unsigned __int128>(a.raw) * b.raw));
}
HWY_API Vec1<uint64_t> MulOdd(Vec1<uint64_t> a, Vec1<uint64_t> b) {  // high 64 bits (lane pair of 1)
  return Vec1<uint64_t>(static_cast<uint64_t>((static_cast<unsigned __int128>(a.raw) * b.raw) >> 64));
}

// ---- shifts ---------------------------------------------------------------------------
template <int kBits, typename T>
HWY_API Vec1<T> ShiftLeft(Vec1<T> v) {
  using U = MakeUnsigned<T>;
  return Vec1<T>(static_cast<T>(static_cast<U>(static_cast<U>(v.raw) << kBits)));
}
template <int kBits, typename T>
HWY_API Vec1<T> ShiftRight(Vec1<T> v) {
  return Vec1<T>(static_cast<T>(v.raw >> kBits));  // arithmetic for signed (gcc)
}
template <typename T>
HWY_API Vec1<T> ShiftLeftSame(Vec1<T> v, int bits) {
  using U = MakeUnsigned<T>;
  return Vec1<T>(static_cast<T>(static_cast<U>(static_cast<U>(v.raw) << bits)));
}
template <typename T>
HWY_API Vec1<T> ShiftRightSame(Vec1<T> v, int bits) {
  return Vec1<T>(static_cast<T>(v.raw >> bits));
}
template <typename T>
HWY_API Vec1<T> Shl(Vec1<T> v, Vec1<T> bits) {
  return ShiftLeftSame(v, static_cast<int>(bits.raw));
}
template <typename T>
HWY_API Vec1<T> Shr(Vec1<T> v, Vec1<T> bits) {
  return ShiftRightSame(v, static_cast<int>(bits.raw));
}
template <typename T>
HWY_API Vec1<T> operator<<(Vec1<T> v, Vec1<T> bits) {
  return Shl(v, bits);
}
template <typename T>
HWY_API Vec1<T> operator>>(Vec1<T> v, Vec1<T> bits) {
  return Shr(v, bits);
}
template <int kBits, typename T>
HWY_API Vec1<T> RotateRight(Vec1<T> v) {
  using U = MakeUnsigned<T>;
  constexpr int kN = static_cast<int>(sizeof(T) * 8);
  const U u = static_cast<U>(v.raw);
  return Vec1<T>(static_cast<T>(kBits == 0 ? u : static_cast<U>((u >> kBits) | (u << ((kN - kBits) % kN)))));
}
template <typename T>
HWY_API Vec1<T> BroadcastSignBit(Vec1<T> v) {
  return Vec1<T>(v.raw < 0 ? T(-1) : T(0));
}

// ---- conversions ---------------------------------------------------------------------------
namespace detail {
template <typename To, typename From>
HWY_API To SatCast(From f) {
  // integer -> narrower integer with saturation
  const From lo = static_cast<From>(std::numeric_limits<To>::min() < 0 && std::is_signed<From>::value
                                        ? (sizeof(To) < sizeof(From) ? static_cast<From>(std::numeric_limits<To>::min())
                                                                     : std::numeric_limits<From>::min())
                                        : From(0));
  if (std::is_signed<From>::value && f < lo) return static_cast<To>(lo);
  using Wide = typename std::conditional<std::is_signed<From>::value, int64_t, uint64_t>::type;
  const Wide hi = static_cast<Wide>(std::numeric_limits<To>::max());
  if (static_cast<Wide>(f) > hi && f > From(0)) return std::numeric_limits<To>::max();
  return static_cast<To>(f);
}
template <typename ToI, typename F>
HWY_API ToI FloatToIntSat(F f) {
  if (f != f) return ToI(0);
  const F hi = static_cast<F>(std::numeric_limits<ToI>::max());
  const F lo = static_cast<F>(std::numeric_limits<ToI>::min());
  if (f >= hi) return std::numeric_limits<ToI>::max();
  if (f <= lo) return std::numeric_limits<ToI>::min();
  return static_cast<ToI>(f);  // truncation toward zero
}
}  // namespace detail

// PromoteTo: widen
template <class D, typename FromT,
          EnableIf<std::is_arithmetic<FromT>::value && std::is_arithmetic<TFromD<D>>::value>* = nullptr>
HWY_API VFromD<D> PromoteTo(D, Vec1<FromT> v) {
  return VFromD<D>(static_cast<TFromD<D>>(v.raw));
}
template <class D>
HWY_API Vec1<float> PromoteTo(D, Vec1<float16_t> v) {
  return Vec1<float>(F32FromF16(v.raw));
}
template <class D>
HWY_API Vec1<float> PromoteTo(D, Vec1<bfloat16_t> v) {
  return Vec1<float>(F32FromBF16(v.raw));
}
template <class D, typename FromT>
HWY_API VFromD<D> PromoteLowerTo(D d, Vec1<FromT> v) {
  return PromoteTo(d, v);
}

// DemoteTo: narrow with saturation (integers) / rounding (floats)
template <class D, typename FromT,
          EnableIf<std::is_integral<FromT>::value && std::is_integral<TFromD<D>>::value>* = nullptr>
HWY_API VFromD<D> DemoteTo(D, Vec1<FromT> v) {
  return VFromD<D>(detail::SatCast<TFromD<D>>(v.raw));
}
template <class D, EnableIf<std::is_same<TFromD<D>, float>::value>* = nullptr>
HWY_API Vec1<float> DemoteTo(D, Vec1<double> v) {
  return Vec1<float>(static_cast<float>(v.raw));
}
template <class D, EnableIf<std::is_same<TFromD<D>, int32_t>::value>* = nullptr>
HWY_API Vec1<int32_t> DemoteTo(D, Vec1<double> v) {
  return Vec1<int32_t>(detail::FloatToIntSat<int32_t>(v.raw));
}
template <class D, EnableIf<std::is_same<TFromD<D>, float16_t>::value>* = nullptr>
HWY_API Vec1<float16_t> DemoteTo(D, Vec1<float> v) {
  Vec1<float16_t> r;
  r.raw = F16FromF32(v.raw);
  return r;
}
template <class D, typename FromT>
HWY_API VFromD<D> TruncateTo(D, Vec1<FromT> v) {
  return VFromD<D>(static_cast<TFromD<D>>(v.raw));
}
HWY_API Vec1<uint8_t> U8FromU32(Vec1<uint32_t> v) { return Vec1<uint8_t>(static_cast<uint8_t>(v.raw)); }

// ConvertTo: same-width int <-> float
template <class D, typename FromT,
          EnableIf<std::is_floating_point<TFromD<D>>::value && std::is_integral<FromT>::value>* = nullptr>
HWY_API VFromD<D> ConvertTo(D, Vec1<FromT> v) {
  return VFromD<D>(static_cast<TFromD<D>>(v.raw));
}
template <class D, typename FromT,
          EnableIf<std::is_integral<TFromD<D>>::value && std::is_floating_point<FromT>::value>* = nullptr>
HWY_API VFromD<D> ConvertTo(D, Vec1<FromT> v) {
  return VFromD<D>(detail::FloatToIntSat<TFromD<D>>(v.raw));
}

// ---- lane permutations that exist for one lane -----------------------------------
template <int kLane, typename T>
HWY_API Vec1<T> Broadcast(Vec1<T> v) {
  static_assert(kLane == 0, "single-lane vector");
  return v;
}
template <typename T>
HWY_API Vec1<T> DupEven(Vec1<T> v) {
  return v;
}
template <typename T>
HWY_API Vec1<T> DupOdd(Vec1<T> v) {
  return v;
}
template <typename T>
HWY_API Vec1<T> OddEven(Vec1<T> /*odd*/, Vec1<T> even) {
  return even;
}
template <typename T>
HWY_API Vec1<T> Reverse(DFromV<Vec1<T>>, Vec1<T> v) {
  return v;
}
template <typename T>
HWY_API Vec1<T> InterleaveLower(Vec1<T> a, Vec1<T>) {
  return a;
}
template <class D>
HWY_API VFromD<D> InterleaveLower(D, VFromD<D> a, VFromD<D>) {
  return a;
}
template <class D>
HWY_API VFromD<D> LowerHalf(D, VFromD<D> v) {
  return v;
}
template <typename T>
HWY_API Vec1<T> LowerHalf(Vec1<T> v) {
  return v;
}
template <class D>
HWY_API VFromD<D> ZeroExtendVector(D, VFromD<D> v) {
  return v;
}
template <typename T, typename TI>
HWY_API Vec1<TI> TableLookupBytes(Vec1<T> bytes, Vec1<TI> from) {
  uint8_t in[sizeof(T)], idx[sizeof(TI)], out[sizeof(TI)];
  memcpy(in, &bytes.raw, sizeof(T));
  memcpy(idx, &from.raw, sizeof(TI));
  for (size_t i = 0; i < sizeof(TI); i++) out[i] = idx[i] < sizeof(T) ? in[idx[i]] : 0;
  TI r;
  memcpy(&r, out, sizeof(TI));
  return Vec1<TI>(r);
}
template <typename T>
struct Indices1 {
  int32_t raw;
};
template <class D, typename TI>
HWY_API Indices1<TFromD<D>> SetTableIndices(D, const TI* idx) {
  return Indices1<TFromD<D>>{static_cast<int32_t>(idx[0])};
}
template <class D, typename TI>
HWY_API Indices1<TFromD<D>> IndicesFromVec(D, Vec1<TI> v) {
  return Indices1<TFromD<D>>{static_cast<int32_t>(v.raw)};
}
template <typename T>
HWY_API Vec1<T> TableLookupLanes(Vec1<T> v, Indices1<T>) {
  return v;
}

// ---- reductions ----------------------------------------------------------------------
template <class D>
HWY_API VFromD<D> SumOfLanes(D, VFromD<D> v) {
  return v;
}
template <class D>
HWY_API VFromD<D> MinOfLanes(D, VFromD<D> v) {
  return v;
}
template <class D>
HWY_API VFromD<D> MaxOfLanes(D, VFromD<D> v) {
  return v;
}
template <class D>
HWY_API TFromD<D> ReduceSum(D, VFromD<D> v) {
  return v.raw;
}
template <class D>
HWY_API TFromD<D> ReduceMin(D, VFromD<D> v) {
  return v.raw;
}
template <class D>
HWY_API TFromD<D> ReduceMax(D, VFromD<D> v) {
  return v.raw;
}

}  // namespace N_SCALAR

// target introspection used by tools/benchmarks
static inline int64_t SupportedTargets() { return HWY_SCALAR; }
static inline const char* TargetName(int64_t) { return "SCALAR(shim)"; }

}  // namespace hwy

#endif  // ORACLE_HWY_SHIM_HIGHWAY_H_
