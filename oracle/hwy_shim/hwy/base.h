// TEST INFRASTRUCTURE ONLY (part of oracle/): a minimal single-lane stand-in
// for Google Highway's <hwy/base.h>, written from scratch so that the libjxl
// reference sources under /root/reference can be compiled IN PLACE into
// oracle/_ref/ without the un-vendored third_party/highway submodule.
// It implements only the subset of the Highway API that libjxl's decoder uses,
// with the semantics of Highway's HWY_SCALAR target (vectors of ONE lane).
// Nothing here is shipped or used by the product path.
#ifndef ORACLE_HWY_SHIM_BASE_H_
#define ORACLE_HWY_SHIM_BASE_H_

#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <cstring>

#define HWY_MAJOR 1
#define HWY_MINOR 2
#define HWY_PATCH 0

#define HWY_COMPILER_GCC 1300
#define HWY_COMPILER_GCC_ACTUAL 1300
#define HWY_COMPILER_CLANG 0
#define HWY_COMPILER_MSVC 0
#define HWY_ARCH_X86 1
#define HWY_ARCH_X86_64 1
#define HWY_ARCH_ARM 0
#define HWY_IS_LITTLE_ENDIAN 1
#define HWY_IS_BIG_ENDIAN 0

#define HWY_RESTRICT __restrict__
#define HWY_INLINE inline __attribute__((always_inline))
#define HWY_NOINLINE __attribute__((noinline))
#define HWY_FLATTEN __attribute__((flatten))
#define HWY_MAYBE_UNUSED __attribute__((unused))
#define HWY_LIKELY(x) __builtin_expect(!!(x), 1)
#define HWY_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define HWY_API static HWY_INLINE HWY_FLATTEN HWY_MAYBE_UNUSED
#define HWY_DLLEXPORT
#define HWY_CONTRIB_DLLEXPORT
#define HWY_ATTR
#define HWY_FENCE std::atomic_thread_fence(std::memory_order_acq_rel)
#define HWY_MIN(a, b) ((a) < (b) ? (a) : (b))
#define HWY_MAX(a, b) ((a) > (b) ? (a) : (b))
#define HWY_ASSERT(c) \
  do {                \
    if (!(c)) __builtin_trap(); \
  } while (0)
#define HWY_DASSERT(c) \
  do {                 \
  } while (0)
#define HWY_ABORT(...) __builtin_trap()
#define HWY_REP4(literal) literal, literal, literal, literal

#define HWY_ALIGNMENT 64
#define HWY_ALIGN_MAX alignas(64)
#define HWY_ALIGN alignas(16)
#define HWY_MAX_BYTES 16
#define HWY_LANES(T) 1

namespace hwy {

static constexpr size_t kMaxVectorSize = 64;

// 16-bit floating point carrier (binary16 bit pattern).
struct float16_t {
  uint16_t bits;
};
struct bfloat16_t {
  uint16_t bits;
};

template <typename T>
struct SizeTag {};

template <bool B, class T = void>
struct EnableIfT {};
template <class T>
struct EnableIfT<true, T> {
  using type = T;
};
template <bool B, class T = void>
using EnableIf = typename EnableIfT<B, T>::type;

template <typename T>
HWY_API size_t PopCount(T x) {
  return static_cast<size_t>(__builtin_popcountll(static_cast<uint64_t>(x)));
}
HWY_API size_t Num0BitsBelowLS1Bit_Nonzero32(uint32_t x) {
  return static_cast<size_t>(__builtin_ctz(x));
}
HWY_API size_t Num0BitsBelowLS1Bit_Nonzero64(uint64_t x) {
  return static_cast<size_t>(__builtin_ctzll(x));
}
HWY_API size_t Num0BitsAboveMS1Bit_Nonzero32(uint32_t x) {
  return static_cast<size_t>(__builtin_clz(x));
}
HWY_API size_t Num0BitsAboveMS1Bit_Nonzero64(uint64_t x) {
  return static_cast<size_t>(__builtin_clzll(x));
}

template <typename To, typename From>
HWY_API void CopyBytesTo(const From* from, To* to) {
  memcpy(to, from, sizeof(From) < sizeof(To) ? sizeof(From) : sizeof(To));
}
template <size_t kBytes, typename From, typename To>
HWY_API void CopyBytes(const From* from, To* to) {
  memcpy(to, from, kBytes);
}
template <typename From, typename To>
HWY_API void CopySameSize(const From* from, To* to) {
  static_assert(sizeof(From) == sizeof(To), "");
  memcpy(to, from, sizeof(From));
}

template <typename T>
struct MakeSignedT;
template <typename T>
struct MakeUnsignedT;
template <typename T>
struct MakeFloatT;
template <typename T>
struct MakeWideT;
template <typename T>
struct MakeNarrowT;
#define ORACLE_HWY_TYPES(T, S, U, F, W, N) \
  template <>                              \
  struct MakeSignedT<T> {                  \
    using type = S;                        \
  };                                       \
  template <>                              \
  struct MakeUnsignedT<T> {                \
    using type = U;                        \
  };                                       \
  template <>                              \
  struct MakeFloatT<T> {                   \
    using type = F;                        \
  };                                       \
  template <>                              \
  struct MakeWideT<T> {                    \
    using type = W;                        \
  };                                       \
  template <>                              \
  struct MakeNarrowT<T> {                  \
    using type = N;                        \
  };
ORACLE_HWY_TYPES(uint8_t, int8_t, uint8_t, void, uint16_t, void)
ORACLE_HWY_TYPES(int8_t, int8_t, uint8_t, void, int16_t, void)
ORACLE_HWY_TYPES(uint16_t, int16_t, uint16_t, float16_t, uint32_t, uint8_t)
ORACLE_HWY_TYPES(int16_t, int16_t, uint16_t, float16_t, int32_t, int8_t)
ORACLE_HWY_TYPES(uint32_t, int32_t, uint32_t, float, uint64_t, uint16_t)
ORACLE_HWY_TYPES(int32_t, int32_t, uint32_t, float, int64_t, int16_t)
ORACLE_HWY_TYPES(uint64_t, int64_t, uint64_t, double, void, uint32_t)
ORACLE_HWY_TYPES(int64_t, int64_t, uint64_t, double, void, int32_t)
ORACLE_HWY_TYPES(float, int32_t, uint32_t, float, double, float16_t)
ORACLE_HWY_TYPES(double, int64_t, uint64_t, double, void, float)
ORACLE_HWY_TYPES(float16_t, int16_t, uint16_t, float16_t, float, void)
#undef ORACLE_HWY_TYPES
template <typename T>
using MakeSigned = typename MakeSignedT<T>::type;
template <typename T>
using MakeUnsigned = typename MakeUnsignedT<T>::type;
template <typename T>
using MakeFloat = typename MakeFloatT<T>::type;
template <typename T>
using MakeWide = typename MakeWideT<T>::type;
template <typename T>
using MakeNarrow = typename MakeNarrowT<T>::type;

template <typename T>
constexpr bool IsFloat() {
  return static_cast<T>(0.25) != static_cast<T>(0);
}
template <>
constexpr bool IsFloat<float16_t>() {
  return true;
}
template <typename T>
constexpr bool IsSigned() {
  return static_cast<T>(-1) < static_cast<T>(0);
}
template <>
constexpr bool IsSigned<float16_t>() {
  return true;
}

}  // namespace hwy

#endif  // ORACLE_HWY_SHIM_BASE_H_
