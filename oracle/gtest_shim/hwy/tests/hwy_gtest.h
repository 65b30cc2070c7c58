// TEST INFRASTRUCTURE ONLY (part of oracle/): stand-in for <hwy/tests/hwy_gtest.h>.  Highway runs a test once per
// compiled SIMD target; the stand-ins have exactly one target per build (HWY_SCALAR in oracle/hwy_shim, HWY_AVX2 in
// oracle/hwy_shim_v), so every target-parameterised suite has one "target" value.
#ifndef ORACLE_GTEST_SHIM_HWY_GTEST_H_
#define ORACLE_GTEST_SHIM_HWY_GTEST_H_

#include <stdint.h>

#include <tuple>

#include "gtest/gtest.h"
#include "hwy/highway.h"

namespace hwy {

class TestWithParamTarget : public ::testing::TestWithParam<int64_t> {};

template <typename T>
class TestWithParamTargetAndT : public ::testing::Test, public ::testing::WithParamInterface<std::tuple<int64_t, T>> {
 public:
  using ParamType = std::tuple<int64_t, T>;
  T GetParam() const { return std::get<1>(::testing::WithParamInterface<std::tuple<int64_t, T>>::GetParam()); }
};

template <typename T>
std::vector<std::tuple<int64_t, T>> WithTarget(const ::testing::ParamList<T>& p) {
  std::vector<std::tuple<int64_t, T>> out;
  for (const T& v : p.v) out.emplace_back(static_cast<int64_t>(HWY_TARGET), v);
  return out;
}

}  // namespace hwy

#define HWY_TARGET_INSTANTIATE_TEST_SUITE_P(suite)                                                          \
  static int GTEST_SHIM_CAT(hwy_gtest_inst_, __LINE__) = [] {                                               \
    ::testing::ParamSuite<suite>::EnsureRegistered(#suite);                                                 \
    ::testing::ParamSuite<suite>::Instances().push_back({"Target", {static_cast<int64_t>(HWY_TARGET)}});    \
    return 0;                                                                                               \
  }()
#define HWY_TARGET_INSTANTIATE_TEST_SUITE_P_T(suite, generator)                                             \
  static int GTEST_SHIM_CAT(hwy_gtest_inst_, __LINE__) = [] {                                               \
    ::testing::ParamSuite<suite>::EnsureRegistered(#suite);                                                 \
    ::testing::ParamSuite<suite>::Instances().push_back({"TargetAndT", ::hwy::WithTarget(generator)});      \
    return 0;                                                                                               \
  }()
#define HWY_EXPORT_AND_TEST_P(suite, func) \
  TEST_P(suite, func) { HWY_DYNAMIC_DISPATCH(func)(); }                                                     \
  static_assert(true, "")
#define HWY_EXPORT_AND_TEST_P_T(suite, func) \
  TEST_P(suite, func) { HWY_DYNAMIC_DISPATCH(func)(GetParam()); }                                           \
  static_assert(true, "")
#define HWY_BEFORE_TEST(suite) class suite : public hwy::TestWithParamTarget {}; HWY_TARGET_INSTANTIATE_TEST_SUITE_P(suite); static_assert(true, "")
#define HWY_AFTER_TEST() static_assert(true, "")
#define HWY_TEST_MAIN() int main(int argc, char** argv) { ::testing::InitGoogleTest(&argc, argv); return RUN_ALL_TESTS(); } static_assert(true, "")

#endif  // ORACLE_GTEST_SHIM_HWY_GTEST_H_
