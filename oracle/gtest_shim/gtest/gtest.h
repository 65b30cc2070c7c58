// TEST INFRASTRUCTURE ONLY (part of oracle/): a from-scratch stand-in for the small part of googletest that the libjxl
// reference's own unit tests of the VarDCT hot path use (dct_test.cc, ac_strategy_test.cc, opsin_inverse_test.cc,
// quant_weights_test.cc: TEST / TEST_P, value-parameterised suites over testing::Range / ValuesIn, EXPECT_* / ASSERT_*
// with streamed messages).  googletest is not installed in this image and is not vendored in /root/reference.
// oracle/build_ref_tests.py compiles those test sources IN PLACE against this header and the Highway stand-ins
// (oracle/hwy_shim: one lane; oracle/hwy_shim_v: eight) and runs them: the reference's own known-answer tests check
// the checkers.
#ifndef ORACLE_GTEST_SHIM_GTEST_H_
#define ORACLE_GTEST_SHIM_GTEST_H_

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

class Message {
 public:
  Message() = default;
  Message(const Message& o) { ss_ << o.ss_.str(); }
  template <typename T>
  Message& operator<<(const T& v) {
    ss_ << v;
    return *this;
  }
  Message& operator<<(std::ostream& (*f)(std::ostream&)) {
    ss_ << f;
    return *this;
  }
  std::string str() const { return ss_.str(); }

 private:
  std::stringstream ss_;
};

struct State {
  int failures_in_test = 0;
  int failed_tests = 0, run_tests = 0, skipped = 0;
  bool skip_current = false;
  static State& Get() {
    static State s;
    return s;
  }
};

// `AssertHelper(...) = Message() << ...`: reports when the temporary is assigned (googletest's own shape)
class AssertHelper {
 public:
  AssertHelper(const char* file, int line, std::string what, bool fatal) : file_(file), line_(line), what_(std::move(what)), fatal_(fatal) {}
  void operator=(const Message& m) const {
    State::Get().failures_in_test++;
    std::fprintf(stderr, "%s:%d: Failure\n%s\n%s\n", file_, line_, what_.c_str(), m.str().c_str());
    (void)fatal_;
  }

 private:
  const char* file_;
  int line_;
  std::string what_;
  bool fatal_;
};
class SkipHelper {
 public:
  void operator=(const Message&) const { State::Get().skip_current = true; }
};

template <typename T>
std::string Print(const T& v) {
  if constexpr (std::is_enum<T>::value) {
    return std::to_string(static_cast<long long>(v));
  } else if constexpr (std::is_arithmetic<T>::value || std::is_convertible<T, std::string>::value) {
    std::stringstream ss;
    ss << v;
    return ss.str();
  } else {
    return "(object)";
  }
}

class Test {
 public:
  virtual ~Test() = default;
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

template <typename T>
class WithParamInterface {
 public:
  using ParamType = T;
  const T& GetParam() const { return *Current(); }
  static const T*& Current() {
    static const T* p = nullptr;
    return p;
  }
};
template <typename T>
class TestWithParam : public Test, public WithParamInterface<T> {};

struct Registry {
  struct Entry {
    std::string name;
    std::function<void()> run;
  };
  static std::vector<Entry>& Tests() {
    static std::vector<Entry> t;
    return t;
  }
  static int Add(std::string name, std::function<void()> run) {
    Tests().push_back({std::move(name), std::move(run)});
    return 0;
  }
};

template <class Fixture>
void RunOne(const std::string& name) {
  State& s = State::Get();
  s.failures_in_test = 0;
  s.skip_current = false;
  Fixture t;
  t.SetUp();
  if (!s.skip_current) t.TestBody();
  t.TearDown();
  s.run_tests++;
  if (s.skip_current) {
    s.skipped++;
    std::printf("[  SKIPPED ] %s\n", name.c_str());
  } else if (s.failures_in_test) {
    s.failed_tests++;
    std::printf("[  FAILED  ] %s\n", name.c_str());
  } else {
    std::printf("[       OK ] %s\n", name.c_str());
  }
}

// ---- value-parameterised suites: TEST_P bodies and INSTANTIATE_TEST_SUITE_P generators meet at static-init time in
// either order, so both are kept per suite and expanded in RUN_ALL_TESTS
template <class Suite>
struct ParamSuite {
  using P = typename Suite::ParamType;
  struct Body {
    std::string name;
    std::function<void(const std::string&)> run;  // runs the body on the current parameter
  };
  static std::vector<Body>& Bodies() {
    static std::vector<Body> b;
    return b;
  }
  static std::vector<std::pair<std::string, std::vector<P>>>& Instances() {
    static std::vector<std::pair<std::string, std::vector<P>>> i;
    return i;
  }
  static bool& Registered() {
    static bool r = false;
    return r;
  }
  static void EnsureRegistered(const char* suite) {
    if (Registered()) return;
    Registered() = true;
    const std::string sname = suite;
    Registry::Add(sname + ".*", [sname] {
      for (auto& inst : Instances())
        for (size_t i = 0; i < inst.second.size(); i++)
          for (auto& b : Bodies()) {
            WithParamInterface<P>::Current() = &inst.second[i];
            b.run(inst.first + "/" + sname + "." + b.name + "/" + std::to_string(i));
          }
    });
  }
};

template <typename T>
struct ParamList {
  std::vector<T> v;
  template <typename U>
  operator std::vector<U>() const {
    return std::vector<U>(v.begin(), v.end());
  }
};
template <typename T>
ParamList<T> Range(T begin, T end) {
  ParamList<T> p;
  for (T i = begin; i < end; i = static_cast<T>(i + 1)) p.v.push_back(i);
  return p;
}
template <typename T, typename... Ts>
ParamList<T> Values(T first, Ts... rest) {
  ParamList<T> p;
  p.v = {first, static_cast<T>(rest)...};
  return p;
}
template <class C>
ParamList<typename C::value_type> ValuesIn(const C& c) {
  ParamList<typename C::value_type> p;
  p.v.assign(c.begin(), c.end());
  return p;
}
inline void InitGoogleTest(int*, char**) {}

}  // namespace testing

inline int RUN_ALL_TESTS() {
  for (auto& t : ::testing::Registry::Tests()) t.run();
  const ::testing::State& s = ::testing::State::Get();
  std::printf("[==========] %d tests ran, %d failed, %d skipped\n", s.run_tests, s.failed_tests, s.skipped);
  return s.failed_tests ? 1 : 0;
}

#define GTEST_SHIM_CAT_(a, b) a##b
#define GTEST_SHIM_CAT(a, b) GTEST_SHIM_CAT_(a, b)
#define GTEST_SHIM_CLASS(suite, name) suite##_##name##_Test

#define GTEST_SHIM_TEST_(suite, name, parent)                                                               \
  class GTEST_SHIM_CLASS(suite, name) : public parent {                                                     \
   public:                                                                                                  \
    void TestBody() override;                                                                               \
  };                                                                                                        \
  static int GTEST_SHIM_CAT(gtest_shim_reg_, __LINE__) = ::testing::Registry::Add(                          \
      #suite "." #name, [] { ::testing::RunOne<GTEST_SHIM_CLASS(suite, name)>(#suite "." #name); });        \
  void GTEST_SHIM_CLASS(suite, name)::TestBody()
#define TEST(suite, name) GTEST_SHIM_TEST_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GTEST_SHIM_TEST_(fixture, name, fixture)

#define TEST_P(suite, name)                                                                                 \
  class GTEST_SHIM_CLASS(suite, name) : public suite {                                                      \
   public:                                                                                                  \
    void TestBody() override;                                                                               \
  };                                                                                                        \
  static int GTEST_SHIM_CAT(gtest_shim_regp_, __LINE__) = [] {                                              \
    ::testing::ParamSuite<suite>::EnsureRegistered(#suite);                                                 \
    ::testing::ParamSuite<suite>::Bodies().push_back(                                                       \
        {#name, [](const std::string& full) { ::testing::RunOne<GTEST_SHIM_CLASS(suite, name)>(full); }});  \
    return 0;                                                                                               \
  }();                                                                                                      \
  void GTEST_SHIM_CLASS(suite, name)::TestBody()

#define INSTANTIATE_TEST_SUITE_P(prefix, suite, ...)                                                        \
  static int GTEST_SHIM_CAT(gtest_shim_inst_, __LINE__) = [] {                                              \
    ::testing::ParamSuite<suite>::EnsureRegistered(#suite);                                                 \
    ::testing::ParamSuite<suite>::Instances().push_back(                                                    \
        {#prefix, static_cast<std::vector<suite::ParamType>>(GTEST_SHIM_FIRST_(__VA_ARGS__, 0))});          \
    return 0;                                                                                               \
  }()
#define GTEST_SHIM_FIRST_(first, ...) first
#define GTEST_ALLOW_UNINSTANTIATED_PARAMETERIZED_TEST(suite) static_assert(true, "")

#define GTEST_SHIM_AMBIGUOUS_ELSE_ switch (0) case 0: default:
#define GTEST_SHIM_CHECK_(cond, text, fatal_return)                                                          \
  GTEST_SHIM_AMBIGUOUS_ELSE_ if (cond) ; else fatal_return ::testing::AssertHelper(__FILE__, __LINE__, text, true) = ::testing::Message()

#define GTEST_SHIM_CMP_(a, b, op, name, fatal_return)                                                        \
  GTEST_SHIM_AMBIGUOUS_ELSE_ if ([&] { return (a)op(b); }()) ; else fatal_return ::testing::AssertHelper(     \
      __FILE__, __LINE__,                                                                                    \
      std::string("Expected: (" #a ") " #op " (" #b "), actual: ") + ::testing::Print(a) + " vs " + ::testing::Print(b), true) = ::testing::Message()
#define GTEST_SHIM_NEAR_(a, b, tol, fatal_return)                                                            \
  GTEST_SHIM_AMBIGUOUS_ELSE_ if ([&] { return std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol); }()) ; else fatal_return ::testing::AssertHelper( \
      __FILE__, __LINE__,                                                                                    \
      std::string("The difference between " #a " and " #b " exceeds " #tol ": ") + ::testing::Print(a) + " vs " + ::testing::Print(b), true) = ::testing::Message()

#define EXPECT_TRUE(c) GTEST_SHIM_CHECK_(static_cast<bool>(c), "Value of: " #c "\n  Actual: false\nExpected: true", )
#define EXPECT_FALSE(c) GTEST_SHIM_CHECK_(!static_cast<bool>(c), "Value of: " #c "\n  Actual: true\nExpected: false", )
#define ASSERT_TRUE(c) GTEST_SHIM_CHECK_(static_cast<bool>(c), "Value of: " #c "\n  Actual: false\nExpected: true", return)
#define ASSERT_FALSE(c) GTEST_SHIM_CHECK_(!static_cast<bool>(c), "Value of: " #c "\n  Actual: true\nExpected: false", return)
#define EXPECT_EQ(a, b) GTEST_SHIM_CMP_(a, b, ==, EQ, )
#define EXPECT_NE(a, b) GTEST_SHIM_CMP_(a, b, !=, NE, )
#define EXPECT_LT(a, b) GTEST_SHIM_CMP_(a, b, <, LT, )
#define EXPECT_LE(a, b) GTEST_SHIM_CMP_(a, b, <=, LE, )
#define EXPECT_GT(a, b) GTEST_SHIM_CMP_(a, b, >, GT, )
#define EXPECT_GE(a, b) GTEST_SHIM_CMP_(a, b, >=, GE, )
#define ASSERT_EQ(a, b) GTEST_SHIM_CMP_(a, b, ==, EQ, return)
#define ASSERT_NE(a, b) GTEST_SHIM_CMP_(a, b, !=, NE, return)
#define ASSERT_LT(a, b) GTEST_SHIM_CMP_(a, b, <, LT, return)
#define ASSERT_LE(a, b) GTEST_SHIM_CMP_(a, b, <=, LE, return)
#define ASSERT_GT(a, b) GTEST_SHIM_CMP_(a, b, >, GT, return)
#define ASSERT_GE(a, b) GTEST_SHIM_CMP_(a, b, >=, GE, return)
#define EXPECT_NEAR(a, b, tol) GTEST_SHIM_NEAR_(a, b, tol, )
#define ASSERT_NEAR(a, b, tol) GTEST_SHIM_NEAR_(a, b, tol, return)
#define EXPECT_FLOAT_EQ(a, b) GTEST_SHIM_NEAR_(a, b, 4 * 1.1920929e-7 * std::fabs(static_cast<double>(b)), )
#define EXPECT_DOUBLE_EQ(a, b) GTEST_SHIM_NEAR_(a, b, 4 * 2.220446e-16 * std::fabs(static_cast<double>(b)), )
#define FAIL() return ::testing::AssertHelper(__FILE__, __LINE__, "Failed", true) = ::testing::Message()
#define ADD_FAILURE() ::testing::AssertHelper(__FILE__, __LINE__, "Failed", false) = ::testing::Message()
#define SUCCEED() ::testing::Message()
#define GTEST_SKIP() return ::testing::SkipHelper() = ::testing::Message()

#endif  // ORACLE_GTEST_SHIM_GTEST_H_
