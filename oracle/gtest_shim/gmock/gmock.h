// TEST INFRASTRUCTURE ONLY: lib/jxl/test_utils.h includes gmock for matchers the hot-path tests never use.
#ifndef ORACLE_GTEST_SHIM_GMOCK_H_
#define ORACLE_GTEST_SHIM_GMOCK_H_
#include "gtest/gtest.h"
#endif
