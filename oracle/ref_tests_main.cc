// TEST INFRASTRUCTURE ONLY (oracle/build_ref_tests.py): main() of the reference's own unit tests built against
// oracle/gtest_shim, and the two helpers of lib/jxl/test_utils.cc they need (that file itself pulls in the encoder
// front-end and lib/extras codecs).
#include <stdio.h>
#include <stdlib.h>

#include "gtest/gtest.h"

struct JxlCmsInterface;

namespace jxl {
namespace test {
// lib/jxl/test_utils.h:45,67: Check(OK) aborts the test binary
void CheckImpl(bool ok, const char* condition, const char* file, int line) {
  if (!ok) {
    fprintf(stderr, "Check(%s) failed at %s:%d\n", condition, file, line);
    abort();
  }
}
}  // namespace test
}  // namespace jxl

// lib/include/jxl/cms.h: opsin_inverse_test.cc passes the default CMS to ToXYB for an image that is already linear
// sRGB -- ToXYB does not call into it (enc_xyb.cc: no transform when the encoding matches); a CMS library is not
// linked into the checker
extern "C" const JxlCmsInterface* JxlGetDefaultCms() {
  static const long long dummy[16] = {0};
  return reinterpret_cast<const JxlCmsInterface*>(dummy);
}

int main(int argc, char** argv) {
  ::testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
