// TEST INFRASTRUCTURE ONLY (part of oracle/): stand-in for <hwy/per_target.h>, see base.h.
#ifndef ORACLE_HWY_SHIM_PER_TARGET_H_
#define ORACLE_HWY_SHIM_PER_TARGET_H_

#include <stdint.h>

#include "hwy/highway.h"

namespace hwy {
static inline int64_t DispatchedTarget() { return HWY_SCALAR; }
}  // namespace hwy

#endif  // ORACLE_HWY_SHIM_PER_TARGET_H_
