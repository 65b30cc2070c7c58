// TEST INFRASTRUCTURE ONLY (part of oracle/): stand-in for <hwy/targets.h>, see base.h.  The one
// target of this shim is the single-lane scalar one; tools/codec_config.cc prints the list.
#ifndef ORACLE_HWY_SHIM_TARGETS_H_
#define ORACLE_HWY_SHIM_TARGETS_H_

#include <stdint.h>

#include <vector>

#include "hwy/highway.h"

namespace hwy {
static inline std::vector<int64_t> SupportedAndGeneratedTargets() { return {HWY_SCALAR}; }
}  // namespace hwy

#endif  // ORACLE_HWY_SHIM_TARGETS_H_
