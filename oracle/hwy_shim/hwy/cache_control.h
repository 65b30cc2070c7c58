// TEST INFRASTRUCTURE ONLY: stand-in for <hwy/cache_control.h> (see base.h).
#ifndef ORACLE_HWY_SHIM_CACHE_CONTROL_H_
#define ORACLE_HWY_SHIM_CACHE_CONTROL_H_
#include "hwy/base.h"
namespace hwy {
template <typename T>
HWY_API void Prefetch(const T* p) {
  __builtin_prefetch(p, 0, 3);
}
HWY_API void FlushStream() {}
HWY_API void Pause() {}
}  // namespace hwy
#endif
