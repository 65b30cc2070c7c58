// TEST INFRASTRUCTURE ONLY: stand-in for <hwy/aligned_allocator.h>.
#ifndef ORACLE_HWY_SHIM_ALIGNED_ALLOCATOR_H_
#define ORACLE_HWY_SHIM_ALIGNED_ALLOCATOR_H_
#include <stdlib.h>

#include <memory>

#include "hwy/base.h"
namespace hwy {
struct AlignedFreer {
  template <typename T>
  void operator()(T* p) const {
    free(const_cast<void*>(static_cast<const void*>(p)));
  }
};
template <typename T>
using AlignedFreeUniquePtr = std::unique_ptr<T, AlignedFreer>;
template <typename T>
AlignedFreeUniquePtr<T[]> AllocateAligned(size_t items) {
  void* p = nullptr;
  if (posix_memalign(&p, HWY_ALIGNMENT, items * sizeof(T) + HWY_ALIGNMENT)) p = nullptr;
  return AlignedFreeUniquePtr<T[]>(static_cast<T*>(p), AlignedFreer());
}
}  // namespace hwy
#endif
