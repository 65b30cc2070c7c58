// kernels_filters_fast_b.hip -- part 1 of kernels_filters_fast.hip: one more set of packed output formats fixed at
// compile time (see JXLHIP_FIXED_FORMATS_1 there), compiled in parallel with the rest.
#define JXLHIP_FAST_PART 1
#include "kernels_filters_fast.hip"
