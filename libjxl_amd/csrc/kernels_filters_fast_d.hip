// kernels_filters_fast_d.hip -- part 3 of kernels_filters_fast.hip: one more set of packed output formats fixed at
// compile time (see JXLHIP_FIXED_FORMATS_3 there), compiled in parallel with the rest.
#define JXLHIP_FAST_PART 3
#include "kernels_filters_fast.hip"
