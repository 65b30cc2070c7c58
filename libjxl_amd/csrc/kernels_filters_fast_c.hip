// kernels_filters_fast_c.hip -- part 2 of kernels_filters_fast.hip: one more set of packed output formats fixed at
// compile time (see JXLHIP_FIXED_FORMATS_2 there), compiled in parallel with the rest.
#define JXLHIP_FAST_PART 2
#include "kernels_filters_fast.hip"
