// kernels_fused_epf0.hip -- fourth translation unit of the fused kernel: k_fused_pc0, the producer / consumer form of
// [Gaborish] + EPF0 for epf_iters = 3 (see the note in kernels_fused.hip, part 3).
#define JXLHIP_FUSED_PART 3
#include "kernels_fused.hip"
