// kernels_fused_pc.hip -- third translation unit of the fused kernel: its producer / consumer form (k_fused_pc,
// see the note in kernels_fused.hip).
#define JXLHIP_FUSED_PART 2
#include "kernels_fused.hip"
