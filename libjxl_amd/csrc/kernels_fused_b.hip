// kernels_fused_b.hip -- second translation unit of the fused kernel (see the note above FusedSupported in
// kernels_fused.hip): the EPF1 stage lists, compiled in parallel with the rest.
#define JXLHIP_FUSED_PART 1
#include "kernels_fused.hip"
