#include <hip/hip_runtime.h>
__global__ void k(float* p){ p[threadIdx.x]*=2.f; }
extern "C" void run(float* p){ hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p); }
