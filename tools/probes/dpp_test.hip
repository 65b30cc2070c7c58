// Probe: semantics of the wave-wide DPP shifts on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  int v = lane + 100;
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + lane] = __shfl_up(v, 1, 64);
  out[192 + lane] = __shfl_down(v, 1, 64);
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; r++) { printf("row %d:", r); for (int i : {0,1,2,15,16,17,31,32,33,62,63}) printf(" [%d]=%d", i, h[r*64+i]); printf("\n"); }
  return 0;
}
