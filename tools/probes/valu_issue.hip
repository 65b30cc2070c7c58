// valu_issue.hip -- how many cycles does a SIMD of gfx950 take per wave64 VALU instruction?
//
// DESIGN.md's "k_fused_pc is 59 % VALU-busy" assumed 4 cycles per wave64 instruction (a 16-lane SIMD, CDNA3);
// /opt/skills/guides/MI355X_MICROARCH.md gives 2 (SIMD-32) for v_fma_f32.  This probe measures it: every wave runs
// ITER x 64 instructions of one kind, as 8 independent chains (throughput) or as one dependent chain (latency), at
// 1 .. 4 waves per SIMD (blocks of 256 threads = one wave per SIMD, WPS blocks per CU, 256 CUs).  Reported:
//   cyc/instr/SIMD = wave's s_memtime delta (shader clock) / (instructions per wave x waves per SIMD)
// and the same from the hipEvent time at the nominal 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/valu_issue.hip -o tools/probes/build/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Kind { FMA_IND, FMA_DEP, PKFMA_IND, PKFMA_DEP, PKADD_IND, ADD_DPP_IND, RCP_IND, MIX_MARCH, MAX3_IND, CVT_IND, KINDS };
static const char* kNames[KINDS] = {"v_fma_f32 8 chains", "v_fma_f32 1 chain", "v_pk_fma_f32 8 chains", "v_pk_fma_f32 1 chain",
                                    "v_pk_add_f32 8 chains", "v_add_f32_dpp 8 chains", "v_rcp_f32 8 chains",
                                    "mix: pk_fma,pk_add,add_dpp,fma (2:2:2:2)", "v_max3_f32 8 chains", "v_cvt_f32_i32 8 chains"};

template <int K>
__global__ __launch_bounds__(256) void k_probe(float* out, unsigned long long* cyc, int iters) {
  float a[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a[i] = (float)(threadIdx.x + i) * 1e-3f;
    p[i] = v2f{a[i], a[i] + 1.0f};
  }
  const float m = 0.999f, c = 1e-4f;
  const v2f pm = v2f{m, m}, pc = v2f{c, c};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if constexpr (K == FMA_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      } else if constexpr (K == FMA_DEP) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
      } else if constexpr (K == PKFMA_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
      } else if constexpr (K == PKFMA_DEP) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pm), "v"(pc));
      } else if constexpr (K == PKADD_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      } else if constexpr (K == ADD_DPP_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1" : "+v"(a[i]) : "v"(c));
      } else if constexpr (K == RCP_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      } else if constexpr (K == MIX_MARCH) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pm), "v"(pc));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[1]) : "v"(pc));
        asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1" : "+v"(a[2]) : "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(m), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[4]) : "v"(pm), "v"(pc));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[5]) : "v"(pc));
        asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1" : "+v"(a[6]) : "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[7]) : "v"(m), "v"(c));
      } else if constexpr (K == MAX3_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      } else if constexpr (K == CVT_IND) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K>
static void Run(float* out, unsigned long long* cyc, unsigned long long* hcyc, int cus) {
  const int iters = 4096;
  for (int wps = 1; wps <= 4; wps++) {
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    k_probe<K><<<blocks, 256>>>(out, cyc, 64);  // warm-up
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    double wave_cyc = 0;
    for (int rep = 0; rep < 5; rep++) {
      CHK(hipEventRecord(e0));
      k_probe<K><<<blocks, 256>>>(out, cyc, iters);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) {
        best = ms;
        CHK(hipMemcpy(hcyc, cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < blocks * 4; i++) s += (double)hcyc[i];
        wave_cyc = s / (blocks * 4);
      }
    }
    const double instr = (double)iters * 64.0;
    printf("%-44s waves/SIMD %d: %8.3f ms  wave-clock %10.0f ticks  -> %.2f ticks/instr/SIMD (s_memtime), %.2f cyc/instr/SIMD (events @2.4 GHz)\n",
           kNames[K], wps, best, wave_cyc, wave_cyc / (instr * wps), best * 1e-3 * 2.4e9 / (instr * wps));
  }
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz; s_memtime may tick at a fixed reference clock (compare with the event column)\n",
         prop.gcnArchName, cus, prop.clockRate);
  float* out;
  unsigned long long *cyc, *hcyc;
  CHK(hipMalloc(&out, sizeof(float) * cus * 4 * 256));
  CHK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 4 * 4));
  hcyc = (unsigned long long*)malloc(sizeof(unsigned long long) * cus * 4 * 4);
  Run<FMA_IND>(out, cyc, hcyc, cus);
  Run<FMA_DEP>(out, cyc, hcyc, cus);
  Run<PKFMA_IND>(out, cyc, hcyc, cus);
  Run<PKFMA_DEP>(out, cyc, hcyc, cus);
  Run<PKADD_IND>(out, cyc, hcyc, cus);
  Run<ADD_DPP_IND>(out, cyc, hcyc, cus);
  Run<RCP_IND>(out, cyc, hcyc, cus);
  Run<MAX3_IND>(out, cyc, hcyc, cus);
  Run<CVT_IND>(out, cyc, hcyc, cus);
  Run<MIX_MARCH>(out, cyc, hcyc, cus);
  return 0;
}
