// tools/occ.hip -- how many 64-thread workgroups with X bytes of LDS fit on one CU (occupancy API + census)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template<int BYTES> __global__ __launch_bounds__(64) void k(uint32_t* out, int spin){
  __shared__ uint8_t lds[BYTES];
  lds[threadIdx.x] = threadIdx.x; __syncthreads();
  // census: record CU/XCC id, spin a while so all resident blocks overlap
  uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) { }
  if (threadIdx.x==0) { out[2*blockIdx.x] = hwid; out[2*blockIdx.x+1] = (xcc&0xf) | (lds[5]<<20); }
}
template<int BYTES> void run(){
  int nb=0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<BYTES>, 64, 0);
  printf("LDS %6d B: occupancy API says %d blocks/CU\n", BYTES, nb);
}
int main(){
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  printf("sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, CUs %d\n", p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.multiProcessorCount);
  run<32768>(); run<32704>(); run<32512>(); run<32256>(); run<31744>(); run<30720>(); run<28672>(); run<26624>(); run<16384>();
  // timing census: launch 1280 blocks that each spin ~2 ms; if 5/CU fit, total ~ 1 round
  uint32_t* d; hipMalloc(&d, 8*4096);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto t=[&](const char* nm, auto fn){ fn(); hipDeviceSynchronize(); hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("%s: %.3f ms\n", nm, ms); };
  t("1024 blocks lds32768 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<32768>, dim3(1024), dim3(64), 0, 0, d, 2000000); });
  t("1280 blocks lds32768 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<32768>, dim3(1280), dim3(64), 0, 0, d, 2000000); });
  t("1280 blocks lds32512 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<32512>, dim3(1280), dim3(64), 0, 0, d, 2000000); });
  t("1280 blocks lds32256 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<32256>, dim3(1280), dim3(64), 0, 0, d, 2000000); });
  t("1280 blocks lds30720 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<30720>, dim3(1280), dim3(64), 0, 0, d, 2000000); });
  t("1536 blocks lds26624 spin 2M cycles", [&]{ hipLaunchKernelGGL(k<26624>, dim3(1536), dim3(64), 0, 0, d, 2000000); });
  return 0;
}
