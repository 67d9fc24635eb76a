// tools/occexp.hip -- what would a second wave per SIMD buy?  Compresses 8 KiB text blocks (table 8192 x u16
// = 16 KiB) with the product kernel built at 32 KiB LDS (4 blocks/CU) vs 16 KiB LDS (8 blocks/CU).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_EXPERIMENT_TABLE=<16384|8192> -DSZ_EXPERIMENT_WAVES=<1|2> -I../include -I../compression-accelerator_amd/csrc occexp.hip -o occexp_<x>
#include "../compression-accelerator_amd/csrc/snappy_kernels.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv){
  FILE* f=fopen(argv[1],"rb"); if(!f){perror(argv[1]);return 1;} fseek(f,0,SEEK_END); size_t L=ftell(f); fseek(f,0,SEEK_SET);
  std::vector<uint8_t> h(L); if(fread(h.data(),1,L,f)!=L) return 1; fclose(f);
  uint32_t nb = argc>2 ? atoi(argv[2]) : 65536; uint32_t bs = argc>3 ? atoi(argv[3]) : 8192;
  std::vector<uint8_t> in((size_t)nb*bs); for(size_t i=0;i<in.size();i++) in[i]=h[i%L];
  uint8_t *d_in,*d_out; uint32_t* d_len; hipMalloc(&d_in,in.size()); hipMalloc(&d_out,(size_t)nb*76544); hipMalloc(&d_len,nb*4);
  hipMemcpy(d_in,in.data(),in.size(),hipMemcpyHostToDevice);
  #ifdef SZ_EXPERIMENT_GLOBAL_TABLE
  uint16_t* tabs; hipMalloc(&tabs,(size_t)nb*SZ_EXPERIMENT_TABLE*2); hipMemcpyToSymbol(HIP_SYMBOL(szk::g_experiment_tables), &tabs, sizeof(tabs));
#endif
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); float best=1e9;
  for(int rep=0;rep<4;rep++){ hipEventRecord(e0); szk::launch_compress_blocks(d_in,bs,nullptr,nb,d_out,76544,d_len,1,0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  std::vector<uint32_t> len(nb); hipMemcpy(len.data(),d_len,nb*4,hipMemcpyDeviceToHost); unsigned long long tot=0; for(auto x:len) tot+=x;
  printf("table=%d waves_per_eu=%d bs=%u: %u blocks: best %.3f ms -> %.2f GB/s input (out bytes %llu)\n", SZ_EXPERIMENT_TABLE, SZ_EXPERIMENT_WAVES, bs, nb, best, (double)nb*bs/best/1e6, tot);
  return 0; }
