// tools/chainprof.hip -- diagnostic build of the compressor with per-phase cycle stamps.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_PROF -I../include -I../compression-accelerator_amd/csrc chainprof.hip -o chainprof
// usage: chainprof <raw input file> [n_blocks]   (blocks = consecutive 64 KiB windows, tiled)
#include "../compression-accelerator_amd/csrc/snappy_compress.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv){
  FILE* f=fopen(argv[1],"rb"); if(!f){perror(argv[1]);return 1;} fseek(f,0,SEEK_END); size_t L=ftell(f); fseek(f,0,SEEK_SET);
  std::vector<uint8_t> h(L); if(fread(h.data(),1,L,f)!=L) return 1; fclose(f);
  uint32_t nb = argc>2 ? atoi(argv[2]) : 64;
  std::vector<uint8_t> in((size_t)nb*65536); for(size_t i=0;i<in.size();i++) in[i]=h[i%L];
  uint8_t *d_in,*d_out; uint32_t* d_len; hipMalloc(&d_in,in.size()); hipMalloc(&d_out,(size_t)nb*76544); hipMalloc(&d_len,nb*4);
  hipMemcpy(d_in,in.data(),in.size(),hipMemcpyHostToDevice);
  const char* names[16]={"dense: window load+hash","dense: LDS lookup/readback/cut","dense: cand loads+compare","dense: walk","dense: commit+emit","dense: long match","strided: batch","strided: hit+emit","","","STAT lim (avg)","STAT cut<64 (fraction)","STAT path hits (avg)","STAT advance (avg)","STAT all hits in window (avg)",""};
  for(int rep=0;rep<2;rep++){
    unsigned long long z[32]={0}; hipMemcpyToSymbol(HIP_SYMBOL(szk::g_prof), z, sizeof z);
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    szk::launch_compress_blocks(d_in,65536,nullptr,nb,d_out,76544,d_len,1,nullptr,0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
    unsigned long long p[32]; hipMemcpyFromSymbol(p, HIP_SYMBOL(szk::g_prof), sizeof p);
    if(rep==1){ printf("nb=%u kernel %.3f ms\n", nb, ms); unsigned long long tot=0; for(int i=0;i<16;i++) tot+=p[i];
      for(int i=0;i<16;i++) if(p[16+i]) printf("%-30s total %12llu cyc  count %9llu  avg %7.1f cyc  (%4.1f%%)  per block %.0f\n", names[i], p[i], p[16+i], (double)p[i]/p[16+i], 100.0*p[i]/tot, (double)p[i]/nb);
      printf("sum %.0f cycles per block (memtime ticks = 100MHz? see below)\n",(double)tot/nb); }
  }
  return 0;
}
