// tools/copybench.hip -- microbenchmark: what does a wave-per-64KiB-block copy reach on MI355X?
// Used to calibrate the roofline of the C3 (incompressible) path of the compressor.
// Build: hipcc --offload-arch=gfx950 -O3 -o copybench copybench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

__device__ __forceinline__ u32x4 ld128(const uint8_t* p){ u32x4 v; __builtin_memcpy(&v,p,16); return v; }
__device__ __forceinline__ void st128(uint8_t* p, u32x4 v){ __builtin_memcpy(p,&v,16); }

// grid-stride plain copy, 256 threads
__global__ __launch_bounds__(256) void k_plain(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t nvec){
  size_t i = blockIdx.x*(size_t)256 + threadIdx.x; size_t stride = (size_t)gridDim.x*256;
  for (; i + 3*stride < nvec; i += 4*stride){ u32x4 a=s[i],b=s[i+stride],c=s[i+2*stride],e=s[i+3*stride]; d[i]=a;d[i+stride]=b;d[i+2*stride]=c;d[i+3*stride]=e; }
  for (; i < nvec; i += stride) d[i]=s[i];
}

// wave per 64 KiB block; MODE 0: src+soff unaligned loads, dst+doff: aligned-by-construction stores if doff==0
template<int U, int LDSKB>
__global__ __launch_bounds__(64) void k_wave(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t soff, uint32_t doff, uint64_t dstride){
  __shared__ uint8_t pad[LDSKB*1024 ? LDSKB*1024 : 4];
  if (LDSKB && threadIdx.x==9999) pad[soff]=1;
  const uint32_t lane = threadIdx.x;
  const uint8_t* s = src + (size_t)blockIdx.x*65536 + soff;
  uint8_t* d = dst + (size_t)blockIdx.x*dstride + doff;
  const uint32_t nvec = (65536 - 64) / 16;   // keep in-bounds with offsets
  uint32_t i = lane;
  for (; i + (U-1)*64 < nvec; i += U*64){
    u32x4 q[U];
    #pragma unroll
    for (int u=0;u<U;u++) q[u]=ld128(s+16*(size_t)(i+64*u));
    #pragma unroll
    for (int u=0;u<U;u++) st128(d+16*(size_t)(i+64*u), q[u]);
  }
  for (; i < nvec; i += 64) st128(d+16*(size_t)i, ld128(s+16*(size_t)i));
  if (LDSKB && threadIdx.x==9999) dst[0]=pad[doff];
}

template<int U, int NTL, int NTS>
__global__ __launch_bounds__(64) void k_wave_nt(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t dstride){
  __shared__ uint8_t pad[32768];
  if (threadIdx.x==9999) pad[dstride&1023]=1;
  const uint32_t lane = threadIdx.x;
  const u32x4* s = (const u32x4*)(src + (size_t)blockIdx.x*65536);
  u32x4* d = (u32x4*)(dst + (size_t)blockIdx.x*dstride);
  const uint32_t nvec = 4096;
  for (uint32_t i = lane; i < nvec; i += U*64){
    u32x4 q[U];
    #pragma unroll
    for (int u=0;u<U;u++) q[u]= NTL ? __builtin_nontemporal_load(s+i+64*u) : s[i+64*u];
    #pragma unroll
    for (int u=0;u<U;u++) { if (NTS) __builtin_nontemporal_store(q[u], d+i+64*u); else d[i+64*u]=q[u]; }
  }
  if (threadIdx.x==9999) dst[0]=pad[5];
}
template<int U>
__global__ __launch_bounds__(64) void k_wave_ro(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst){
  __shared__ uint8_t pad[32768];
  if (threadIdx.x==9999) pad[5]=1;
  const uint32_t lane = threadIdx.x;
  const u32x4* s = (const u32x4*)(src + (size_t)blockIdx.x*65536);
  u32x4 acc = {0,0,0,0};
  for (uint32_t i = lane; i < 4096; i += U*64){
    #pragma unroll
    for (int u=0;u<U;u++) { u32x4 q = s[i+64*u]; acc.x^=q.x; acc.y^=q.y; acc.z^=q.z; acc.w^=q.w; }
  }
  if ((acc.x^acc.y^acc.z^acc.w)==0x12345678u) dst[lane]=pad[5];
}
template<int U>
__global__ __launch_bounds__(64) void k_wave_wo(uint8_t* __restrict__ dst, uint64_t dstride){
  __shared__ uint8_t pad[32768];
  if (threadIdx.x==9999) pad[5]=1;
  const uint32_t lane = threadIdx.x;
  u32x4* d = (u32x4*)(dst + (size_t)blockIdx.x*dstride);
  u32x4 v = {lane,1,2,3};
  for (uint32_t i = lane; i < 4096; i += 64) d[i]=v;
  if (threadIdx.x==9999) dst[0]=pad[5];
}

// aligned loads, realign across lanes by `delta` bytes (0..15), aligned stores.
// out chunk j = src bytes [16j+delta, 16j+delta+16)
template<int U, int LDSKB>
__global__ __launch_bounds__(64) void k_shuf(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t delta, uint64_t dstride){
  __shared__ uint8_t pad[LDSKB*1024 ? LDSKB*1024 : 4];
  if (LDSKB && threadIdx.x==9999) pad[delta]=1;
  const uint32_t lane = threadIdx.x;
  const u32x4* s = (const u32x4*)(src + (size_t)blockIdx.x*65536);
  u32x4* d = (u32x4*)(dst + (size_t)blockIdx.x*dstride);
  const uint32_t nvec = 65536/16 - 64;
  const uint32_t dw = delta >> 2, db = delta & 3;
  for (uint32_t base = 0; base + U*64 <= nvec; base += U*64){
    u32x4 q[U+1];
    #pragma unroll
    for (int u=0;u<U;u++) q[u]=s[base+64*u+lane];
    // extra: first chunk of the next row for lane 63's neighbour
    q[U] = s[base+64*U+ (lane==63?0:lane)];   // only lane 63's neighbour (next row lane 0) is needed; cheap load (same line)
    #pragma unroll
    for (int u=0;u<U;u++){
      // neighbour chunk = lane+1 of same row, or lane 0 of next row
      u32x4 nb;
      nb.x = __shfl_down(q[u].x,1); nb.y=__shfl_down(q[u].y,1); nb.z=__shfl_down(q[u].z,1); nb.w=__shfl_down(q[u].w,1);
      u32x4 nx; nx.x=__shfl(q[u+1].x,0); nx.y=__shfl(q[u+1].y,0); nx.z=__shfl(q[u+1].z,0); nx.w=__shfl(q[u+1].w,0);
      if (lane==63) nb = nx;
      uint32_t w[8] = {q[u].x,q[u].y,q[u].z,q[u].w,nb.x,nb.y,nb.z,nb.w};
      u32x4 o;
      // select dwords dw..dw+4 then byte-align
      uint32_t a0,a1,a2,a3,a4;
      switch(dw){ case 0:a0=w[0];a1=w[1];a2=w[2];a3=w[3];a4=w[4];break; case 1:a0=w[1];a1=w[2];a2=w[3];a3=w[4];a4=w[5];break;
                  case 2:a0=w[2];a1=w[3];a2=w[4];a3=w[5];a4=w[6];break; default:a0=w[3];a1=w[4];a2=w[5];a3=w[6];a4=w[7];break; }
      o.x=__builtin_amdgcn_alignbyte(a1,a0,db); o.y=__builtin_amdgcn_alignbyte(a2,a1,db); o.z=__builtin_amdgcn_alignbyte(a3,a2,db); o.w=__builtin_amdgcn_alignbyte(a4,a3,db);
      d[base+64*u+lane]=o;
    }
  }
}

// contiguous tile per workgroup (the shape of a generic elementwise kernel): TPB threads, U x 16 B each
template<int TPB, int U, int NT>
__global__ __launch_bounds__(TPB) void k_tile(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t dstride){
  const size_t tile = (size_t)TPB * U * 16;
  const size_t off = (size_t)blockIdx.x * tile;          // byte offset in the input
  const size_t blk = off >> 16, inb = off & 65535;       // 64 KiB block and offset inside it (tile <= 64 KiB, divides it)
  const u32x4* s = (const u32x4*)(src + off);
  u32x4* d = (u32x4*)(dst + blk * dstride + inb);
  u32x4 q[U];
  #pragma unroll
  for (int u=0;u<U;u++) q[u] = NT ? __builtin_nontemporal_load(s + threadIdx.x + TPB*u) : s[threadIdx.x + TPB*u];
  #pragma unroll
  for (int u=0;u<U;u++) { if (NT) __builtin_nontemporal_store(q[u], d + threadIdx.x + TPB*u); else d[threadIdx.x + TPB*u] = q[u]; }
}
// wave per 64 KiB block without the LDS reservation (occupancy not limited to 4 waves per CU)
template<int U, int NT>
__global__ __launch_bounds__(64) void k_wave_nolds(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t dstride){
  const uint32_t lane = threadIdx.x;
  const u32x4* s = (const u32x4*)(src + (size_t)blockIdx.x*65536);
  u32x4* d = (u32x4*)(dst + (size_t)blockIdx.x*dstride);
  for (uint32_t i = lane; i < 4096; i += U*64){
    u32x4 q[U];
    #pragma unroll
    for (int u=0;u<U;u++) q[u]= NT ? __builtin_nontemporal_load(s+i+64*u) : s[i+64*u];
    #pragma unroll
    for (int u=0;u<U;u++) { if (NT) __builtin_nontemporal_store(q[u], d+i+64*u); else d[i+64*u]=q[u]; }
  }
}

// wave per 64 KiB block, whole block in flight (64 x 16 B per lane), chunk order rotated per block
template<int ROT, int NT>
__global__ __launch_bounds__(64) void k_wave_all(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t dstride){
  __shared__ uint8_t pad[32768];
  if (threadIdx.x==9999) pad[dstride&1023]=1;
  const uint32_t lane = threadIdx.x;
  const u32x4* s = (const u32x4*)(src + (size_t)blockIdx.x*65536);
  u32x4* d = (u32x4*)(dst + (size_t)blockIdx.x*dstride);
  const uint32_t rot = ROT ? (blockIdx.x * 7u) & 63u : 0u;
  u32x4 q[64];
  #pragma unroll
  for (int u=0;u<64;u++) { const uint32_t c = (u + rot) & 63u; q[u] = NT ? __builtin_nontemporal_load(s+lane+64*c) : s[lane+64*c]; }
  #pragma unroll
  for (int u=0;u<64;u++) { const uint32_t c = (u + rot) & 63u; if (NT) __builtin_nontemporal_store(q[u], d+lane+64*c); else d[lane+64*c]=q[u]; }
  if (threadIdx.x==9999) dst[0]=pad[5];
}

int main(int argc, char** argv){
  size_t nblk = argc>1 ? atol(argv[1]) : 32768;
  size_t dstride = 76544;
  uint8_t *src,*dst; CK(hipMalloc(&src, nblk*65536+4096)); CK(hipMalloc(&dst, nblk*dstride+4096));
  CK(hipMemset(src, 1, nblk*65536+4096)); CK(hipMemset(dst, 0, nblk*dstride+4096));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit=[&](const char* name, auto launch){
    for(int i=0;i<3;i++) launch(); CK(hipDeviceSynchronize());
    float best=1e9, sum=0; int R=10;
    for(int r=0;r<R;r++){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); best=ms<best?ms:best; sum+=ms; }
    double bytes = (double)nblk*65536;
    printf("%-44s avg %.3f ms  best %.3f ms  -> input %.0f GB/s, in+out %.0f GB/s (best)\n", name, sum/R, best, bytes/best/1e6, 2*bytes/best/1e6);
  };
  size_t nvec = nblk*65536/16;
  timeit("plain grid-stride 256thr x 2048 blocks", [&]{ hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, nvec); });
  timeit("plain grid-stride 256thr x 8192 blocks", [&]{ hipLaunchKernelGGL(k_plain, dim3(8192), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, nvec); });
#define W(U,L,so,dof,name) timeit(name, [&]{ hipLaunchKernelGGL((k_wave<U,L>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint32_t)so, (uint32_t)dof, (uint64_t)dstride); });
  W(4,0,0,0,  "wave/blk U=4  lds0  aligned");
  W(4,32,0,0, "wave/blk U=4  lds32 aligned");
  W(8,32,0,0, "wave/blk U=8  lds32 aligned");
  W(16,32,0,0,"wave/blk U=16 lds32 aligned");
  W(32,32,0,0,"wave/blk U=32 lds32 aligned");
  W(16,24,0,0,"wave/blk U=16 lds24 aligned (6/CU)");
  W(16,16,0,0,"wave/blk U=16 lds16 aligned (10/CU)");
  W(4,32,10,0, "wave/blk U=4  lds32 src+10 (unaligned ld)");
  W(16,32,10,0,"wave/blk U=16 lds32 src+10 (unaligned ld)");
  W(16,32,8,0, "wave/blk U=16 lds32 src+8  (8B-aligned ld)");
  W(16,32,4,0, "wave/blk U=16 lds32 src+4  (4B-aligned ld)");
  W(16,32,0,6, "wave/blk U=16 lds32 dst+6 (unaligned st)");
  W(16,32,0,8, "wave/blk U=16 lds32 dst+8 (8B-aligned st)");
#define N(U,a,b,name) timeit(name, [&]{ hipLaunchKernelGGL((k_wave_nt<U,a,b>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  N(4,0,0,"nt: U=4 plain/plain");
  N(4,1,0,"nt: U=4 nt-load/plain");
  N(4,0,1,"nt: U=4 plain/nt-store");
  N(4,1,1,"nt: U=4 nt-load/nt-store");
  N(8,1,1,"nt: U=8 nt-load/nt-store");
  N(2,1,1,"nt: U=2 nt-load/nt-store");
  timeit("read-only U=4 (input GB/s = read BW)", [&]{ hipLaunchKernelGGL((k_wave_ro<4>), dim3(nblk), dim3(64), 0, 0, src, dst); });
  timeit("read-only U=8", [&]{ hipLaunchKernelGGL((k_wave_ro<8>), dim3(nblk), dim3(64), 0, 0, src, dst); });
  timeit("write-only (input GB/s = write BW)", [&]{ hipLaunchKernelGGL((k_wave_wo<4>), dim3(nblk), dim3(64), 0, 0, dst, (uint64_t)dstride); });
  { size_t ds2=65536; timeit("nt U=4 nt/nt dst stride 65536", [&]{ hipLaunchKernelGGL((k_wave_nt<4,1,1>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)ds2); }); }
#define S(U,L,dl,name) timeit(name, [&]{ hipLaunchKernelGGL((k_shuf<U,L>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint32_t)dl, (uint64_t)dstride); });
  S(8,32,10, "wave/blk shuffle-realign U=8  delta=10");
  S(16,32,10,"wave/blk shuffle-realign U=16 delta=10");
#define T(TPB,U,NT,name) { size_t tile=(size_t)TPB*U*16; timeit(name, [&]{ hipLaunchKernelGGL((k_tile<TPB,U,NT>), dim3(nblk*65536/tile), dim3(TPB), 0, 0, src, dst, (uint64_t)dstride); }); }
  T(256,1,0,"tile 4 KiB/WG 256thr plain");
  T(256,4,0,"tile 16 KiB/WG 256thr plain");
  T(256,4,1,"tile 16 KiB/WG 256thr nt");
  T(256,16,1,"tile 64 KiB/WG 256thr nt");
  T(512,4,0,"tile 32 KiB/WG 512thr plain");
  T(1024,4,0,"tile 64 KiB/WG 1024thr plain");
  T(1024,4,1,"tile 64 KiB/WG 1024thr nt");
  T(64,4,0,"tile 4 KiB/WG 64thr plain");
  T(64,4,1,"tile 4 KiB/WG 64thr nt");
  timeit("wave/blk no LDS U=4 plain", [&]{ hipLaunchKernelGGL((k_wave_nolds<4,0>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  timeit("wave/blk no LDS U=4 nt", [&]{ hipLaunchKernelGGL((k_wave_nolds<4,1>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  timeit("wave/blk no LDS U=16 nt", [&]{ hipLaunchKernelGGL((k_wave_nolds<16,1>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  timeit("wave/blk whole block in regs nt", [&]{ hipLaunchKernelGGL((k_wave_all<0,1>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  timeit("wave/blk whole block in regs nt, rotated", [&]{ hipLaunchKernelGGL((k_wave_all<1,1>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  timeit("wave/blk whole block in regs plain, rotated", [&]{ hipLaunchKernelGGL((k_wave_all<1,0>), dim3(nblk), dim3(64), 0, 0, src, dst, (uint64_t)dstride); });
  return 0;
}
