// tools/isacost.hip -- cycles per instruction of short dependent sequences, one wave on a CU (s_memtime around
// 64 x the pattern): what the serial part of a team's turn (snappy_compress.hip) pays per instruction.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 isacost.hip -o isacost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void k(unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[256];
    const uint32_t lane = threadIdx.x;
    lds[lane] = (lane * 7 + 1) & 63;
    lds[lane + 64] = (lane * 5 + 3) & 63;
    __syncthreads();
    unsigned long long t0, t1;
    uint32_t s = seed, s2 = seed + 1, s3 = seed + 2, s4 = seed + 3;
    uint32_t v = lane ^ seed, v2 = (lane * 3 + 1) & 63;
    uint64_t m = 0x123456789abcdefull + seed, m2 = 1;
    int idx = 0;
#define MEASURE(...)                                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    __VA_ARGS__                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if (lane == 0) out[blockIdx.x * 32 + idx] = t1 - t0;                                          \
    idx++;
    // 0: empty
    MEASURE()
    // 1: 64 dependent s_add_u32
    MEASURE(asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s));)
    // 2: 64 independent s_add_u32 (4 chains)
    MEASURE(asm volatile(REP16("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") : "+s"(s), "+s"(s2), "+s"(s3), "+s"(s4));)
    // 3: 64 dependent v_add_u32
    MEASURE(asm volatile(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(v));)
    // 4: walk step x16: s_lshl_b64, s_or_b64, s_nop 1, v_readlane (64 instrs)
    {
        uint32_t x = seed & 63;
        uint64_t S = 0, bit;
        MEASURE(asm volatile(REP16("s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t") : "+s"(S), "+s"(x), "=&s"(bit) : "v"(v2));)
        m ^= S + x;
    }
    // 5: v_readlane -> s_and -> v_readlane (32 pairs = 64 instrs + nops)
    {
        uint32_t x = seed & 63;
        MEASURE(asm volatile(REP16("v_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\ts_nop 3\n\tv_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\ts_nop 3\n\t") : "+s"(x) : "v"(v2));)
        m ^= x;
    }
    // 6: v_cmp -> s_and_b64 -> v_cndmask (VALU/SALU ping-pong) x 21 = 63 instrs
    MEASURE(asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, %2\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(v) : "v"(v2), "s"(m) : "vcc");)
    // 7: 16 dependent ds_read_b32 (address from the value read)
    {
        uint32_t a = (lane & 63) * 4;
        MEASURE(asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 %0, 2, %0\n\t") : "+v"(a) :: "memory");)
        v ^= a;
    }
    // 8: 16 dependent ds_bpermute
    {
        uint32_t a = (lane * 4) & 255, d = v2 * 4;
        MEASURE(asm volatile(REP16("ds_bpermute_b32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) : "v"(d) : "memory");)
        v ^= a;
    }
    // 9: 64 dependent s_lshl_b64 / s_or_b64 / s_sub_u32 / s_subb_u32 mix on 64-bit masks
    MEASURE(asm volatile(REP16("s_lshl_b64 %0, %0, 1\n\ts_or_b64 %0, %0, %1\n\ts_andn2_b64 %0, %0, %1\n\ts_xor_b64 %0, %0, %1\n\t") : "+s"(m) : "s"(m2));)
    // 10: 16 taken branches (loop of 1 s_add + branch)
    {
        uint32_t c = 16;
        MEASURE(asm volatile("1:\n\ts_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\t" : "+s"(c));)
        s ^= c;
    }
    // 11: ballot -> scalar use -> inverse (v_cmp; s_ff1; v_readlane) x 16
    {
        uint32_t x = 0;
        MEASURE(asm volatile(REP16("v_cmp_ne_u32 vcc, %1, %2\n\ts_ff1_i32_b64 %0, vcc\n\ts_nop 3\n\tv_readlane_b32 %0, %1, %0\n\t") : "+s"(x) : "v"(v), "v"(v2) : "vcc");)
        s ^= x;
    }
    // 12: ds_write_b16 + ds_read_u16 read-back x 8
    {
        uint32_t a = (lane & 63) * 2, d = lane;
        MEASURE(asm volatile(REP4(REP4("ds_write_b16 %0, %1\n\tds_read_u16 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t")) : "+v"(a), "+v"(d) :: "memory");)
        v ^= d;
    }
    // 13: s_waitcnt with nothing outstanding x 64
    MEASURE(asm volatile(REP64("s_waitcnt lgkmcnt(0)\n\t") ::: "memory");)
    // 14: v_readfirstlane + s_ use x 32
    {
        uint32_t x = 0;
        MEASURE(asm volatile(REP16("v_readfirstlane_b32 %0, %1\n\ts_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, %0\n\tv_readfirstlane_b32 %0, %1\n\t") : "+s"(x), "+v"(v));)
        s ^= x;
    }
    if (s + s2 + s3 + s4 + v + (uint32_t)m == 0x12345) out[0] = 1;
}
int main()
{
    unsigned long long *d, h[32 * 8];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, d, 5u);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[15] = {"empty (stamp overhead)", "64 dependent s_add_u32", "64 s_add_u32, 4 independent chains", "64 dependent v_add_u32",
                             "16 walk steps (s_lshl_b64, s_or_b64, s_nop 1, v_readlane)", "32 x (v_readlane, s_and, s_nop 3)", "16 x (v_cmp, s_and_b64 vcc, v_cndmask)",
                             "16 dependent ds_read_b32 (+wait, +v_lshl)", "16 dependent ds_bpermute (+wait)", "64 dependent 64-bit SALU ops", "16 taken branches (3-instr loop)",
                             "16 x (v_cmp, s_ff1, s_nop 3, v_readlane)", "16 x (ds_write_b16, ds_read_u16, wait)", "64 s_waitcnt (nothing outstanding)", "16 x (readfirstlane, s_add, v_add, readfirstlane)"};
    const int ninstr[15] = {0, 64, 64, 64, 64, 96, 48, 48, 32, 64, 48, 64, 48, 64, 64};
    for (int i = 0; i < 15; i++) {
        unsigned long long best = ~0ull;
        for (int b = 0; b < 8; b++) best = h[b * 32 + i] < best ? h[b * 32 + i] : best;
        printf("%-62s %6llu ticks", names[i], best);
        if (i) printf("  (%.1f per instruction over the empty %llu)", (double)(best - h[0]) / ninstr[i], h[0]);
        printf("\n");
    }
    return 0;
}
