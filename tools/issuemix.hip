// tools/issuemix.hip -- how many instructions does one SIMD issue per 4-clock slot when several waves share it?
// One workgroup of 4*K waves per CU (K waves per SIMD: the dispatcher deals a workgroup's waves round the SIMDs);
// the k-th wave of a SIMD runs role[k]: a long dependent or independent stream of scalar or vector instructions,
// or the parse's usual ping-pong between the two units.  Prints clocks per instruction per wave and the SIMD's
// instructions per slot.   hipcc --offload-arch=gfx950 -O3 -std=c++17 issuemix.hip -o issuemix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
enum Role { S = 0, V = 1, S4 = 2, V4 = 3, M = 4, L = 5, IDLE = 6 };
static const char *role_name[] = {"S", "V", "S4", "V4", "M", "L", "-"};
// per iteration: 64 instructions of the role's kind (M: 16 x [v_readlane, s_and, s_lshl_b64, v_cndmask-ish] = 64)
__global__ __launch_bounds__(1024) void k(unsigned long long *out, const int *roles, int K, uint32_t iters, uint32_t seed)
{
    __shared__ uint32_t lds[1024];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds[threadIdx.x] = (threadIdx.x * 7 + 1) & 63;
    __syncthreads();
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int role = __builtin_amdgcn_readfirstlane(roles[wave >> 2]);
    seed = __builtin_amdgcn_readfirstlane(seed);
    uint32_t s = seed, s2 = seed + 1, s3 = seed + 2, s4 = seed + 3;
    uint32_t v = lane ^ seed, v2 = (lane * 3 + 1) & 63, v3 = lane + 5, v4 = lane + 9;
    uint64_t m = 0x123456789abcdefull + seed;
    uint32_t a = lane * 4;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define LOOP(...) for (uint32_t i = 0; i < iters; i++) { REP4(asm volatile(__VA_ARGS__);) }
    switch (role) {  // (256 instructions per trip of the loop: its own 3 instructions and taken branch are ~3 % of it)
    case S: LOOP(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s) :: "scc") break;
    case V: LOOP(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(v)) break;
    case S4: LOOP(REP16("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") : "+s"(s), "+s"(s2), "+s"(s3), "+s"(s4) :: "scc") break;
    case V4: LOOP(REP16("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1\n\t") : "+v"(v), "+v"(v2), "+v"(v3), "+v"(v4)) break;
    case M: LOOP(REP16("v_readlane_b32 %0, %2, %0\n\ts_and_b32 %0, %0, 63\n\ts_lshl_b64 %1, %1, 1\n\tv_add_u32 %3, %3, %0\n\t") : "+s"(s), "+s"(m), "+v"(v2), "+v"(v) :: "scc") break;
    case L: LOOP(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 %0, 2, %0\n\tv_and_b32 %0, 0xfc, %0\n\t") : "+v"(a) :: "memory") break;
    default: break;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = hw + (s + s2 + s3 + s4 + (uint32_t)m == 0x12345 ? 1 : 0) + (v + v2 + v3 + v4 + a == 0x54321 ? 1 : 0);
    }
}
int main()
{
    const int configs[][5] = {
        {1, S}, {1, V}, {1, S4}, {1, V4}, {1, M}, {1, L},
        {2, S, S}, {2, V, V}, {2, S, V}, {2, S4, V4}, {2, S4, S4}, {2, V4, V4}, {2, M, M}, {2, M, S}, {2, M, V},
        {4, S, S, S, S}, {4, V, V, V, V}, {4, S, S, V, V}, {4, M, M, M, M}, {4, S4, S4, V4, V4}, {4, M, M, L, L}, {4, S, V, L, M},
    };
    const uint32_t iters = 500, nwg = 256;
    unsigned long long *d; int *dr;
    if (hipMalloc(&d, nwg * 16 * 2 * 8) != hipSuccess || hipMalloc(&dr, 16) != hipSuccess) return 1;
    std::vector<unsigned long long> h(nwg * 32);
    double mhz_ratio = 1.0;  // s_memtime ticks: 100 MHz on this part; clocks measured against a known stream below
    for (auto &c : configs) {
        const int K = c[0];
        if (hipMemcpy(dr, &c[1], 16, hipMemcpyHostToDevice) != hipSuccess) return 1;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(nwg), dim3(256 * K), 0, 0, d, dr, K, iters, 3u);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 2; }
        }
        if (hipMemcpy(h.data(), d, nwg * 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        // workgroup 0: per role the ticks per instruction (mean over its 4 waves), and whether the K waves of a role index sit on 4 SIMDs
        printf("K=%d roles", K);
        double per_slot = 0;
        for (int r = 0; r < K; r++) {
            double t = 0; uint32_t simds = 0;
            for (int w = 4 * r; w < 4 * r + 4; w++) {
                t += (double)h[w * 2] / 4;
                simds |= 1u << ((h[w * 2 + 1] >> 4) & 3u);
            }
            const double tpi = t / (iters * 256.0);
            printf("  %s: %.3f ticks/instr%s", role_name[c[1 + r]], tpi, simds == 15 ? "" : " (!simd spread)");
            per_slot += 1.0 / tpi;
        }
        printf("   | sum of instr per tick per SIMD: %.3f\n", per_slot);
    }
    (void)mhz_ratio;
    return 0;
}
