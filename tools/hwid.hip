// tools/hwid.hip -- where do the waves of the team kernel's workgroups land?  1024 workgroups of 4 waves with the
// team kernel's LDS footprint (4 per CU); every wave records HW_ID and XCC_ID.  Prints, per CU, which workgroup's
// which wave sits on which SIMD.   hipcc --offload-arch=gfx950 -O3 hwid.hip -o hwid
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t spin)
{
    __shared__ uint8_t pad[38000];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = (uint8_t)hw;
    for (uint32_t i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(8);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + w) * 2] = hw;
        out[(blockIdx.x * 4 + w) * 2 + 1] = xcc + (pad[threadIdx.x] & 0);
    }
}
int main()
{
    const uint32_t nwg = 1024;
    uint32_t *d; hipMalloc(&d, nwg * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, d, 20000u);
    std::vector<uint32_t> h(nwg * 8);
    if (hipMemcpy(h.data(), d, nwg * 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, std::vector<std::tuple<uint32_t, uint32_t, uint32_t>>> cu;
    for (uint32_t b = 0; b < nwg; b++)
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15u;
            const uint32_t simd = (hw >> 4) & 3u, cuid = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            cu[{xcc, se, sh, cuid}].push_back({b, w, simd});
        }
    printf("%zu distinct CUs\n", cu.size());
    int shown = 0; uint32_t hist[8] = {0};
    for (auto &kv : cu) {
        uint32_t teams_on_simd[4] = {0, 0, 0, 0};
        for (auto &e : kv.second) if (std::get<1>(e) < 2) teams_on_simd[std::get<2>(e)]++;
        for (int s = 0; s < 4; s++) hist[teams_on_simd[s] < 7 ? teams_on_simd[s] : 7]++;
        if (shown++ < 6) {
            printf("xcc %u se %u sh %u cu %2u:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
            for (auto &e : kv.second) printf("  wg%4u.w%u->simd%u", std::get<0>(e), std::get<1>(e), std::get<2>(e));
            printf("\n");
        }
    }
    printf("SIMDs by number of team waves (waves 0,1 of a workgroup) on them: ");
    for (int i = 0; i < 8; i++) printf(" %d:%u", i, hist[i]);
    printf("\n");
    return 0;
}
