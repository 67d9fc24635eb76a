// tools/rolecount.hip -- who compresses how many blocks of a launch of the team kernel: the teams or the lone waves of
// its workgroups?  (-DSZ_EXP_COUNT build of the kernels: a lone wave's length carries bit 31.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_EXP_COUNT -I../include -I../compression-accelerator_amd/csrc rolecount.hip -o rolecount
// usage: rolecount <raw input file> [n_blocks]
#include "../compression-accelerator_amd/csrc/snappy_compress.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); size_t L = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> h(L);
    if (fread(h.data(), 1, L, f) != L) return 1;
    fclose(f);
    uint32_t nb = argc > 2 ? atoi(argv[2]) : 16384;
    std::vector<uint8_t> in((size_t)nb * 65536);
    for (size_t i = 0; i < in.size(); i++) in[i] = h[i % L];
    uint8_t *d_in, *d_out; uint32_t *d_len;
    if (hipMalloc(&d_in, in.size()) != hipSuccess || hipMalloc(&d_out, (size_t)nb * 76544) != hipSuccess || hipMalloc(&d_len, nb * 4) != hipSuccess) return 2;
    if (hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
    szk::DualScratch sc{};
    sc.n_slots = 1024; sc.min_blocks = 2; sc.mode = szk::kDualAlways; sc.kernel = szk::kKernelTeam;
    if (hipMalloc(&sc.tables, (size_t)sc.n_slots * szk::kDualTablesPerGroup * szk::kDualTableBytes) != hipSuccess || hipMalloc(&sc.select, 64) != hipSuccess || hipMemset(sc.select, 0, 64) != hipSuccess) return 2;
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0);
        szk::launch_compress_blocks(d_in, 65536, nullptr, nb, d_out, 76544, d_len, 1, &sc, 0);
        (void)hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed\n"); return 3; }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint32_t> ol(nb);
        if (hipMemcpy(ol.data(), d_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        uint32_t sel[16] = {0};
        for (uint32_t b = 0; b < nb; b++) sel[(ol[b] >> 31) ? 12 : 13]++;
        printf("team %d + lone %d waves, nb=%u: %.3f ms %.1f GB/s; blocks by lone waves %u (%.1f %%), by teams %u\n", szk::kTeamWaves, szk::kLoneWaves, nb, ms,
               nb * 65536.0 / ms / 1e6, sel[12], 100.0 * sel[12] / nb, sel[13]);
    }
    return 0;
}
