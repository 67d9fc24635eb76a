// tools/chainprof2.hip -- like chainprof.hip, for the multi-wave kernel: per-phase cycle stamps,
// separately for the waves whose table is in LDS and those whose table is in global memory.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_PROF -I../include -I../compression-accelerator_amd/csrc chainprof2.hip -o chainprof2
// usage: chainprof2 <raw input file> [n_blocks]
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
    uint32_t nb = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<uint8_t> in((size_t)nb * 65536);
    for (size_t i = 0; i < in.size(); i++) in[i] = h[i % L];
    uint8_t *d_in, *d_out; uint32_t *d_len;
    hipMalloc(&d_in, in.size()); hipMalloc(&d_out, (size_t)nb * 76544); hipMalloc(&d_len, nb * 4);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice);
    szk::DualScratch sc{};
    sc.n_slots = 1024; sc.min_blocks = 2; sc.mode = szk::kDualAlways;
    const size_t ntab = (size_t)sc.n_slots * szk::kDualTablesPerGroup;
    hipMalloc(&sc.tables, ntab * szk::kDualTableBytes); hipMemset(sc.tables, 0, ntab * szk::kDualTableBytes);
    hipMalloc(&sc.select, 64); hipMemset(sc.select, 0, 64);
    const char *names[32] = {"window load+hash", "table lookup issue", "cand loads+compare", "walk+commit+verify", "emit", "long match",
                             "strided batch", "strided hit", "walk (per pass)", "commit+readback (per pass)", "lim", "cut windows", "hits/window", "advance/window",
                             "F: enter (since last lap)", "F: visit top + walk prep", "F: walk loop", "F: exit + commit", "F: long: copy elements (vm drained)", "-",
                             "F: enter tile: validate/head", "F: flush", "F: compare (vm drained)", "F: same-tile validate", "F: stale patch after adopt", "F: spec head",
                             "F: long: drain before", "F: long: match_length", "", "", "", ""};
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long z[160] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(szk::g_prof), z, sizeof z);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        szk::launch_compress_blocks(d_in, 65536, nullptr, nb, d_out, 76544, d_len, 1, &sc, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long p[160];
        hipMemcpyFromSymbol(p, HIP_SYMBOL(szk::g_prof), sizeof p);
        if (rep == 1) {
            printf("nb=%u kernel %.3f ms  %.1f GB/s\n", nb, ms, nb * 65536.0 / ms / 1e6);
            for (int t = 0; t < 2; t++) {
                const unsigned long long *q = p + 64 * t;
                unsigned long long tot = 0;
                for (int i = 0; i < 10; i++) tot += q[i];
                for (int i = 14; i < 28; i++) tot += q[i];
                printf("--- %s waves: %llu windows + %llu tiles (%llu visits), %.0f ticks in all\n", t ? "GLOBAL-table" : "LDS-table", q[32 + 3], q[32 + 21], q[32 + 15], (double)tot);
                for (int i = 0; i < 32; i++)
                    if (q[32 + i])
                        printf("  %-28s total %12llu  count %9llu  avg %8.2f  (%4.1f%%)\n", names[i], q[i], q[32 + i], (double)q[i] / q[32 + i], (i < 10 || (i >= 14 && i < 28)) ? 100.0 * q[i] / tot : 0.0);
            }
            const char *ml[4] = {"first <= 64 bytes", "whole step, all equal", "whole step with the difference", "last step (masked) + final bytes"};
            printf("--- match_length, all waves (each lap drains vmcnt)\n");
            for (int i = 0; i < 4; i++)
                if (p[144 + i])
                    printf("  %-34s total %12llu  count %9llu  avg %8.2f\n", ml[i], p[128 + i], p[144 + i], (double)p[128 + i] / p[144 + i]);
        }
    }
    return 0;
}
