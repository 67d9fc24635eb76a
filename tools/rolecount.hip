// tools/rolecount.hip -- who compresses how many blocks of a launch of the team kernel: the teams or the lone waves of
// its workgroups?  (-DSZ_EXP_COUNT build of the kernels: a lone wave's length carries bit 31.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_EXP_COUNT -I../include -I../compression-accelerator_amd/csrc rolecount.hip -o rolecount
// usage: rolecount <raw input file> [n_blocks] [mix]
#include "../compression-accelerator_amd/csrc/snappy_compress.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
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
    if (argc > 3 && !strcmp(argv[3], "mix")) {  // every third block a 4 KiB pattern x 16, every third noise
        unsigned long long x = 88172645463325252ull;
        for (uint32_t b = 0; b < nb; b++) {
            uint8_t *p = &in[(size_t)b * 65536];
            if (b % 3 == 1) for (uint32_t i = 4096; i < 65536; i++) p[i] = p[i - 4096];
            if (b % 3 == 2) for (uint32_t i = 0; i < 65536; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; p[i] = (uint8_t)(x >> 32); }
        }
    }
    uint8_t *d_in, *d_out; uint32_t *d_len;
    if (hipMalloc(&d_in, in.size()) != hipSuccess || hipMalloc(&d_out, (size_t)nb * 76544) != hipSuccess || hipMalloc(&d_len, nb * 4) != hipSuccess) return 2;
    if (hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
    szk::DualScratch sc{};
    sc.n_slots = 1024; sc.min_blocks = 2; sc.mode = szk::kDualAlways; sc.kernel = szk::kKernelTeam;
    if (argc > 3 && !strcmp(argv[3], "mix")) {  // as the library launches it: sampled, queue in cost order
        sc.mode = szk::kDualAuto;
        sc.queue_cap = szk::kDualQueueEntries;
        if (hipMalloc(&sc.queue, (size_t)sc.queue_cap * 4 * 2) != hipSuccess) return 2;
    }
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
        if (rep == 2) {  // when were the blocks finished?  (s_memrealtime, 100 MHz: from the first to the last finish, in tenths of that span)
            std::vector<unsigned long long> fin(nb);
            if (hipMemcpyFromSymbol(fin.data(), HIP_SYMBOL(szk::g_finish), (size_t)nb * 8) != hipSuccess) return 2;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (uint32_t b = 0; b < nb; b++) { if (fin[b] < t0) t0 = fin[b]; if (fin[b] > t1) t1 = fin[b]; }
            unsigned hist[2][10] = {{0}};
            for (uint32_t b = 0; b < nb; b++) {
                int k = (int)((fin[b] - t0) * 10 / (t1 - t0 + 1));
                hist[(ol[b] >> 31) ? 1 : 0][k]++;
            }
            printf("  blocks finished per tenth of the launch, by teams:");
            for (int k = 0; k < 10; k++) printf(" %u", hist[0][k]);
            printf("\n  by lone waves:                                   ");
            for (int k = 0; k < 10; k++) printf(" %u", hist[1][k]);
            printf("\n  first to last finish: %.3f ms\n", (t1 - t0) / 1e5);
            // the end of the launch (round 6): when was the last block taken out of the queue, how long did the blocks of
            // each kind of worker take, and how many were still being worked on k x 0.25 ms after the queue was empty
            std::vector<unsigned long long> sta(nb);
            if (hipMemcpyFromSymbol(sta.data(), HIP_SYMBOL(szk::g_start), (size_t)nb * 8) != hipSuccess) return 2;
            unsigned long long s0 = ~0ull, s1 = 0;
            double dur[2] = {0, 0}; unsigned cnt[2] = {0, 0};
            for (uint32_t b = 0; b < nb; b++) { if (sta[b] < s0) s0 = sta[b]; if (sta[b] > s1) s1 = sta[b]; int r = (ol[b] >> 31) ? 1 : 0; dur[r] += (double)(fin[b] - sta[b]); cnt[r]++; }
            printf("  first block taken at 0, last block taken at %.3f ms, last block finished at %.3f ms: the launch ends %.3f ms after its queue is empty\n",
                   (s1 - s0) / 1e5, (t1 - s0) / 1e5, (t1 - s1) / 1e5);
            printf("  mean time per block: teams %.3f ms, lone waves %.3f ms\n", cnt[0] ? dur[0] / cnt[0] / 1e5 : 0.0, cnt[1] ? dur[1] / cnt[1] / 1e5 : 0.0);
            printf("  blocks in work k x 0.25 ms after the queue was empty (teams / lone):");
            for (int k = 0; k < 20; k++) {
                unsigned long long at = s1 + (unsigned long long)k * 25000ull;
                unsigned w[2] = {0, 0};
                for (uint32_t b = 0; b < nb; b++) if (fin[b] > at) w[(ol[b] >> 31) ? 1 : 0]++;
                printf(" %u/%u", w[0], w[1]);
                if (w[0] + w[1] == 0) break;
            }
            // the blocks that were finished last: when were they taken?
            printf("\n  the 8 blocks finished last (role, taken at, finished at, ms):");
            std::vector<uint32_t> idx(nb);
            for (uint32_t b = 0; b < nb; b++) idx[b] = b;
            std::partial_sort(idx.begin(), idx.begin() + (nb < 8 ? nb : 8), idx.end(), [&](uint32_t x, uint32_t y) { return fin[x] > fin[y]; });
            for (uint32_t k = 0; k < 8 && k < nb; k++) printf(" [%s %.2f %.2f]", (ol[idx[k]] >> 31) ? "lone" : "team", (sta[idx[k]] - s0) / 1e5, (fin[idx[k]] - s0) / 1e5);
            printf("\n");
        }
    }
    return 0;
}
