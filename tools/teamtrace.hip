// tools/teamtrace.hip -- time line of the team kernel's turns (snappy_compress.hip, TEAM FAST DENSE PATH; -DSZ_TRACE):
// the waves of the first blocks stamp s_memtime at the phase boundaries of every turn; this prints where the
// serial chain of a block (token acquired -> ... -> token acquired by the next wave) spends its cycles.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_TRACE -DSZ_TEAM_WAVES=3 -I../include -I../compression-accelerator_amd/csrc teamtrace.hip -o teamtrace3
// usage: teamtrace3 <raw input file> [n_blocks] [turns of block 0 to list]
#include "../compression-accelerator_amd/csrc/snappy_compress.hip"
#include <stdio.h>
#include <algorithm>
#include <map>
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
    int list = argc > 3 ? atoi(argv[3]) : 12;
    std::vector<uint8_t> in((size_t)nb * 65536);
    for (size_t i = 0; i < in.size(); i++) in[i] = h[i % L];
    uint8_t *d_in, *d_out; uint32_t *d_len;
    if (hipMalloc(&d_in, in.size()) != hipSuccess || hipMalloc(&d_out, (size_t)nb * 76544) != hipSuccess || hipMalloc(&d_len, nb * 4) != hipSuccess) return 2;
    if (hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
    szk::DualScratch sc{};
    sc.n_slots = 1024; sc.min_blocks = 2; sc.mode = szk::kDualAlways; sc.kernel = szk::kKernelTeam;
    if (hipMalloc(&sc.tables, (size_t)sc.n_slots * szk::kDualTablesPerGroup * szk::kDualTableBytes) != hipSuccess || hipMalloc(&sc.select, 64) != hipSuccess || hipMemset(sc.select, 0, 64) != hipSuccess) return 2;
    const int W = szk::kTeamWaves;
    const size_t ntr = (size_t)szk::kTraceBlocks * W * szk::kTraceCap;
    unsigned long long *d_tr;
    if (hipMalloc(&d_tr, ntr * 8) != hipSuccess) return 2;
    std::vector<unsigned long long> tr(ntr);
    for (int rep = 0; rep < 2; rep++) {
        if (hipMemset(d_tr, 0, ntr * 8) != hipSuccess) return 2;
        unsigned long long *ptr = rep == 1 ? d_tr : nullptr;  // (the first launch warms up, untraced)
        if (hipMemcpyToSymbol(HIP_SYMBOL(szk::g_trace), &ptr, sizeof ptr) != hipSuccess) return 2;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        szk::launch_compress_blocks(d_in, 65536, nullptr, nb, d_out, 76544, d_len, 1, &sc, 0);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed\n"); return 3; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("team of %d waves, nb=%u, launch %d: %.3f ms  %.1f GB/s\n", W, nb, rep, ms, nb * 65536.0 / ms / 1e6);
    }
    if (hipMemcpy(tr.data(), d_tr, ntr * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    // per block: stamps by turn
    const char *ph[11] = {"trip", "token", "checked", "patched", "visited", "long", "published", "op", "written", "end", "headed"};
    double fine[5] = {0}; unsigned long long nfine = 0; double seg[11] = {0}, handoff = 0, period = 0, opwait = 0, idle = 0, busy = 0, headt = 0; unsigned long long nturn = 0, nper = 0;
    for (uint32_t b = 0; b < szk::kTraceBlocks && b < nb; b++) {
        std::map<uint32_t, std::vector<long long>> turns;  // turn -> time of phase 0..9 (-1: none)
        std::map<uint32_t, int> who;
        for (int w = 0; w < W; w++) {
            const unsigned long long *q = &tr[((size_t)b * W + w) * szk::kTraceCap];
            for (uint32_t i = 0; i < szk::kTraceCap && q[i]; i++) {
                const uint32_t id = q[i] & 15u, turn = (uint32_t)(q[i] >> 4) & 0xffffffu;
                const long long t = (long long)(q[i] >> 28);
                // a stamp 0 ("starts waiting") carries the turn the wave waits for; 9 carries the turn + W it goes to next
                uint32_t key = id == 9 ? turn - W : turn;
                auto &v = turns[key];
                if (v.empty()) v.assign(16, -1);
                if (v[id] < 0) v[id] = t;
                who[key] = w;
            }
        }
        long long t00 = -1;
        int listed = 0;
        std::vector<long long> *prev = nullptr;
        for (auto &kv : turns) {
            auto &v = kv.second;
            if (v[1] < 0 || v[6] < 0) { prev = nullptr; continue; }
            if (t00 < 0) t00 = v[1];
            nturn++;
            for (int i = 2; i <= 6; i++) if (v[i] >= 0 && v[i - 1] >= 0) seg[i] += v[i] - v[i - 1];
            if (v[7] >= 0) opwait += v[7] - v[6];
            if (v[8] >= 0 && v[7] >= 0) seg[8] += v[8] - v[7];
            if (v[9] >= 0 && v[8] >= 0) seg[9] += v[9] - v[8];
            if (v[11] >= 0 && v[3] >= 0 && v[15] >= 0) { fine[0] += v[11] - v[3]; fine[1] += v[12] - v[11]; fine[2] += v[13] - v[12]; fine[3] += v[14] - v[13]; fine[4] += v[15] - v[14]; nfine++; }
            if (v[10] >= 0) idle += v[1] - v[10];
            if (v[9] >= 0) busy += v[9] - v[1];
            if (v[10] >= 0 && v[0] >= 0) headt += v[10] - v[0];
            if (prev && (*prev)[6] >= 0) { handoff += v[1] - (*prev)[6]; period += v[1] - (*prev)[1]; nper++; }
            if (b == 0 && listed < list && kv.first >= 40) {
                printf("  turn %5u wave %d  token@%7lld", kv.first, who[kv.first], v[1] - t00);
                for (int i = 2; i <= 9; i++) printf("  %s +%lld", ph[i], v[i] < 0 ? -1 : v[i] - v[1]);
                printf("  (head %lld, waited %lld)\n", v[0] < 0 || v[10] < 0 ? -1 : v[10] - v[0], v[10] < 0 ? -1 : v[1] - v[10]);
                listed++;
            }
            prev = &v;
        }
    }
    if (!nturn) { printf("no stamps\n"); return 0; }
    printf("%llu traced turns; per turn (ticks):\n", nturn);
    printf("  period token -> next wave's token   %8.1f  (over %llu consecutive pairs)\n", period / nper, nper);
    printf("    P-1, lookups checked              %8.1f\n", seg[2] / nturn);
    printf("    patch (first visit)               %8.1f\n", seg[3] / nturn);
    printf("    visits (walk, exit, commit)       %8.1f\n", seg[4] / nturn);
    if (nfine) printf("      first visit: entry lane + quit checks %.1f, walk %.1f, exit state + masks %.1f, conflict test %.1f, commit %.1f (each with one stamp: see 'long match length' for a stamp alone)\n", fine[0] / nfine, fine[1] / nfine, fine[2] / nfine, fine[3] / nfine, fine[4] / nfine);
    printf("    long match length                 %8.1f\n", seg[5] / nturn);
    printf("    publish                           %8.1f\n", seg[6] / nturn);
    printf("    hand-off (published -> acquired)  %8.1f\n", handoff / nper);
    printf("  off the chain: wait for op %8.1f, element writer %8.1f, op publish %8.1f, head of the next tile %8.1f\n", opwait / nturn, seg[8] / nturn, seg[9] / nturn, headt / nturn);
    printf("  a wave per turn of its own: busy %8.1f, waiting for the token %8.1f\n", busy / nturn, idle / nturn);
    return 0;
}
