// tools/c3prof.hip -- where an incompressible block's time goes in the single kernel: the library's kernels built with
// -DSZ_PROF (s_memtime stamps, 100 MHz), 32 768 blocks of tiled randomASCII through the instance the sampled choice
// uses.  Phases 28-31 are the block's (entry -> stream requested -> main loop -> main loop done -> return), 0-7 the
// main loop's own (dense window pieces, strided batches).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSZ_PE -I../include -I../compression-accelerator_amd/csrc c3prof.hip -o c3prof
#include "../compression-accelerator_amd/csrc/snappy_compress.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); size_t L = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> h(L); if (fread(h.data(), 1, L, f) != L) return 1; fclose(f);
    uint32_t nb = argc > 2 ? atoi(argv[2]) : 32768;
    std::vector<uint8_t> in((size_t)nb * 65536); for (size_t i = 0; i < in.size(); i++) in[i] = h[i % L];
    uint8_t *d_in, *d_out; uint32_t *d_len, *d_sel;
    hipMalloc(&d_in, in.size()); hipMalloc(&d_out, (size_t)nb * 76544); hipMalloc(&d_len, nb * 4); hipMalloc(&d_sel, 64);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice); hipMemset(d_sel, 0, 64);  // select = 0: "incompressible"
    const char *names[32] = {"dense: window load+hash", "dense: lookup", "dense: cand+compare", "dense: walk", "dense: emit", "dense: long match",
                             "strided: batch", "strided: hit+emit", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "",
                             "BLOCK: entry -> stream requested", "BLOCK: -> main loop (wait, probes, clear)", "BLOCK: main loop", "BLOCK: -> return (stores issued)"};
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        hipLaunchKernelGGL(szk::compress_blocks_kernel, dim3(nb), dim3(64), 0, 0, d_in, (uint64_t)65536, (const uint32_t *)nullptr, (uint64_t)0, nb,
                           d_out, (uint64_t)76544, d_len, 1u, (const uint32_t *)d_sel);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            std::vector<unsigned long long> p((size_t)65536 * 4);
            hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(szk::g_pe), p.size() * 8);
            const uint32_t m = nb < 65536 ? nb : 65536;
            printf("nb=%u kernel %.3f ms (with stamps)\n", nb, ms);
            double tot = 0;
            for (int i = 0; i < 4; i++) {
                double sum = 0; for (uint32_t b = 0; b < m; b++) sum += (double)p[(size_t)b * 4 + i];
                tot += sum / m;
                printf("%-44s avg %9.1f clocks per block\n", names[28 + i], sum / m);
            }
            printf("sum %.1f clocks per block; 4 blocks per CU in flight: %.3f ms per %u blocks at the clock the kernel time implies (%.2f GHz)\n",
                   tot, ms, nb, tot * nb / 1024.0 / (ms * 1e6));
        }
    }
    return 0;
}
