// tools/hostxfer.hip -- what the host link of this box does, to size the host-buffer API's pipeline (sz_compress /
// sz_uncompress on pageable caller memory): pageable and pinned copies in both directions, alone and at once;
// hipHostRegister / Unregister of caller memory in chunks; CPU staging copies (pageable <-> pinned) by 1..16 threads.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/hostxfer tools/hostxfer.hip -lpthread      Run: tools/hostxfer [MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(uint8_t *dst, const uint8_t *src, size_t n, int threads)
{
    std::vector<std::thread> th;
    const size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; t++) {
        const size_t b = (size_t)t * per, e = b + per < n ? b + per : n;
        if (b < e)
            th.emplace_back([=] { memcpy(dst + b, src + b, e - b); });
    }
    for (auto &t : th)
        t.join();
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    uint8_t *page_a = (uint8_t *)malloc(n), *page_b = (uint8_t *)malloc(n), *pin_a, *pin_b, *d_a, *d_b;
    for (size_t i = 0; i < n; i += 4096)
        page_a[i] = (uint8_t)i, page_b[i] = 1;
    CK(hipHostMalloc((void **)&pin_a, n, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin_b, n, hipHostMallocDefault));
    memset(pin_a, 1, n);
    memset(pin_b, 2, n);
    CK(hipMalloc((void **)&d_a, n));
    CK(hipMalloc((void **)&d_b, n));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    printf("hardware threads %u, buffer %zu MiB\n", std::thread::hardware_concurrency(), n >> 20);
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipMemcpyAsync(d_a, page_a, n, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("pageable H2D            %6.1f GB/s\n", n / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(page_b, d_a, n, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        printf("pageable D2H            %6.1f GB/s\n", n / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(d_a, pin_a, n, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("pinned   H2D            %6.1f GB/s\n", n / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(pin_b, d_b, n, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        printf("pinned   D2H            %6.1f GB/s\n", n / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(d_a, pin_a, n, hipMemcpyHostToDevice, s0));
        CK(hipMemcpyAsync(pin_b, d_b, n, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        printf("pinned   H2D + D2H at once  %6.1f GB/s each way\n", n / (now() - t) / 1e9);
    }
    // chunks of 16 MiB: what a pipeline would issue
    for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
        double t = now();
        for (size_t o = 0; o < n; o += chunk)
            CK(hipMemcpyAsync(d_a + o, pin_a + o, chunk, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("pinned   H2D in %3zu MiB chunks %6.1f GB/s\n", chunk >> 20, n / (now() - t) / 1e9);
    }
    // register caller memory in place
    for (size_t chunk : {(size_t)64 << 20, (size_t)256 << 20, n}) {
        double t = now(), treg = 0, tun = 0, tcp = 0;
        for (size_t o = 0; o + chunk <= n; o += chunk) {
            double a = now();
            CK(hipHostRegister(page_a + o, chunk, hipHostRegisterDefault));
            double b = now();
            CK(hipMemcpyAsync(d_a + o, page_a + o, chunk, hipMemcpyHostToDevice, s0));
            CK(hipStreamSynchronize(s0));
            double c = now();
            CK(hipHostUnregister(page_a + o));
            double d = now();
            treg += b - a, tcp += c - b, tun += d - c;
        }
        printf("register %4zu MiB chunks: register %6.1f GB/s, copy %6.1f GB/s, unregister %6.1f GB/s, serial total %6.1f GB/s\n", chunk >> 20,
               n / treg / 1e9, n / tcp / 1e9, n / tun / 1e9, n / (now() - t) / 1e9);
    }
    // CPU staging copies
    for (int threads : {1, 2, 4, 8, 12, 16}) {
        par_copy(pin_a, page_a, n, threads);
        double t = now();
        par_copy(pin_a, page_a, n, threads);
        double a = now() - t;
        t = now();
        par_copy(page_b, pin_b, n, threads);
        double b = now() - t;
        printf("CPU copy, %2d threads: pageable -> pinned %6.1f GB/s, pinned -> pageable %6.1f GB/s\n", threads, n / a / 1e9, n / b / 1e9);
    }
    // staging + H2D pipelined by hand: 4 threads copy chunk k+1 while chunk k travels
    for (int threads : {4, 8}) {
        const size_t chunk = (size_t)32 << 20;
        double t = now();
        hipEvent_t ev[2];
        CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        int k = 0;
        for (size_t o = 0; o < n; o += chunk, k++) {
            uint8_t *st = pin_a + (size_t)(k & 1) * chunk;
            if (k >= 2)
                CK(hipEventSynchronize(ev[k & 1]));
            par_copy(st, page_a + o, chunk, threads);
            CK(hipMemcpyAsync(d_a + o, st, chunk, hipMemcpyHostToDevice, s0));
            CK(hipEventRecord(ev[k & 1], s0));
        }
        CK(hipStreamSynchronize(s0));
        printf("staged H2D, 32 MiB chunks, %d copy threads, double-buffered: %6.1f GB/s\n", threads, n / (now() - t) / 1e9);
    }
    return 0;
}
