// tools/hostpipe.cc -- sz_compress / sz_uncompress on large pageable host buffers, straight through the C ABI of a
// library given by path (so that a diagnostic build, e.g. tools/mkvariant.sh hosttime -DSZ_HOST_TIMING, can be looked
// at): per-call wall time, no files, no Python.
// Build: g++ -O2 -std=c++11 -o tools/hostpipe tools/hostpipe.cc -ldl     Run: tools/hostpipe <lib.so> <corpus-file> <MiB> [calls]
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef int (*codec_fn)(const void *, size_t, void *, size_t, size_t *);
typedef size_t (*bound_fn)(size_t);
typedef int (*begin_fn)(const void *, size_t);
typedef int (*finish_fn)(void *, size_t, size_t *);
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <lib.so> <corpus-file> <MiB> [calls]\n", argv[0]);
        return 2;
    }
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) {
        fprintf(stderr, "%s\n", dlerror());
        return 2;
    }
    codec_fn compress = (codec_fn)dlsym(h, "sz_compress"), uncompress = (codec_fn)dlsym(h, "sz_uncompress");
    bound_fn bound = (bound_fn)dlsym(h, "sz_max_compressed_length");
    begin_fn begin = (begin_fn)dlsym(h, "sz_compress_begin");
    finish_fn finish = (finish_fn)dlsym(h, "sz_compress_finish");
    FILE *f = fopen(argv[2], "rb");
    if (!f)
        return 2;
    fseek(f, 0, SEEK_END);
    const size_t cn = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> corpus(cn);
    if (cn == 0 || fread(corpus.data(), 1, cn, f) != cn) {  // (an empty corpus would never fill the input)
        fprintf(stderr, "%s: empty or unreadable\n", argv[2]);
        return 2;
    }
    const size_t n = (size_t)atoi(argv[3]) << 20;
    const int calls = argc > 4 ? atoi(argv[4]) : 6;
    std::vector<char> in(n), out(bound(n)), back(n);
    for (size_t o = 0; o < n; o += cn)
        memcpy(in.data() + o, corpus.data(), n - o < cn ? n - o : cn);
    if (argc > 5) {  // a second kind of data behind the first (a stream whose fragments are not all alike)
        FILE *g = fopen(argv[5], "rb");
        if (!g)
            return 2;
        fseek(g, 0, SEEK_END);
        const size_t bn = (size_t)ftell(g);
        fseek(g, 0, SEEK_SET);
        std::vector<char> second(bn);
        if (fread(second.data(), 1, bn, g) != bn)
            return 2;
        for (size_t o = n / 3 * 2; o < n; o += bn)
            memcpy(in.data() + o, second.data(), n - o < bn ? n - o : bn);
    }
    memset(out.data(), 1, out.size());
    memset(back.data(), 1, back.size());
    size_t clen = 0, ulen = 0;
    for (int i = 0; i < calls; i++) {
        const double t = now_ms();
        const int rc = compress(in.data(), n, out.data(), out.size(), &clen);
        const double dt = now_ms() - t;
        printf("compress   call %d: rc %d, %zu -> %zu bytes, %.2f ms = %.2f GB/s\n", i, rc, n, clen, dt, n / dt / 1e6);
        if (rc)
            return 1;
    }
    // the std::string shape: snappy::Compress's own sequence (include/snappy/snappy.h) on a string that is reused
    std::string str;
    for (int i = 0; i < calls && begin && finish; i++) {
        const double t = now_ms();
        int rc = begin(in.data(), n);
        str.resize(bound(n));
        const double t1 = now_ms();
        if (rc == 0)
            rc = finish(&str[0], str.size(), &clen);
        str.resize(clen);
        const double dt = now_ms() - t;
        printf("string     call %d: rc %d, %zu -> %zu bytes, %.2f ms (resize %.2f) = %.2f GB/s\n", i, rc, n, clen, dt, t1 - t, n / dt / 1e6);
        if (rc || memcmp(str.data(), out.data(), clen) != 0)
            return 1;
    }
    for (int i = 0; i < calls; i++) {
        const double t = now_ms();
        const int rc = uncompress(out.data(), clen, back.data(), n, &ulen);
        const double dt = now_ms() - t;
        printf("uncompress call %d: rc %d, %zu -> %zu bytes, %.2f ms = %.2f GB/s of output\n", i, rc, clen, ulen, dt, n / dt / 1e6);
        if (rc)
            return 1;
    }
    printf("round trip %s\n", ulen == n && memcmp(back.data(), in.data(), n) == 0 ? "ok" : "MISMATCH");
    return 0;
}
