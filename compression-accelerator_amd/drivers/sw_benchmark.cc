// sw_benchmark.cc -- the reference's software harness shape (tests/swcompression.cc:43-45,
// workloads/snappy/overlay/root/compress.cc:12-34) on the MI355X-backed snappy:: facade:
// snappy::Compress 20 times per input, report the mean time and the compressed length, then
// check snappy::Uncompress gives the input back.
//
// Build:  g++ -std=c++11 -Iinclude sw_benchmark.cc -lsnappy_hip
// Usage:  sw_benchmark [-r] <type> <file> [...]   one CSV row per file, columns of the reference's
//         benchmark/sw_results.csv with nanoseconds in place of Rocket cycles:
//         type;length;ns;compressed-length
//         -r: snappy::RawCompress into a caller's buffer of MaxCompressedLength bytes (the call of
//         tests/swcompression.cc:45) instead of snappy::Compress into a std::string (compress.cc:17).  The string form
//         pays std::string::resize(MaxCompressedLength) on every call -- a memset of the part beyond the last result,
//         on the CPU, 1.6 GiB for 2 GiB of text -- exactly as google/snappy's own Compress does with libstdc++.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "snappy/snappy.h"

int main(int argc, char **argv)
{
    bool raw = false;
    if (argc > 1 && std::string(argv[1]) == "-r") {
        raw = true;
        argv++;
        argc--;
    }
    if (argc < 3) {
        fprintf(stderr, "usage: %s [-r] <type> <file> [...]\n", argv[0]);
        return 2;
    }
    const int runs = 20;
    int bad = 0;
    printf("type;length;ns;compressed-length\n");
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        if (!f) {
            perror(argv[a]);
            return 2;
        }
        f.seekg(0, std::ios::end);
        std::string input((size_t)f.tellg(), '\0');
        f.seekg(0);
        f.read(&input[0], (std::streamsize)input.size());
        std::string packed, back;
        size_t clen = 0;
        if (raw)
            packed.resize(snappy::MaxCompressedLength(input.size()));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < runs; i++) {
            if (raw)
                snappy::RawCompress(input.data(), input.size(), &packed[0], &clen);
            else
                clen = snappy::Compress(input.data(), input.size(), &packed);
        }
        auto t1 = std::chrono::steady_clock::now();
        long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / runs;
        printf("%s;%zu;%lld;%zu\n", argv[1], input.size(), ns, clen);
        // the way back, timed too (not a column of the reference's table: to stderr)
        bool ok = snappy::Uncompress(packed.data(), clen, &back);
        auto u0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 5 && ok; i++)
            ok = snappy::Uncompress(packed.data(), clen, &back);
        auto u1 = std::chrono::steady_clock::now();
        fprintf(stderr, "uncompress;%zu;%lld\n", input.size(),
                (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(u1 - u0).count() / 5));
        if (!ok || back != input) {
            fprintf(stderr, "%s: round trip FAILED\n", argv[a]);
            bad = 1;
        }
    }
    return bad;
}
