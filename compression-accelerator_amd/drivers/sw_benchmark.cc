// sw_benchmark.cc -- the reference's software harness shape (tests/swcompression.cc:43-45,
// workloads/snappy/overlay/root/compress.cc:12-34) on the MI355X-backed snappy:: facade:
// snappy::Compress 20 times per input, report the mean time and the compressed length, then
// check snappy::Uncompress gives the input back.
//
// Build:  g++ -std=c++11 -Iinclude sw_benchmark.cc -lsnappy_hip
// Usage:  sw_benchmark <type> <file> [...]   one CSV row per file, columns of the reference's
//         benchmark/sw_results.csv with nanoseconds in place of Rocket cycles:
//         type;length;ns;compressed-length
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "snappy/snappy.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <type> <file> [...]\n", argv[0]);
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
        const std::string input((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::string packed, back;
        size_t clen = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < runs; i++)
            clen = snappy::Compress(input.data(), input.size(), &packed);
        auto t1 = std::chrono::steady_clock::now();
        long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / runs;
        printf("%s;%zu;%lld;%zu\n", argv[1], input.size(), ns, clen);
        if (!snappy::Uncompress(packed.data(), clen, &back) || back != input) {
            fprintf(stderr, "%s: round trip FAILED\n", argv[a]);
            bad = 1;
        }
    }
    return bad;
}
