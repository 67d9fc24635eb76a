// include/snappy/snappy.h -- the snappy:: C++ surface the reference's software harness includes
// (`#include "snappy/snappy.h"`: tests/swcompression.cc:4, workloads/snappy/overlay/root/compress.cc:3),
// backed by the MI355X kernels through the C ABI of include/snappy_hip.h.
//
// Header-only inline wrappers: link with -lsnappy_hip.  Semantics follow google/snappy 1.1.8's
// public header for the calls on the hot path; output bytes are identical to that release.
// There is no CPU fallback: if no GPU is usable, Compress() prints the reason and aborts
// (the reference API has no error channel for compression), Uncompress() returns false.
#ifndef SNAPPY_HIP_FACADE_SNAPPY_H_
#define SNAPPY_HIP_FACADE_SNAPPY_H_

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../snappy_hip.h"

namespace snappy {

inline size_t MaxCompressedLength(size_t source_bytes) { return sz_max_compressed_length(source_bytes); }

// tests/swcompression.cc:27,45
inline void RawCompress(const char *input, size_t input_length, char *compressed, size_t *compressed_length)
{
    size_t n = 0;
    int rc = sz_compress(input, input_length, compressed, MaxCompressedLength(input_length), &n);
    if (rc != SZ_OK) {
        fprintf(stderr, "snappy (MI355X): compression failed: %s (hip error %d)\n", sz_strerror(rc),
                sz_last_hip_error());
        abort();
    }
    *compressed_length = n;
}

// workloads/snappy/overlay/root/compress.cc:17.  As in google/snappy, the string is first made MaxCompressedLength
// long -- with libstdc++ a memset of everything beyond its old size, on the CPU (1.6 GiB when 2 GiB of text are
// compressed into a string that held the last result: ~50 ms) --; here that runs between sz_compress_begin and
// sz_compress_finish, while the input is on its way to the GPU and being compressed.
inline size_t Compress(const char *input, size_t input_length, std::string *compressed)
{
    int rc = sz_compress_begin(input, input_length);
    size_t n = 0;
    if (rc == SZ_OK) {
        try {
            compressed->resize(MaxCompressedLength(input_length));
        } catch (...) {
            sz_compress_finish(NULL, 0, NULL);  // no room for the result: the call is abandoned, the thread free for the next
            throw;
        }
        rc = sz_compress_finish(&(*compressed)[0], compressed->size(), &n);
    }
    if (rc != SZ_OK) {
        fprintf(stderr, "snappy (MI355X): compression failed: %s (hip error %d)\n", sz_strerror(rc),
                sz_last_hip_error());
        abort();
    }
    compressed->resize(n);
    return n;
}

inline bool GetUncompressedLength(const char *compressed, size_t compressed_length, size_t *result)
{
    return sz_get_uncompressed_length(compressed, compressed_length, result) == SZ_OK;
}

inline bool RawUncompress(const char *compressed, size_t compressed_length, char *uncompressed)
{
    size_t ulen = 0, n = 0;
    if (!GetUncompressedLength(compressed, compressed_length, &ulen))
        return false;
    return sz_uncompress(compressed, compressed_length, uncompressed, ulen, &n) == SZ_OK;
}

// tests/swcompression.cc:31
inline bool Uncompress(const char *compressed, size_t compressed_length, std::string *uncompressed)
{
    size_t ulen = 0;
    if (!GetUncompressedLength(compressed, compressed_length, &ulen))
        return false;
    uncompressed->resize(ulen);
    size_t n = 0;
    char dummy = 0;
    char *dst = ulen ? &(*uncompressed)[0] : &dummy;
    return sz_uncompress(compressed, compressed_length, dst, ulen, &n) == SZ_OK;
}

inline bool IsValidCompressedBuffer(const char *compressed, size_t compressed_length)
{
    std::string tmp;
    return Uncompress(compressed, compressed_length, &tmp);
}

}  // namespace snappy

#endif  // SNAPPY_HIP_FACADE_SNAPPY_H_
