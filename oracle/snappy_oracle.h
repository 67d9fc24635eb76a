/*
 * oracle/snappy_oracle.h -- CPU restatement of Snappy 1.1.8 block compression.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under compression-accelerator_amd/ or
 * include/ may include, link or dlopen this.  Legal users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * See snappy_oracle.c for the reference citations and for how parity is pinned.
 */
#ifndef SNAPPY_ORACLE_H
#define SNAPPY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_BLOCK 65536u /* Snappy fragment size */

/* 32 + n + n/6 */
size_t oracle_max_compressed_length(size_t n);

/* Body of one fragment (n <= 65536), no varint preamble.  dst must hold
 * oracle_max_compressed_length(n) bytes.  Returns bytes written. */
size_t oracle_compress_fragment(const uint8_t *in, size_t n, uint8_t *dst);

/* Whole raw stream: varint32(n) + bodies of consecutive 64 KiB fragments.
 * Equals snappy::RawCompress / snappy::Compress of Snappy 1.1.8. */
size_t oracle_raw_compress(const uint8_t *in, size_t n, uint8_t *dst);

/* Parses the varint preamble.  Returns number of header bytes, 0 on error. */
size_t oracle_get_uncompressed_length(const uint8_t *src, size_t n, uint32_t *result);

/* Decodes a raw stream.  Returns 0 on success, non-zero on malformed input or
 * if dst_cap is too small.  *out_len receives the decoded size. */
int oracle_uncompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_cap, size_t *out_len);

/* Batch form used by tests and by bench.py's cpu_baseline: block b is
 * src[b*in_stride .. +in_len[b]) (in_len == NULL: every block is in_stride
 * bytes), written as varint + body (per_block_varint != 0) or body only into
 * dst + b*dst_stride, length into out_len[b].  Blocks are split statically
 * over n_threads pthreads. */
int oracle_compress_blocks(const uint8_t *src, size_t in_stride, const uint32_t *in_len,
                           uint32_t n_blocks, uint8_t *dst, size_t dst_stride,
                           uint32_t *out_len, int per_block_varint, int n_threads);

/* Timed form for the CPU baseline: n_threads pinned pthreads created once, one
 * untimed warm-up pass, then `reps` timed passes over the same caller-owned
 * buffers (barrier between passes); *seconds = wall time of the timed passes.
 * Fixed-size blocks only. */
int oracle_time_compress_blocks(const uint8_t *src, size_t in_stride, uint32_t n_blocks, uint8_t *dst,
                                size_t dst_stride, uint32_t *out_len, int per_block_varint,
                                int n_threads, int reps, double *seconds);

/* Same harness around an external compressor with libsnappy's C signature
 * int snappy_compress(const char*, size_t, char*, size_t*) -- the image's real
 * google/snappy, timed exactly like the port (bench.py cpu_baseline). */
int oracle_time_external_blocks(void *snappy_compress_fn, const uint8_t *src, size_t in_stride, uint32_t n_blocks,
                                uint8_t *dst, size_t dst_stride, uint32_t *out_len, int n_threads, int reps,
                                double *seconds);

/* FNV-1a 64 (offset basis 1469598103934665603, prime 1099511628211). */
uint64_t oracle_fnv1a64(const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
