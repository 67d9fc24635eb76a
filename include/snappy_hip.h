/*
 * include/snappy_hip.h -- C ABI of the MI355X-native Snappy block compressor.
 *
 * This is the drop-in boundary for the ONE hot path of nullromo/compression-accelerator:
 *     compress(src, len, dst) -> compressed_len
 * on independent <= 64 KiB blocks, byte-identical to software Snappy 1.1.8.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference repo):
 *
 *   sz_rocc_cmd / rocc_cmd ........ the RoCC custom-3 command port:
 *        funct 2 SET_LENGTH(rs1=len), funct 0 COMPRESS(rs1=src, rs2=dst) -> rd=compressed_len,
 *        funct 1 UNCOMPRESS(rs1=src, rs2=dst) -> rd=0 on success
 *        (tests/compression.c:8-15,20-39; src/main/scala/example/CompressionAccelerator.scala:54-73,
 *         361-389,405-407; instruction encoders tests/rocc.h:30-46)
 *   sz_compress / sz_uncompress ... snappy::Compress / snappy::Uncompress as called at
 *        tests/swcompression.cc:27,31,45 and workloads/snappy/overlay/root/compress.cc:17
 *        (C++ facade over these two: include/snappy/snappy.h)
 *   sz_compress_blocks_hip ........ the accelerator datapath itself (match finder + hash table +
 *        copy emitter + literal streamer: MatchFinder.scala:26-77, HashTable.scala:11-54,
 *        CopyCompress.scala:44-190,211-273, CompressionAccelerator.scala:98-109,226-354) for a
 *        batch of blocks already resident in HBM -- the form a multi-block caller
 *        (workloads/snappy/overlay/root/run-all.sh style batch) binds.
 *   sz_pack_stream_hip ............ turns per-block outputs into the single stream snappy::Compress
 *        emits (one varint preamble + concatenated fragment bodies).
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; 0 on success,
 * negative SZ_E* on failure; no exceptions cross this boundary; no CPU fallback -- without a
 * usable GPU every compute entry point returns SZ_ENODEV / SZ_EHIP.
 * Device entry points are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 * default stream) and never synchronise.  The compress entry points keep a 68 MiB workspace per
 * (device, stream) for the kernels that take compressible launches (global hash tables, launch state,
 * block queue); it is allocated by the first call on a stream that has enough blocks and is not being
 * captured (a call made during graph capture before that simply uses the one-wave kernel), so make
 * one warm-up call before capturing a graph; sz_release_stream_resources() frees it.  Which kernel
 * runs is chosen per launch from a sample of the data -- per BLOCK when the sample meets more than one kind of data: the
 * one-wave kernel keeps the incompressible blocks, the persistent kernel takes the others -- (same output bytes whichever runs).
 *
 * Environment (all optional, read per call):
 *   SZ_KERNEL  forces a compress kernel: `single` (one wave per block, table in LDS: what incompressible
 *              launches get), `team` (two waves share a block and its LDS table + two waves per workgroup
 *              with tables in global memory: what compressible launches get), `dual` (rounds 1-2's kernel
 *              for compressible launches: three independent waves per workgroup; kept as a second
 *              implementation for differential tests and A/B).
 *              Unset: the sampled choice between `single` and `team` (launches of more than one block per SIMD; smaller
 *              ones: `single`).
 *   SZ_NO_ADOPT=1  the `team` kernel ends a launch as it did before round 6: a team whose queue is empty does not take
 *              over the block a lone wave of its workgroup is inside (same bytes, ~2 % slower on compressible launches;
 *              for A/B runs).
 *   SZ_DECODE_KERNEL  forces a decode kernel: `single` (one wave per stream), `pair` (two:
 *              one parses the next window while the other moves this one's bytes).  Unset: `pair` for launches
 *              of up to 4 096 streams, `single` above.  Same output either way.
 *   SZ_SERIAL_UNCOMPRESS=1  makes sz_uncompress skip the parallel fragment index (single-wavefront decode).
 *   SZ_NO_HOST_REGISTER=1   sz_compress / sz_uncompress on large buffers do not register the caller's memory for DMA
 *                           (the runtime's pageable copies instead: same bytes, slower).
 *   SZ_COPY_OUT=engine|kernel  sz_compress on large buffers copies every stage out with the DMA engines / with a kernel
 *                           (default: a kernel for stages that did not compress, the engines for the others).
 *   SZ_TEST_SEG_SHIFT       a hook of the test-suite (segment size of the fragment index).
 *   SZ_DEBUG_FLAGS          only read by diagnostic builds (-DSZ_ABLATION: timing ablations with wrong output).
 */
#ifndef SNAPPY_HIP_H
#define SNAPPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the 25 functions declared here are everything it exports. */
#if defined(__GNUC__) || defined(__clang__)
#define SZ_API __attribute__((visibility("default")))
#else
#define SZ_API
#endif

#define SZ_BLOCK_SIZE 65536u      /* Snappy fragment size: one block per wavefront */
#define SZ_SLOT_STRIDE 76544u     /* default output slot: >= sz_max_compressed_length(65536)=76490, 128-B multiple */
#define SZ_STREAM_INDEX_MAX_BYTES 0xfffc0000ull  /* longest raw stream sz_uncompress_stream_hip indexes: 32-bit positions with room
                                                    for one more segment and one more element behind the last byte (round 6;
                                                    before: 2^31 - 1, which 2 GiB of incompressible data exceed by 98 KB) */

#define SZ_OK 0
#define SZ_EINVAL (-22)
#define SZ_ENOSPC (-28)
#define SZ_EHIP (-5)              /* a HIP runtime call failed; see sz_last_hip_error() */
#define SZ_ENODEV (-19)           /* no usable GPU */
#define SZ_ECORRUPT (-74)         /* malformed compressed stream (EBADMSG) */

/* flags of sz_compress_blocks_hip */
#define SZ_FLAG_BLOCK_VARINT 1u   /* prepend varint32(in_len) to every block: each slot is a
                                     self-contained Snappy raw stream (== snappy::RawCompress of
                                     that block).  Without it slots hold fragment bodies only. */
#define SZ_FLAG_HW_COMPAT 2u      /* sz_compress_blocks_hip only: NOT software Snappy's bytes but the element stream of a
                                     FUNCTIONAL MODEL of the reference accelerator's datapath, read from its RTL by hand
                                     (512-entry table, every position hashed, 60-byte literals, 64-byte copies chained by
                                     `continue`: src/main/scala/example/{HashTable,MatchFinder,CopyCompress,
                                     CompressionAccelerator}.scala; checker tests/hw_model.py).  Valid Snappy
                                     elements -- with SZ_FLAG_BLOCK_VARINT any Snappy decoder reads the slot back -- and
                                     larger than Snappy's.  Fidelity to the hardware is pinned by SIZES only, and only
                                     in part: the length the RTL reports (benchmark/hw_results.csv) is this stream's
                                     size + the size of varint32(n) + 1 if the stream ends with a literal on 31 of its
                                     36 rows; on repeat-5000 and on every `real` row of 5 000 bytes or more (text whose
                                     tail repeats its head 4 185 bytes back) the hardware reports 8 and 204..1 257 bytes
                                     MORE than this model (tests/test_hw_model.py) -- its copy path loses a far match
                                     once, early (per-cycle validity of the data and candidate streams through one read
                                     aligner: MemoryReadAligner.scala:221-224, CopyCompress.scala:131-163), which a
                                     functional model does not have.  (Not the DMA: the whole input is resident before
                                     the datapath starts, MemoryController.scala:152-154.)  Not "the accelerator's
                                     format" beyond that evidence.  One wavefront per block, 64 positions per step: a compatibility
                                     mode, not a fast path. */

/* RoCC command surface constants (tests/compression.c:8-15) */
#define SZ_ROCC_CHANNEL 3
#define SZ_ROCC_COMPRESS 0
#define SZ_ROCC_UNCOMPRESS 1
#define SZ_ROCC_SET_LENGTH 2

/* ---- sizing ---------------------------------------------------------------------------- */

/* 32 + n + n/6, identical to snappy::MaxCompressedLength. */
SZ_API size_t sz_max_compressed_length(size_t n);

/* ---- device-resident batch API (the hot path) --------------------------------------------
 * Block b reads d_src + b*in_stride, d_in_len[b] bytes (d_in_len == NULL: in_stride bytes each;
 * every length must be <= 65536), and writes d_dst + b*dst_stride and d_out_len[b].
 * dst_stride must be >= sz_max_compressed_length(longest block) + 5.
 * One 64-lane wavefront per block, hash table in LDS.
 * d_out_len[b] is the compressed size, or a marker: 0xffffffff = the block was longer than 64 KiB (nothing written),
 * 0xfffffffe = the block was NOT compressed (a wave of the persistent kernel gave up after seconds without its team
 * mate -- a fault, never seen in 6 M soak blocks; the call still returns SZ_OK because it is asynchronous).  A block
 * that a sampled launch hands to the persistent kernel is marked 0xfffffffe before that kernel starts, so a block
 * nobody got to reads as not compressed, never as stale data.  sz_stream_offsets / sz_pack_stream_hip refuse markers. */
SZ_API int sz_compress_blocks_hip(const void *d_src, size_t in_stride, const uint32_t *d_in_len,
                           uint32_t n_blocks, void *d_dst, size_t dst_stride,
                           uint32_t *d_out_len, uint32_t flags, void *stream);

/* Measurement hook (bench.py's `roofline.achieved`; no reference counterpart): the NEXT sz_compress_blocks_hip /
 * sz_compress_stream_hip call of the calling host thread records the given hipEvent_t handles on its stream right
 * around the launch of the kernel that can be the launch's dominant one -- single_start / single_stop around the
 * one-wavefront-per-block kernel, multi_start / multi_stop around the persistent kernel for compressible launches
 * (both are enqueued when a data sample chooses between them; the one not chosen exits at once).  NULL handles are
 * skipped; one shot: the call that follows disarms the hook on every path (an argument error or SZ_FLAG_HW_COMPAT
 * records nothing and disarms it all the same; a launch of fixed blocks of <= 8 KiB has one kernel, recorded by the
 * single_* pair).  Events are created and read by the caller (hipEventCreate, hipEventElapsedTime). */
SZ_API int sz_profile_next_launch(void *single_start, void *single_stop, void *multi_start, void *multi_stop);

/* Same kernel for ONE contiguous input of n bytes (n < 2^32) cut into consecutive 64 KiB fragments
 * (the last one ragged): fragment k goes to d_slots + k*slot_stride, its length to d_len[k].
 * Number of fragments = ceil(n / 65536).  This is the fragment loop of snappy::Compress. */
SZ_API int sz_compress_stream_hip(const void *d_src, size_t n, void *d_slots, size_t slot_stride,
                           uint32_t *d_len, uint32_t flags, void *stream);

/* Exclusive prefix sum of d_len[0..n_blocks) + base into d_off[0..n_blocks], d_off[n_blocks] =
 * total; then gathers slot b (d_slots + b*slot_stride, d_len[b] bytes) to d_packed + d_off[b].
 * With base = size of the stream's varint preamble this yields the body of snappy::Compress.
 * d_off must hold n_blocks+1 uint64.  A d_len[b] above slot_stride (the compress kernels mark a
 * block they could not take with 0xffffffff) packs nothing for that slot and makes
 * d_off[n_blocks] = UINT64_MAX. */
SZ_API int sz_pack_stream_hip(const void *d_slots, size_t slot_stride, const uint32_t *d_len,
                       uint32_t n_blocks, uint64_t base, void *d_packed, uint64_t *d_off,
                       void *stream);

/* Batch decompression: stream b is d_src + b*src_stride, d_src_len[b] bytes (a raw Snappy
 * stream with preamble); decoded bytes go to d_dst + b*dst_stride (capacity dst_stride),
 * d_out_len[b] receives the decoded size, d_status[b] 0 or SZ_ECORRUPT/SZ_ENOSPC. */
SZ_API int sz_uncompress_blocks_hip(const void *d_src, size_t src_stride, const uint32_t *d_src_len,
                             uint32_t n_streams, void *d_dst, size_t dst_stride,
                             uint32_t *d_out_len, int32_t *d_status, void *stream);

/* The inverse of sz_pack_stream_hip for a caller that kept the offsets: fragment b of the packed
 * stream is d_packed[d_off[b], d_off[b+1]) (a body, no preamble; d_off as written by
 * sz_pack_stream_hip, d_off[0] = size of the stream's preamble) and decodes to
 * d_dst + b*65536: 64 KiB each, the last one the rest of total_ulen.  n_blocks must be
 * ceil(total_ulen / 65536).  All fragments are decoded in parallel (one wavefront each) -- the
 * way to decompress one large snappy::Compress stream fast; without the offsets a stream has to
 * be parsed serially (sz_uncompress_blocks_hip with one stream, sz_uncompress).
 * d_out_len[b] / d_status[b] as for sz_uncompress_blocks_hip. */
SZ_API int sz_uncompress_packed_hip(const void *d_packed, const uint64_t *d_off, uint32_t n_blocks,
                             uint64_t total_ulen, void *d_dst, uint32_t *d_out_len, int32_t *d_status,
                             void *stream);

/* ONE raw stream (preamble + elements) resident in HBM, decoded in parallel without side
 * information: the compressed offsets of its 64 KiB fragments are found on the GPU (64 hypothetical
 * parses per 16 KiB segment, then the true one through their maps), then every fragment is decoded by
 * its own wavefront.  body = size of the varint preamble (1..5), ulen = the decoded size it
 * announces (sz_get_uncompressed_length on the first bytes), n <= SZ_STREAM_INDEX_MAX_BYTES; d_src must be readable up to
 * n + 8; d_dst holds ulen bytes; d_work holds sz_uncompress_stream_work_bytes(n, ulen).
 * Asynchronous.  d_status3[0..2] = {the stream does not parse, an element straddles a 64 KiB output
 * boundary, number of fragments that failed to decode}: all zero = d_dst is the decoded stream.
 * Otherwise the stream is malformed OR simply was not written in 64 KiB fragments (every stock
 * Snappy compressor does write them): decode it with sz_uncompress_blocks_hip as one stream, which
 * also tells the two apart. */
SZ_API size_t sz_uncompress_stream_work_bytes(size_t n, uint64_t ulen);
SZ_API int sz_uncompress_stream_hip(const void *d_src, size_t n, size_t body, uint64_t ulen, void *d_dst,
                             void *d_work, size_t work_bytes, int32_t *d_status3, void *stream);

/* ---- host-buffer convenience API (what the snappy:: facade and the RoCC surface call) ---- */

/* Exactly the bytes of snappy::Compress(src, n): varint32(n) + bodies of consecutive 64 KiB
 * fragments.  n < 2^32.  Up to 128 MiB: one copy in, one launch over all fragments, one copy out.  Larger inputs run
 * as a pipeline of 64..256 MiB stages on three streams -- copy in of stage k+1 | fragments of stage k compressed and
 * packed behind stage k-1's bytes | copy out of stage k-1 straight to its final place in `dst` --, with the caller's
 * two buffers registered for DMA for the length of the call (hipHostRegister; if that fails -- or SZ_NO_HOST_REGISTER
 * is set -- the runtime's pageable copies are used: same bytes, no overlap of the two directions).  One
 * synchronisation at the end.  2 GiB on MI355X, PCIe included: 45 GB/s on incompressible data and on text (round 5,
 * unpipelined: 15 / 9). */
SZ_API int sz_compress(const void *src, size_t n, void *dst, size_t dst_cap, size_t *out_len);

/* sz_compress in two halves, for a caller that has something to do between them that needs no result yet --
 * snappy::Compress(const char*, size_t, std::string*) has: making room in the string, std::string::resize
 * (MaxCompressedLength), is a memset on the CPU of up to n + n/6 bytes (include/snappy/snappy.h).
 *   sz_compress_begin   starts the call: everything that does not need `dst` is enqueued (copies in, kernels) and the
 *                       function returns at once (0.3 ms for 2 GiB).  `src` must stay valid and unchanged until the
 *                       matching finish.  One call in flight per host thread: a second begin returns SZ_EINVAL.
 *   sz_compress_finish  completes it into `dst` (the copies out start while later stages still compress); same results
 *                       and errors as sz_compress.  sz_compress_finish(NULL, 0, NULL) abandons the call.  Whatever it
 *                       returns, nothing of the call is in flight afterwards and `src` is the caller's again.
 *                       Begin and finish are calls of ONE host thread (the state between them is the thread's own, like
 *                       the RoCC port's latched length); sz_release_thread_resources() drains and drops a call that
 *                       was begun and never finished.
 * sz_compress(src, n, dst, cap, &len) == sz_compress_begin(src, n) followed by sz_compress_finish(dst, cap, &len). */
SZ_API int sz_compress_begin(const void *src, size_t n);
SZ_API int sz_compress_finish(void *dst, size_t dst_cap, size_t *out_len);

/* snappy::Uncompress on the GPU.  *out_len receives the decoded size.  A stream of four or more
 * 64 KiB fragments (and at most SZ_STREAM_INDEX_MAX_BYTES bytes) is indexed in parallel on the GPU (where does each fragment begin) and
 * decoded one wavefront per fragment; shorter streams, and streams that were not written fragment by
 * fragment, are decoded by a single wavefront.  From 128 MiB of output on, the caller's buffers are registered for
 * DMA for the length of the call and the fragments are decoded in stages of 256 MiB, each copied out under the
 * decoding of the next (2 GiB of text: 37 GB/s of output, PCIe included); `dst` may then have been written in part
 * when the call reports a malformed stream. */
SZ_API int sz_uncompress(const void *src, size_t n, void *dst, size_t dst_cap, size_t *out_len);

/* snappy::GetUncompressedLength (host-side varint parse, no GPU needed). */
SZ_API int sz_get_uncompressed_length(const void *src, size_t n, size_t *result);

/* ---- RoCC command surface ----------------------------------------------------------------
 * Per-thread accelerator state (the latched length), one outstanding command, blocking, as
 * CompressionAccelerator.scala:72 (cmd.ready := !busy).  COMPRESS writes the full Snappy raw
 * stream (preamble + elements) so that UNCOMPRESS and any software Snappy can read it back.
 * Returns 0 and sets *rd, or a negative SZ_E* (bad channel/funct, no GPU). */
SZ_API int sz_rocc_cmd(int channel, int funct, uint64_t rs1, uint64_t rs2, uint64_t *rd);

/* Same, shaped for the instruction macros (include/rocc_host_shim.h): returns rd; on error
 * returns (uint64_t)-1 and records the code for sz_rocc_last_error(). */
SZ_API uint64_t rocc_cmd(int channel, int funct, uint64_t rs1, uint64_t rs2);
SZ_API int sz_rocc_last_error(void);

/* ---- multi-GPU (the Python side of the same two steps: sharding.py) -------
 * The path shards by contiguous ranges of independent 64 KiB blocks, one process per GPU, no data-path
 * collective: every rank selects its device (hipSetDevice -- the entry points above follow the calling
 * thread's current device), compresses blocks [begin, end) of the job into its own slots, and the ranks
 * exchange nothing but the per-block lengths (one all-gather of uint32, RCCL: ncclAllGather).  Both
 * helpers are host arithmetic and need no GPU. */
/* The balanced contiguous range of rank `rank` of `world` (the first n_blocks_total % world ranks get one
 * block more).  SZ_EINVAL if world == 0, rank >= world or a pointer is null. */
SZ_API int sz_shard_range(uint64_t n_blocks_total, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end);
/* Layout of the whole packed stream from the gathered lengths: offsets[i] = base + lens[0] + ... + lens[i-1]
 * for i = 0 .. n_blocks (base = bytes of the stream's preamble).  SZ_ECORRUPT, and *bad_block set if not
 * null, if a length is one of the kernels' markers (>= 0xfffffffd: the block was not compressed) or
 * exceeds `slot_stride`. */
SZ_API int sz_stream_offsets(const uint32_t *lens, uint64_t n_blocks, uint64_t base, uint64_t slot_stride,
                      uint64_t *offsets, uint64_t *bad_block);

/* ---- misc ------------------------------------------------------------------------------- */
/* Frees the calling thread's cached device workspace (its stream, five buffers and the 64 MiB below),
 * used by the host-buffer API and the RoCC surface.  It is NOT freed when the thread exits (the HIP
 * runtime may be gone by then): a thread pool that retires threads calls this in each of them first. */
SZ_API void sz_release_thread_resources(void);
/* Frees the 64 MiB workspace the compress entry points keep for `stream` on the current device
 * (waits for the stream first).  The workspace is created by the first launch on the stream that
 * has enough blocks for the persistent kernel (more than 4 x CUs: one per SIMD), never by small calls.  Call this before
 * destroying a stream that was used for compression if the process creates many; the next large
 * compress call on the stream allocates a new one.  Host threads may launch on one stream at the
 * same time: the launch sequence of each call is enqueued as a unit. */
SZ_API int sz_release_stream_resources(void *stream);
SZ_API int sz_device_count(void);            /* number of visible GPUs (0 if none / no driver) */
SZ_API int sz_last_hip_error(void);          /* last hipError_t seen by this thread */
SZ_API const char *sz_strerror(int code);
SZ_API const char *sz_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPY_HIP_H */
