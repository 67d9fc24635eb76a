// snappy_kernels.h -- internal launch interface between the C ABI (snappy_abi.cpp) and the
// gfx950 kernels (snappy_compress.hip, snappy_pack.hip, snappy_uncompress.hip; shared device helpers: wave_utils.h).  Not installed; the public boundary is include/snappy_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace szk {

constexpr uint32_t kFlagBlockVarint = 1u;  // == SZ_FLAG_BLOCK_VARINT
constexpr int32_t kStatusCorrupt = -74;    // == SZ_ECORRUPT
constexpr int32_t kStatusNoSpace = -28;    // == SZ_ENOSPC

// Workspace of the persistent kernels: n_slots x (waves - 1) hash tables of 16384 x uint16 (never cleared: a
// bitmap in LDS names the buckets a block has written), the launch state and the block queue (cost order); one launch sequence at a time.
enum DualMode : uint32_t { kDualAuto = 0, kDualAlways = 2 };
struct DualScratch {
    uint16_t *tables;
    uint32_t *select;     // launch state (kLaunchStateBytes, zero between launches where it matters): [0] repeats, [1] positions of the sample, [7] cost classes it met;
                          // [2] next entry of the persistent kernel's work; [5], [6] blocks of class 1, 2 in the queue; [15] arrivals of the sample's workgroups
    uint32_t *queue;      // block indices in cost order for the persistent kernels (+ one class byte per block in its upper half)
    uint32_t queue_cap;   // entries in `queue`; launches with more than half as many blocks are not ordered
    uint32_t n_slots;
    uint32_t min_blocks;  // launches with fewer blocks use the single kernel
    uint32_t mode;        // kDualAuto: single kernel or persistent kernel per launch, from a data sample; kDualAlways: the persistent kernel
    uint32_t kernel;      // which kernel takes the compressible launches: kKernelMulti or kKernelTeam
};
enum CompressibleKernel : uint32_t { kKernelMulti = 0, kKernelTeam = 1 };
constexpr size_t kDualTableBytes = 16384 * 2;
constexpr size_t kLaunchStateBytes = 1024;  // select[0..16): launch state; [16..16+3*64): the sample's per-workgroup results
constexpr uint32_t kDualQueueEntries = 1u << 20;  // 4 MiB: 64 GiB of 64 KiB blocks per launch
#ifndef SZ_GROUP_WAVES
#define SZ_GROUP_WAVES 3
#endif
constexpr uint32_t kDualTablesPerGroup = SZ_GROUP_WAVES - 1;  // global tables per workgroup

// dual == nullptr (or too few blocks): one 64-lane workgroup per block, table in LDS.
hipError_t launch_compress_blocks(const void *d_src, uint64_t in_stride, const uint32_t *d_in_len,
                                  uint32_t n_blocks, void *d_dst, uint64_t dst_stride,
                                  uint32_t *d_out_len, uint32_t flags, const DualScratch *dual,
                                  hipStream_t stream);

// Measurement hook (sz_profile_next_launch): the next launch_compress_blocks / launch_compress_stream of THIS host
// thread records single[0] / single[1] right around the launch of compress_blocks_kernel and multi[0] / multi[1]
// right around the persistent kernel's (team or multi-wave), on the launch's stream; null events are skipped.
// One shot: cleared by the launch that used it.
void set_profile_events(hipEvent_t single0, hipEvent_t single1, hipEvent_t multi0, hipEvent_t multi1);
// Disarms the hook: called when sz_compress_blocks_hip / sz_compress_stream_hip return, on EVERY path (argument
// errors, the accelerator-format kernel, which records nothing), so that events handed over for one call are never
// recorded by a later, unrelated one -- possibly after their owner has destroyed them.
void drop_profile_events();

// SZ_FLAG_HW_COMPAT (snappy_hwcompat.hip): the element stream of the reference accelerator's own datapath, one wave per job
constexpr uint32_t kFlagHwCompat = 2u;  // == SZ_FLAG_HW_COMPAT
hipError_t launch_compress_hw_blocks(const void *d_src, uint64_t in_stride, const uint32_t *d_in_len, uint64_t total_len,
                                     uint32_t n_blocks, void *d_dst, uint64_t dst_stride, uint32_t *d_out_len,
                                     uint32_t flags, hipStream_t stream);

hipError_t launch_compress_stream(const void *d_src, uint64_t n, void *d_slots, uint64_t slot_stride,
                                  uint32_t *d_len, uint32_t flags, const DualScratch *dual, hipStream_t stream);

// d_base_from (may be null): read the base from device memory -- the total a pack of the chunk before wrote -- instead.
// host_total (may be null): a word of pinned, device-visible host memory that receives the stream's total as well
// (stored by the scan kernel: visible to the host once an event recorded behind the launch has completed).
hipError_t launch_pack_stream(const void *d_slots, uint64_t slot_stride, const uint32_t *d_len,
                              uint32_t n_blocks, uint64_t base, void *d_packed, uint64_t *d_off,
                              hipStream_t stream, const uint64_t *d_base_from = nullptr, uint64_t *host_total = nullptr);

// n bytes from device memory to a device-visible address of (registered) host memory, by a kernel (no DMA engine)
hipError_t launch_copy_out(void *dst_dev, const void *d_src, uint64_t n, hipStream_t stream);

hipError_t launch_uncompress_streams(const void *d_src, uint64_t src_stride, const uint32_t *d_src_len,
                                     uint32_t n_streams, void *d_dst, uint64_t dst_stride,
                                     uint32_t *d_out_len, int32_t *d_status, hipStream_t stream);

// The fragments of one packed stream (sz_pack_stream_hip layout), decoded in parallel: fragment b is
// d_packed[d_off[b], d_off[b+1]) and lands at d_dst + b * 64 KiB.
hipError_t launch_uncompress_packed(const void *d_packed, const uint64_t *d_off, uint32_t n_blocks,
                                    uint64_t total_ulen, void *d_dst, uint32_t *d_out_len,
                                    int32_t *d_status, hipStream_t stream);

// Fragment index of one raw stream (snappy_stream_index.hip): fills d_off[0 .. n_frag] with the
// compressed offsets of the 64 KiB output boundaries (UINT64_MAX where none was found),
// d_result[0] = 0 if the stream parses cleanly, d_result[1] = 0 if every boundary falls between two
// elements, d_result[2] = 0; launch_count_failed adds the fragments with a non-zero status to [2].
size_t stream_index_scratch_bytes(uint64_t n);
hipError_t launch_stream_index(const void *d_in, uint64_t n, uint32_t body, uint64_t ulen, void *d_scratch,
                               uint64_t *d_off, uint32_t *d_result, hipStream_t stream);
hipError_t launch_count_failed(const int32_t *d_status, uint32_t n_frag, uint32_t *d_result, hipStream_t stream);

}  // namespace szk
