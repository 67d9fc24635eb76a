// snappy_pack.hip -- turns the per-block slots of the compressor into the single stream that
// snappy::Compress emits: exclusive scan of the lengths + gather.
#include "wave_utils.h"

namespace szk {

// ---------------------------------------------------------------------- stream packing (8f-2)

// Exclusive prefix sum of n lengths (+base) by one 1024-thread workgroup.
// `base_from` (may be null): the base is read from device memory instead -- the total of the chunk before this one
// (its off[n]), so that the chunks of one stream are packed one behind the other without the host in between
// (sz_compress's pipeline).  A base of UINT64_MAX (that chunk held an error marker) poisons this chunk too: every
// offset becomes UINT64_MAX, which the gather below skips.
__global__ __launch_bounds__(1024) void scan_lengths_kernel(const uint32_t *__restrict__ len, uint32_t n,
                                                            uint64_t base, const uint64_t *__restrict__ base_from,
                                                            uint64_t max_len, uint64_t *__restrict__ off,
                                                            uint64_t *__restrict__ host_total)
{
    __shared__ uint64_t part[1024];
    __shared__ uint32_t bad;
    const uint32_t t = threadIdx.x;
    if (t == 0)
        bad = 0;
    if (base_from)
        base = *base_from;
    const bool poisoned = base == ~0ull;
    __syncthreads();
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = t * per < n ? t * per : n;
    const uint32_t hi = lo + per < n ? lo + per : n;
    // a length above the slot size is one of the compress kernels' error markers (0xffffffff: block
    // longer than 64 KiB, 0xfffffffe: internal): it counts as 0 and the total becomes UINT64_MAX
    uint64_t s = 0;
    bool any_bad = false;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = len[i];
        any_bad |= v > max_len;
        s += v > max_len ? 0u : v;
    }
    if (any_bad)
        atomicOr(&bad, 1u);
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        uint64_t add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint64_t run = base + (t ? part[t - 1] : 0);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = len[i];
        off[i] = poisoned ? ~0ull : run;
        run += v > max_len ? 0u : v;
    }
    if (t == 1023) {
        const uint64_t total = (bad || poisoned) ? ~0ull : base + part[1023];
        off[n] = total;
        // (sz_compress's pipeline) ... and to a word of pinned host memory, by this store, not by a copy behind the kernel:
        // an 8-byte copy is a job for a DMA engine, the runtime gives it whichever engine is free when it is submitted, and
        // when that was the engine with the call's whole input queued on it the total arrived behind the last input byte
        // -- one call in three took 75 ms instead of 47 (EXPERIMENTS.md R6.9)
        if (host_total)
            __hip_atomic_store(host_total, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void gather_slots_kernel(const uint8_t *__restrict__ slots,
                                                           uint64_t slot_stride,
                                                           const uint32_t *__restrict__ len,
                                                           const uint64_t *__restrict__ off,
                                                           uint8_t *__restrict__ packed)
{
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    const uint8_t *s = slots + (uint64_t)b * slot_stride;
    const uint64_t o = off[b];
    uint8_t *d = packed + o;
    uint32_t n = len[b];
    if (n > slot_stride || o == ~0ull)  // error marker (here, or in a chunk before this one): nothing to move (the scan reports it)
        return;
    uint32_t head = (16u - (uint32_t)((uintptr_t)d & 15u)) & 15u;
    if (head > n)
        head = n;
    if (t < head)
        d[t] = s[t];
    d += head;
    s += head;
    n -= head;
    const uint32_t nvec = n >> 4;
    for (uint32_t i = t; i < nvec; i += 256)
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)i) = ld128(s + 16 * (size_t)i);
    const uint32_t done = nvec << 4;
    if (t < (n & 15u))
        d[done + t] = s[done + t];
}

// A stretch of the packed stream written straight into the caller's (registered, device-visible) host buffer by a kernel
// instead of a DMA engine: 16-byte stores on dst-aligned addresses, a wave's store = 1 KiB in one piece over the link.
// (sz_compress's pipeline: with both directions on the DMA engines the runtime now and then put them on ONE engine --
// its choice, made anew after every synchronisation -- and the copy out waited for the whole copy in; EXPERIMENTS.md R6.9.)
__global__ __launch_bounds__(256) void copy_out_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n)
{
    const uint64_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
    const uint64_t h = head < n ? head : n;
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)gridDim.x * 256;
    if (tid < h)
        dst[tid] = src[tid];
    const uint64_t nvec = (n - h) >> 4;
    uint8_t *d = dst + h;
    const uint8_t *s = src + h;
    for (uint64_t i = tid; i < nvec; i += nthreads)
        __builtin_nontemporal_store(ld128(s + 16 * i), reinterpret_cast<u32x4 *>(d + 16 * i));
    const uint64_t done = nvec << 4, tail = (n - h) & 15u;
    if (tid < tail)
        d[done + tid] = s[done + tid];
}

hipError_t launch_copy_out(void *dst_dev, const void *d_src, uint64_t n, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t want = (n / 16 + 255) / 256 + 1;
    const uint32_t grid = (uint32_t)(want < 128 ? want : 128);  // (32 ... 2 048 workgroups: 51.3-53.0 ms per 2 GiB call, the link's business)
    hipLaunchKernelGGL(copy_out_kernel, dim3(grid), dim3(256), 0, stream, static_cast<uint8_t *>(dst_dev),
                       static_cast<const uint8_t *>(d_src), n);
    return hipGetLastError();
}

hipError_t launch_pack_stream(const void *d_slots, uint64_t slot_stride, const uint32_t *d_len,
                              uint32_t n_blocks, uint64_t base, void *d_packed, uint64_t *d_off,
                              hipStream_t stream, const uint64_t *d_base_from, uint64_t *host_total)
{
    hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, stream, d_len, n_blocks, base, d_base_from, slot_stride, d_off,
                       host_total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || n_blocks == 0)
        return e;
    hipLaunchKernelGGL(gather_slots_kernel, dim3(n_blocks), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(d_slots), slot_stride, d_len, d_off,
                       static_cast<uint8_t *>(d_packed));
    return hipGetLastError();
}

}  // namespace szk
