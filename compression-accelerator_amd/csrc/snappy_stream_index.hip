// snappy_stream_index.hip -- finds, in parallel, where the 64 KiB fragments of ONE raw Snappy stream
// begin, so that the stream can be decoded by one wavefront per fragment (snappy_uncompress.hip)
// instead of by a single wavefront.  Used by the host-buffer decompressor (sz_uncompress =
// snappy::Uncompress, tests/swcompression.cc:31) for streams of more than a few hundred KiB.
//
// The format has no index: where an element starts is known only after the element before it.  But
// where an element WOULD end if it started at a given byte is a local question, so:
//   1. segment maps (one wave per 16-64 KiB segment of the compressed bytes): 64 hypothetical parses run
//      at once, one per lane, starting at the first 64 bytes of the segment; per window of 64 bytes
//      every lane decodes "its" tag, pointer jumping (ds_bpermute) turns that into "where does a
//      chain entering the window at lane l leave it, and how many bytes did it produce", and each
//      hypothetical parse takes one such step.  Result per segment: 64 x (exit position, bytes).
//   2. one wave follows the true parse segment by segment through those maps (after a literal that
//      reaches far into a segment it parses that segment's rest itself) and notes where the true
//      parse enters every segment and with how many bytes produced.
//   3. one wave per segment walks its elements from that entry and records the compressed offset of
//      every element that starts on a 64 KiB output boundary; an element that straddles a boundary
//      means the stream was not produced fragment by fragment: the caller then decodes serially.
// Streams of every stock Snappy compressor (64 KiB fragments) qualify.  Nothing here moves payload
// bytes; the bound is instruction issue of the tag decode (about 100 instructions per 64 bytes).
#include <stdlib.h>

#include "wave_utils.h"

namespace szk {

// compressed bytes per segment: 16, 32 or 64 KiB, chosen per stream (small streams want many segments
// for the parallel passes, large ones few for the serial one)
static inline uint32_t seg_shift_for(uint64_t n)
{
    if (const char *e = getenv("SZ_TEST_SEG_SHIFT")) {  // test hook: every segment size on small streams
        const int v = atoi(e);
        if (v >= 14 && v <= 16)
            return (uint32_t)v;
    }
    return n < (64ull << 20) ? 14u : (n < (256ull << 20) ? 15u : 16u);
}
static constexpr uint32_t kBad = 0xffffffffu;

// The element that would start at byte p < n: compressed size including payload, produced bytes;
// ok = false if it cannot be a valid element of a stream of n bytes.  Reads up to 8 bytes at p.
struct Elem {
    uint32_t size, olen;
    bool ok;
};
__device__ __forceinline__ Elem element_at(const uint8_t *__restrict__ in, uint32_t p, uint32_t n)
{
    const uint64_t x = ld64(in + p);
    const uint32_t tag = (uint32_t)x & 0xffu;
    const uint32_t extra = (uint32_t)(x >> 8);
    const uint32_t kind = tag & 3u, l6 = tag >> 2;
    const bool is_lit = kind == 0;
    const uint32_t nb = (is_lit && l6 >= 60) ? l6 - 59u : 0u;
    const uint32_t lenm1 = nb == 0 ? l6 : (extra & (0xffffffffu >> (32u - 8u * nb)));
    const uint32_t olen = is_lit ? lenm1 + 1u : (kind == 1 ? (l6 & 7u) + 4u : l6 + 1u);
    const uint64_t size = is_lit ? (uint64_t)1 + nb + (uint64_t)lenm1 + 1u
                                 : (kind == 1 ? 2u : (kind == 2 ? 3u : 5u));
    Elem e;
    e.ok = (uint64_t)p + size <= n && olen != 0;
    e.size = (uint32_t)size;
    e.olen = olen;
    return e;
}

__device__ __forceinline__ uint32_t lane_shuffle(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src_lane & 63u) << 2), (int)v);
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = lane_shuffle(v, (threadIdx.x & 63u) ^ (uint32_t)d);
        v = o < v ? o : v;
    }
    return v;
}

// Window of 64 bytes at wb: for a chain of elements that enters it at lane l (byte wb + l),
// E = offset from wb at which the chain leaves the window (>= 64), or (< 64) the offset of the first
// element start at or past `limit` (the end of the segment or of the stream: chains stop there, so
// that a segment's exit is exactly the next segment's entry), or kBad; O = bytes produced on the way.
__device__ __forceinline__ void window_map(const uint8_t *__restrict__ in, uint32_t wb, uint32_t limit, uint32_t n,
                                           uint32_t lane, uint32_t &E, uint32_t &O)
{
    const uint32_t p = wb + lane;
    E = lane;  // at or past the limit: stays where it is
    O = 0;
    if (p < limit) {
        const Elem e = element_at(in, p, n);
        E = e.ok ? lane + e.size : kBad;
        O = e.ok ? e.olen : 0u;
    }
    // an element has at least 2 bytes: at most 32 of them in a window, five doublings cover that
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t e2 = lane_shuffle(E, E), o2 = lane_shuffle(O, E);
        if (E < 64 && E != lane) {
            O += o2;
            E = e2;
        }
    }
}

// ---- 1. segment maps
__global__ __launch_bounds__(64) void stream_segment_maps_kernel(const uint8_t *__restrict__ in, uint32_t n,
                                                                 uint32_t n_seg, uint32_t seg_shift,
                                                                 uint2 *__restrict__ maps)
{
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    if (s >= n_seg)
        return;
    const uint32_t seg_base = s * kSeg;
    const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
    uint32_t pos = seg_base + lane, out = 0;  // hypothetical parse number `lane`
    for (;;) {
        const bool active = pos < seg_end;  // (kBad and exits are >= seg_end)
        if (__ballot(active) == 0)
            break;
        const uint32_t wb = uni(wave_min(active ? pos : kBad));
        uint32_t E, O;
        window_map(in, wb, seg_end, n, lane, E, O);
        const uint32_t rel = pos - wb;
        const uint32_t e = lane_shuffle(E, rel), o = lane_shuffle(O, rel);
        if (active && rel < 64) {  // (an element is >= 2 bytes: every parse in the window moves on)
            out += o;
            pos = e == kBad ? kBad : wb + e;
        }
    }
    maps[(size_t)s * 64 + lane] = make_uint2(pos, out);
}

// ---- 2. the true parse, segment by segment
// entries[s] = (position, bytes produced) where the true parse enters segment s, (kBad, 0) if it
// skips the segment inside a literal; result[0] = 0 if the stream parses cleanly to its end.
__global__ __launch_bounds__(64) void stream_resolve_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t body,
                                                            uint32_t ulen, uint32_t n_seg, uint32_t seg_shift,
                                                            const uint2 *__restrict__ maps,
                                                            uint2 *__restrict__ entries, uint32_t *__restrict__ result)
{
    const uint32_t kSeg = 1u << seg_shift;
    // The walk is a chain of dependent look-ups; straight from global memory each costs a memory
    // latency.  The maps of 64 segments (32 KiB) are therefore staged in LDS with one burst of
    // coalesced loads and stepped through from there.
    __shared__ uint2 lmap[64 * 64];
    const uint32_t lane = threadIdx.x;
    uint32_t p = body, out = 0;
    bool ok = true;
    for (uint32_t s0 = 0; s0 < n_seg && ok; s0 += 64) {
        const uint32_t cnt = n_seg - s0 < 64 ? n_seg - s0 : 64;
        bool staged = false;  // on first use: a run of long literals (incompressible data) never looks a map up
        for (uint32_t k = 0; k < cnt && ok; k++) {
            const uint32_t s = s0 + k;
            const uint32_t seg_base = s * kSeg;
            const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
            if (p >= seg_end) {
                if (lane == 0)
                    entries[s] = make_uint2(kBad, 0);
                continue;
            }
            if (lane == 0)
                entries[s] = make_uint2(p, out);
            if (p - seg_base < 64) {
                if (!staged) {
                    const u32x4 *g = reinterpret_cast<const u32x4 *>(maps + (size_t)s0 * 64);
                    u32x4 *l = reinterpret_cast<u32x4 *>(lmap);
                    for (uint32_t i = lane; i < cnt * 32; i += 64)  // 32 x 16 bytes per segment
                        l[i] = g[i];
                    __syncthreads();
                    staged = true;
                }
                const uint2 m = lmap[k * 64 + (p - seg_base)];
                const uint32_t ex = uni(m.x), os = uni(m.y);
                ok = ex != kBad;
                p = ex;
                out += os;
            } else {  // entered far into the segment (after a long literal): parse its rest here
                while (ok && p < seg_end) {
                    uint32_t E, O;
                    window_map(in, p, seg_end, n, lane, E, O);
                    const uint32_t e = lane_get(E, 0), o = lane_get(O, 0);
                    ok = e != kBad;
                    out += o;
                    p += e;
                }
            }
        }
        __syncthreads();
    }
    ok = ok && p == n && out == ulen;
    if (lane == 0)
        result[0] = ok ? 0u : 1u;
}

// ---- 3. fragment boundaries inside each segment
// off[m] := compressed offset of the element that starts at output position m * 64 KiB (0 < m < number
// of fragments); flag |= 2 if an element straddles such a position, |= 1 on a malformed element.
__global__ __launch_bounds__(64) void stream_boundaries_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t ulen,
                                                               uint32_t n_seg, uint32_t seg_shift,
                                                               const uint2 *__restrict__ entries,
                                                               unsigned long long *__restrict__ off,
                                                               uint32_t *__restrict__ flag)
{
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    if (s >= n_seg)
        return;
    const uint2 en = entries[s];
    uint32_t p = uni(en.x), out = uni(en.y);
    if (p == kBad)
        return;
    const uint32_t seg_base = s * kSeg;
    const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
    while (p < seg_end) {
        const uint32_t q = p + lane;
        Elem e;
        e.size = 0;
        e.olen = 0;
        e.ok = true;
        if (q < n)
            e = element_at(in, q, n);
        // chain from lane 0: lane 63 never starts an element of this window and absorbs the walk
        const uint32_t end = lane + e.size;
        const uint32_t nx = (q >= n || !e.ok || end > 62 || lane == 63) ? 63u : end;
        uint64_t Pm = 0;
        uint32_t j = 0;
        do {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                Pm |= 1ull << j;
                j = lane_get(nx, j);
            }
        } while (j != 63);
        Pm &= ~(1ull << 63);
        // only elements that start inside this segment (and inside the stream) are this wave's
        const bool mine = ((Pm >> lane) & 1) && q < seg_end;
        if (__ballot(mine && !e.ok)) {
            if (lane == 0)
                atomicOr(flag, 1u);
            return;
        }
        const uint64_t Mm = __ballot(mine);  // a prefix of the chain (positions increase along it)
        const uint32_t e_last = 63u - (uint32_t)__builtin_clzll(Mm);  // lane 0 is always mine: Mm != 0
        const uint32_t olen = mine ? e.olen : 0u;
        const uint32_t incl = wave_inclusive_sum(olen);
        const uint32_t start = out + (incl - olen);
        if (mine && (start & 0xffffu) == 0 && start != 0 && start < ulen)
            off[start >> 16] = q;
        if (__ballot(mine && (start >> 16) != ((start + olen - 1u) >> 16))) {
            if (lane == 0)
                atomicOr(flag, 2u);
            return;
        }
        out += lane_get(incl, 63);
        p += e_last + lane_get(e.size, e_last);
    }
}

// off[0] = body, off[n_frag] = n, everything between "not found"; result[0..2] = 0
__global__ __launch_bounds__(256) void stream_index_init_kernel(unsigned long long *__restrict__ off, uint32_t n_frag,
                                                                uint32_t body, uint32_t n, uint32_t *__restrict__ result)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i <= n_frag)
        off[i] = i == 0 ? body : (i == n_frag ? n : ~0ull);
    if (i < 3)
        result[i] = 0;
}

// result[2] += number of fragments whose decode failed (status != 0)
__global__ __launch_bounds__(256) void count_failed_kernel(const int32_t *__restrict__ status, uint32_t n_frag,
                                                           uint32_t *__restrict__ result)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool bad = i < n_frag && status[i] != 0;
    const uint64_t m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(result + 2, (uint32_t)__builtin_popcountll(m));
}

// ------------------------------------------------------------------------------- launcher
size_t stream_index_scratch_bytes(uint64_t n)
{
    const uint64_t seg = 1ull << 14;  // sized for the smallest segments (the choice may be overridden in tests)
    const size_t n_seg = (size_t)((n + seg - 1) / seg);
    return n_seg * 64 * sizeof(uint2) + n_seg * sizeof(uint2) + 64;
}

// d_in: the whole raw stream (n < 2^31 bytes, readable up to n + 8), body = size of its preamble,
// ulen = the decoded size it announces, n_frag = ceil(ulen / 64 KiB).  Fills d_off[0 .. n_frag]
// (boundaries that are not found stay UINT64_MAX) and d_result[0..1] = {parse status, boundary
// flags}; d_result[2] := 0 (see launch_count_failed).
hipError_t launch_stream_index(const void *d_in, uint64_t n, uint32_t body, uint64_t ulen, void *d_scratch,
                               uint64_t *d_off, uint32_t *d_result, hipStream_t stream)
{
    const uint32_t seg_shift = seg_shift_for(n);
    const uint32_t n_seg = (uint32_t)((n + (1ull << seg_shift) - 1) >> seg_shift);
    const uint32_t n_frag = (uint32_t)((ulen + kMaxBlock - 1) / kMaxBlock);
    uint2 *maps = static_cast<uint2 *>(d_scratch);
    uint2 *entries = maps + (size_t)n_seg * 64;
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    unsigned long long *off = reinterpret_cast<unsigned long long *>(d_off);
    hipLaunchKernelGGL(stream_index_init_kernel, dim3(n_frag / 256 + 1), dim3(256), 0, stream, off, n_frag, body,
                       (uint32_t)n, d_result);
    hipLaunchKernelGGL(stream_segment_maps_kernel, dim3(n_seg), dim3(64), 0, stream, in, (uint32_t)n, n_seg, seg_shift, maps);
    hipLaunchKernelGGL(stream_resolve_kernel, dim3(1), dim3(64), 0, stream, in, (uint32_t)n, body, (uint32_t)ulen,
                       n_seg, seg_shift, maps, entries, d_result);
    hipLaunchKernelGGL(stream_boundaries_kernel, dim3(n_seg), dim3(64), 0, stream, in, (uint32_t)n, (uint32_t)ulen,
                       n_seg, seg_shift, entries, off, d_result + 1);
    return hipGetLastError();
}

hipError_t launch_count_failed(const int32_t *d_status, uint32_t n_frag, uint32_t *d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(count_failed_kernel, dim3((n_frag + 255) / 256), dim3(256), 0, stream, d_status, n_frag, d_result);
    return hipGetLastError();
}

}  // namespace szk
