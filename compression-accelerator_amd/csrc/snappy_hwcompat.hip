// snappy_hwcompat.hip -- the element stream the reference ACCELERATOR's own datapath forms (SURVEY.md section 8f-4,
// SZ_FLAG_HW_COMPAT), as opposed to software Snappy's (the parity target of everything else in this library).
// Read from the RTL as text (it cannot be simulated here): a 512-entry hash table with present bits, the stored
// word and a 16-bit offset, cleared per job (HashTable.scala:11-54; hash = (word * 0x1e35a7bd >> 23) & 511 of the
// four bytes at the data pointer, first byte in bits 31..24); every position is hashed while at least 4 bytes
// remain, one byte per step, no skip heuristic and no P-1 insertion (MatchFinder.scala:42-43; CompressionAccelerator
// .scala:139,315); on the cycle `start` fires -- the first position of a job and the first one behind every copy
// -- the table is written but no match is reported (MatchFinder.scala:70-73); a match needs present && stored word
// == word && candidate != position (:64-67); literals are closed every 60 bytes with a one-byte tag (forceEmit,
// CompressionAccelerator.scala:172); a copy grows to at most 64 bytes and goes on as further elements with the
// same offset (CopyCompress.scala:49,72,140-144, `continue`); copy elements are 2 bytes for length 4..11 and offset
// < 2048, else 3 (CopyStreamFormer, :236-250).  The bytes are Snappy elements: behind varint32(n) any Snappy decoder
// reads them back.  tests/hw_model.py is the same walk on the CPU (the checker of this kernel); what the RTL reports
// as its LENGTH carries a residue of its write-back bookkeeping on top (tests/test_hw_model.py: 31 of the 36 rows of
// benchmark/hw_results.csv are this stream's size + the size of varint32(n) + 1 if the stream ends with a literal).
// One wavefront per job; the parse is serial (one table look-up per position), the lanes hold the window's words,
// extend matches 64 bytes at a time and move the literals' bytes.  A compatibility mode, not a fast path.
#include "wave_utils.h"
#include "snappy_kernels.h"

namespace szk {

static constexpr uint32_t kHwTable = 512, kHwMul = 0x1e35a7bdu, kHwMaxLiteral = 60, kHwMaxCopy = 64;

__global__ __launch_bounds__(64) void compress_hw_blocks_kernel(const uint8_t *__restrict__ src, uint64_t in_stride,
                                                                const uint32_t *__restrict__ in_len, uint64_t total_len,
                                                                uint32_t n_blocks, uint8_t *__restrict__ dst,
                                                                uint64_t dst_stride, uint32_t *__restrict__ out_len,
                                                                uint32_t flags)
{
    __shared__ uint32_t t_word[kHwTable];
    __shared__ uint16_t t_off[kHwTable];  // 0xffff: not present (a stored offset is at most n - 4 <= 65 532)
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (b >= n_blocks)
        return;
    uint32_t n;
    if (in_len)
        n = uni(in_len[b]);
    else if (total_len) {
        const uint64_t left = total_len - (uint64_t)b * in_stride;
        n = uni((uint32_t)(left < in_stride ? left : in_stride));
    } else
        n = (uint32_t)in_stride;
    if (n > kMaxBlock) {
        out_len[b] = 0xffffffffu;  // (every lane, same address, same value)
        return;
    }
    const uint8_t *__restrict__ in = src + (uint64_t)b * in_stride;
    uint8_t *out = dst + (uint64_t)b * dst_stride;
    for (uint32_t i = lane; i < kHwTable; i += 64)
        t_off[i] = 0xffffu;
    __syncthreads();
    uint32_t op = 0;
    if (flags & kFlagBlockVarint) {  // (so that the slot is a raw stream any Snappy decoder reads; the RTL writes none)
        uint32_t v = n;
        while (v >= 0x80u) {
            if (lane == 0)
                out[op] = (uint8_t)(v | 0x80u);
            op++;
            v >>= 7;
        }
        if (lane == 0)
            out[op] = (uint8_t)v;
        op++;
    }
    uint32_t p = 0, lit_start = 0;
    bool looking = false;
    auto flush = [&](uint32_t upto) {  // the open literal [lit_start, upto): one tag byte + its bytes (<= 60)
        const uint32_t ln = upto - lit_start;
        if (ln) {
            if (lane == 0)
                out[op] = (uint8_t)((ln - 1u) << 2);
            if (lane < ln)
                out[op + 1u + lane] = in[lit_start + lane];
            op += 1u + ln;
        }
        lit_start = upto;
    };
    while (p < n) {
        // the words of the next 64 positions, one per lane (first byte in bits 31..24)
        const uint32_t base = p;
        const uint32_t q = base + lane;
        uint32_t wv = 0;
        if (q + 4u <= n)
            wv = __builtin_bswap32(ld32(in + q));
        for (uint32_t j = 0; j < 64u && p < n; j++) {
            if (n - p < 4u) {
                p++;  // the last bytes are streamed out as literal bytes
            } else {
                const uint32_t w = lane_get(wv, j);
                const uint32_t h = ((w * kHwMul) >> 23) & (kHwTable - 1u);
                const uint32_t to = uni(t_off[h]), tw = uni(t_word[h]);
                const bool hit = looking && to != 0xffffu && tw == w && to != p;
                t_off[h] = (uint16_t)p;  // (every lane, same address, same value)
                t_word[h] = w;
                __syncthreads();  // (single-wave workgroup: orders the table accesses of its lanes)
                if (!hit) {
                    looking = true;
                    p++;
                } else {
                    flush(p);
                    uint32_t m = 4;  // the match grows while the bytes agree, to the end of the job
                    for (;;) {
                        const uint32_t i = p + m + lane;
                        const bool eq = i < n && in[to + m + lane] == in[i];
                        const uint64_t ne = ~ballot64(eq);
                        if (ne) {
                            m += ctz64(ne);
                            break;
                        }
                        m += 64;
                    }
                    const uint32_t off = p - to;
                    uint32_t rest = m;
                    while (rest > 0) {  // elements of at most 64 bytes, same offset (`continue`)
                        const uint32_t ln = rest < kHwMaxCopy ? rest : kHwMaxCopy;
                        const bool two = off < 2048u && ln >= 4u && ln <= 11u;
                        const uint32_t word = two ? (1u | ((ln - 4u) << 2) | ((off >> 8) << 5) | ((off & 0xffu) << 8))
                                                  : (2u | ((ln - 1u) << 2) | (off << 8));
                        const uint32_t nb = two ? 2u : 3u;
                        if (lane < nb)
                            out[op + lane] = (uint8_t)(word >> (8u * lane));
                        op += nb;
                        rest -= ln;
                    }
                    p += m;
                    lit_start = p;
                    looking = false;
                    break;  // (the window's words are of positions the parse has left: a new window at p)
                }
            }
            if (p - lit_start == kHwMaxLiteral)
                flush(p);
        }
    }
    flush(n);
    out_len[b] = op;  // every lane, same address, same value
}

hipError_t launch_compress_hw_blocks(const void *d_src, uint64_t in_stride, const uint32_t *d_in_len, uint64_t total_len,
                                     uint32_t n_blocks, void *d_dst, uint64_t dst_stride, uint32_t *d_out_len,
                                     uint32_t flags, hipStream_t stream)
{
    if (n_blocks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(compress_hw_blocks_kernel, dim3(n_blocks), dim3(64), 0, stream, static_cast<const uint8_t *>(d_src),
                       in_stride, d_in_len, total_len, n_blocks, static_cast<uint8_t *>(d_dst), dst_stride, d_out_len, flags);
    return hipGetLastError();
}

}  // namespace szk
