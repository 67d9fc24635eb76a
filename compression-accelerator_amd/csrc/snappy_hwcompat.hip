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
// One wavefront per job, 64 positions per step (every lane one position: its word, its hash, the entry it meets); a
// step ends at the first position that matches.  A compatibility mode, not a fast path.
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
    // one entry = (stored word, offset) in 8 bytes: one LDS read and one write per position.
    // offset 0xffffffff: not present (a stored offset is at most n - 4)
    __shared__ __attribute__((aligned(8))) uint2 table[kHwTable];
    __shared__ uint8_t mark[kHwTable];  // per bucket: a lane of the current window that has it (see the window loop)
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
        table[i] = make_uint2(0u, 0xffffffffu);
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
    const uint64_t below_me = (1ull << lane) - 1ull;  // the lanes below this one
    // 64 positions at a time.  The datapath visits every position up to its next match and writes the table at each
    // (the match position too): lane l is position p + l, the entry it meets is the table's -- or, if an earlier
    // position of the window has its bucket, that position's (it was visited: everything in front of the first
    // match is); the first lane that meets a match ends the window, the lanes up to it commit their entries (per
    // bucket the last of them), the literal bytes in front of it join the open literal (closed every 60 bytes).
    while (p < n) {
        const uint32_t cnt = p + 4u <= n ? (n - 3u - p < 64u ? n - 3u - p : 64u) : 0u;  // positions with 4 bytes left
        if (cnt == 0) {
            p = n;  // the last bytes are streamed out as literal bytes
        } else {
            const bool act = lane < cnt;
            uint32_t w = 0, h = 0;
            uint2 ent = make_uint2(0u, 0xffffffffu);
            if (act) {
                w = __builtin_bswap32(ld32(in + p + lane));
                h = ((w * kHwMul) >> 23) & (kHwTable - 1u);
                ent = table[h];
                mark[h] = (uint8_t)lane;
            }
            __syncthreads();  // (single-wave workgroup: orders the LDS accesses of its lanes)
            uint64_t group = 1ull << lane;  // the lanes of this window with this lane's bucket
            {
                const uint32_t r = act ? mark[h] : lane;
                uint64_t losers = ballot64(r != lane);
                while (losers) {  // (one trip per bucket that two or more positions of the window share)
                    const uint32_t hv = lane_get(h, ctz64(losers));
                    const bool mine = act && h == hv;
                    const uint64_t G = ballot64(mine);
                    losers &= ~G;
                    if (mine)
                        group = G;
                }
            }
            const uint64_t earlier = group & below_me;
            const bool has_pred = earlier != 0;
            const uint32_t pred = has_pred ? 63u - (uint32_t)__builtin_clzll(earlier) : lane;
            const uint32_t w_pred = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pred << 2), (int)w);
            const uint32_t seen_w = has_pred ? w_pred : ent.x;
            const uint32_t seen_o = has_pred ? p + pred : ent.y;
            const bool hit = act && seen_o != 0xffffffffu && seen_w == w && (lane != 0 || looking);
            const uint64_t H = ballot64(hit);
            const uint32_t j0 = H ? ctz64(H) : cnt;  // positions [0, j0) are literal bytes; position j0 (if any) matches
            const uint32_t n_vis = H ? j0 + 1u : cnt;  // positions visited: each writes the table
            {
                const uint64_t vis = n_vis >= 64u ? ~0ull : (1ull << n_vis) - 1ull;
                const uint64_t later = group & vis & ~below_me & ~(1ull << lane);  // visited positions of my bucket behind me
                if (lane < n_vis && later == 0)
                    table[h] = make_uint2(w, p + lane);
            }
            __syncthreads();
            const uint32_t pos = p + j0;
            while (pos - lit_start >= kHwMaxLiteral)
                flush(lit_start + kHwMaxLiteral);
            if (!H) {
                p = pos;  // (= p + cnt)
                looking = true;
            } else {
                const uint32_t to = lane_get(seen_o, j0);
                flush(pos);
                uint32_t m = 4;  // the match grows while the bytes agree, to the end of the job
                for (;;) {
                    const uint32_t i = pos + m + lane;
                    const bool eq = i < n && in[to + m + lane] == in[i];
                    const uint64_t ne = ~ballot64(eq);
                    if (ne) {
                        m += ctz64(ne);
                        break;
                    }
                    m += 64;
                }
                const uint32_t off = pos - to;
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
                p = pos + m;
                lit_start = p;
                looking = false;
            }
        }
        while (p - lit_start >= kHwMaxLiteral)
            flush(lit_start + kHwMaxLiteral);
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
