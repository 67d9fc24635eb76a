// snappy_uncompress.hip -- gfx950 kernel of the Snappy block decompressor (SURVEY.md section 8f-1:
// the UNCOMPRESS command of tests/compression.c:32-39, snappy::Uncompress of tests/swcompression.cc:31).
#include "wave_utils.h"
#include <stdlib.h>
#include <string.h>

namespace szk {

// -------------------------------------------------------------------- decompression (8f-1)

// One wavefront per raw stream.  Elements are parsed serially (the format has no index), 64 compressed
// bytes (a window) at a time; the window's output is assembled in LDS -- one element per lane for literals
// and for copies of bytes written before the window, then the copies that read the window's own output,
// in order -- and stored in whole 16-byte pieces.  Copies of older bytes read back this wave's own earlier
// stores to global memory: same wave, same address path, program order.
// Two ways to name the streams: fixed stride + per-stream length, each a raw stream with its own
// preamble (src_off == nullptr); or the fragments of ONE packed stream, fragment b = bytes
// [src_off[b], src_off[b+1]) of src, a body without preamble that decodes to 64 KiB (the last one to
// the rest of total_ulen) -- the layout sz_pack_stream_hip produces.
// A window of the stream as the parser hands it to the mover.  Per lane (lane l = the compressed byte at ip + l):
// d0 = produced bytes of the element that starts there (0: none does) | 0x100 if it is a literal, aux = the literal's
// header bytes or the copy's offset, o_rel = where its bytes go inside the window's output.
struct WinLane {
    uint32_t d0, aux, o_rel;
};
enum : uint32_t { kWinElems = 0, kWinBigLiteral = 1, kWinEnd = 2 };
struct WinHead {
    uint32_t type, ip, op, total, hdr;  // kWinBigLiteral: total = its length, hdr = its header bytes
    int32_t st;
};
static constexpr uint32_t kHist = 64, kWinCap = 1472;

// ---- the parser's half of a window: where do elements start, what do they produce, is all of it valid?
// ip, op, st advance; x_next / have_next carry the next window's bytes (requested as soon as its start is known).
__device__ __forceinline__ void parse_window(const uint8_t *__restrict__ in, uint32_t n, uint32_t ulen, uint32_t lane,
                                             uint32_t &ip, uint32_t &op, int32_t &st, uint64_t &x_next, bool &have_next,
                                             WinLane &wl, WinHead &wh)
{
    wh.type = kWinEnd;
    wh.ip = ip;
    wh.op = op;
    wh.total = 0;
    wh.hdr = 0;
    wh.st = st;
    wl.d0 = 0;
    wl.aux = 0;
    wl.o_rel = 0;
    if (!(st == 0 && (uint64_t)ip + 72 <= n && n <= 0x7fffffffu))  // (32-bit offset sums below)
        return;
    const uint64_t x = have_next ? x_next : ld64(in + ip + lane);
    have_next = false;
    const uint32_t tag = (uint32_t)x & 0xffu;
    const uint32_t extra = (uint32_t)(x >> 8);
    const uint32_t kind_v = tag & 3u;
    const uint32_t l6 = tag >> 2;
    // compressed size, produced bytes, (literal: header bytes | copy: offset) -- selects, no branches
    const bool is_lit_v = kind_v == 0;
    const uint32_t nb = (is_lit_v && l6 >= 60) ? l6 - 59u : 0u;                  // literal: length bytes
    const uint32_t lenm1 = nb == 0 ? l6 : (extra & (0xffffffffu >> (32u - 8u * nb)));
    const uint32_t c_off = kind_v == 1 ? (((tag >> 5) << 8) | (extra & 0xffu))
                                       : (kind_v == 2 ? (extra & 0xffffu) : extra);
    const uint32_t aux_v = is_lit_v ? 1u + nb : c_off;
    const uint32_t olen_v = is_lit_v ? lenm1 + 1u                                  // wraps to 0 for 2^32: rejected below
                                     : (kind_v == 1 ? (l6 & 7u) + 4u : l6 + 1u);
    const uint32_t size_v = is_lit_v ? aux_v + olen_v                              // may wrap: the bound checks use 64 bits
                                     : (kind_v == 1 ? 2u : (kind_v == 2 ? 3u : 5u));
    // ---- which lanes start an element?  Every lane knows where an element starting at it would
    // end (63 = outside the window, also for a literal that cannot be valid: lane 63 never starts
    // an element of this window, so it can stand for "out" and absorb the walk); the serial part
    // is one v_readlane per element.
    // (32-bit sums: ip + lane + 9 <= n < 2^31 here, aux_v <= 5, and a length is clamped before it is added)
    const uint32_t rem_v = n - ip - lane;  // bytes of the stream from this lane's byte on
    const bool bad_v = is_lit_v && (olen_v == 0 || olen_v > rem_v - aux_v);
    const uint32_t end_v = lane + (is_lit_v ? aux_v + (olen_v < 64u ? olen_v : 64u) : size_v);
    const uint32_t nx_v = (bad_v || end_v > 62 || lane == 63) ? 63u : end_v;
    // The chain 0 -> nx(0) -> nx(nx(0)) ... by pointer doubling: the lanes known to be on it send a mark to
    // where their 2^r-th successor starts (ds_permute; everybody else sends to lane 63), then every lane's
    // pointer is doubled (ds_bpermute).  An element has at least 2 bytes: at most 32 in a window, 5 rounds.
    // (A walk with one v_readlane per element was 3.5 instructions per element, 16 elements a window on text,
    // in a kernel that is bound by instruction issue.)
    uint64_t Pm;
    uint32_t j;
    {
        uint32_t jump = nx_v, on = lane == 0 ? 1u : 0u;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            on |= (uint32_t)__builtin_amdgcn_ds_permute((int)((on ? jump : 63u) << 2), (int)on);
            jump = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jump << 2), (int)jump);
        }
        Pm = __ballot(on != 0) & ~(1ull << 63);
    }
    if (__ballot(bad_v) & Pm) {
        st = wh.st = kStatusCorrupt;
        return;
    }
    // A literal of more than 64 bytes does not go through the window's buffer: the window ends in front of
    // it, and a window that begins with one is that literal alone, input to output by all lanes.
    const uint64_t BL = __ballot(kind_v == 0 && olen_v > 64) & Pm;
    if (BL & 1ull) {
        const uint32_t len = lane_get(olen_v, 0), hdr = lane_get(aux_v, 0);
        if ((uint64_t)op + len > ulen) {
            st = wh.st = kStatusCorrupt;
            return;
        }
        wh.type = kWinBigLiteral;
        wh.total = len;
        wh.hdr = hdr;
        ip += hdr + len;
        op += len;
        return;
    }
    if (BL) {
        j = ctz64(BL);
        Pm &= (1ull << j) - 1ull;
    } else {  // the last element may reach past the window: the next window starts where it really ends
        const uint32_t e_last = 63u - (uint32_t)__builtin_clzll(Pm);
        j = e_last + lane_get(size_v, e_last);  // (sizes of valid elements fit: n < 2^31)
    }
    if ((uint64_t)ip + j + 72 <= n) {
        x_next = ld64(in + ip + j + lane);
        have_next = true;
    }
    // ---- output offsets of the window's elements
    const bool is_elem = (Pm >> lane) & 1;
    const uint32_t olen = is_elem ? olen_v : 0u;
    const uint32_t incl = wave_inclusive_sum(olen);
    const uint32_t o_rel = incl - olen;  // offset inside this window's output
    const uint32_t total = lane_get(incl, 63);
    if ((uint64_t)op + total > ulen || total > kWinCap) {  // (more than kWinCap: not reached, see above)
        st = wh.st = kStatusCorrupt;
        return;
    }
    const bool is_copy = is_elem && kind_v != 0;
    if (__ballot(is_copy && (aux_v == 0 || aux_v > op + o_rel)) != 0) {
        st = wh.st = kStatusCorrupt;
        return;
    }
    wh.type = kWinElems;
    wh.total = total;
    wl.d0 = olen | (kind_v == 0 ? 0x100u : 0u);
    wl.aux = aux_v;
    wl.o_rel = o_rel;
    op += total;
    ip += j;
}

// ---- the mover's half: the window's bytes (wbuf: kHist bytes of history + the window; `in`/`out` as the kernel's)
__device__ __forceinline__ void move_window(const uint8_t *__restrict__ in, uint8_t *out, uint8_t *wbuf, uint32_t lane,
                                            const WinLane &wl, const WinHead &wh)
{
    const uint32_t ip = wh.ip, op = wh.op;
    if (wh.type == kWinBigLiteral) {  // a literal of more than 64 bytes: input to output by all lanes
        wave_copy<6>(out + op, in + ip + wh.hdr, wh.total, lane);
        wbuf[lane] = in[ip + wh.hdr + wh.total - 64 + lane];  // (what a copy right behind it can reach)
        return;
    }
    const uint32_t total = wh.total;
    const uint32_t olen = wl.d0 & 0xffu, aux_v = wl.aux, o_rel = wl.o_rel;
    const uint32_t olen_v = olen;
    const bool is_elem = olen != 0;
    const uint32_t kind_v = (wl.d0 & 0x100u) ? 0u : 1u;  // (0: literal; which kind of copy does not matter here)
    const bool is_copy = is_elem && kind_v != 0;
    // A copy whose source ends before this window's output begins ("far") depends on nothing that
    // is produced here, nor does a literal: those are all done at once, one element per lane, into the
    // window's buffer.  Copies that read this window's own output ("near") follow in order, LDS to LDS:
    // no trip to memory per copy.
    const bool far_copy = is_copy && aux_v >= o_rel + olen;
    const bool near_copy = is_copy && !far_copy;
    const bool small_lit = is_elem && kind_v == 0;
    uint8_t *const win = wbuf + kHist;
    {
        const bool par = far_copy || small_lit;
        const uint8_t *sp = small_lit ? in + ip + lane + aux_v : out + op + o_rel - aux_v;
        uint8_t *dp = win + o_rel;
        const uint32_t mylen = par ? olen : 0u;
        // 16 bytes per step while any element has that many left, then the rest in one step: few
        // predicated memory instructions for the short elements that dominate.
        // (an element is at most 64 bytes: up to four 16-byte pieces, all requested before the first one is stored
        // -- one trip to memory for the window, not one per piece)
        const uint32_t nch = mylen >> 4;
        const uint32_t i = nch << 4;
        if (__ballot(nch >= 1) != 0) {
            u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
            const uint64_t b2 = __ballot(nch >= 2), b3 = __ballot(nch >= 3), b4 = __ballot(nch >= 4);
            if (nch >= 1) v0 = ld128(sp);
            if (b2) {
                if (nch >= 2) v1 = ld128(sp + 16);
                if (b3) {
                    if (nch >= 3) v2 = ld128(sp + 32);
                    if (b4) {
                        if (nch >= 4) v3 = ld128(sp + 48);
                    }
                }
            }
            if (nch >= 1) __builtin_memcpy(dp, &v0, 16);
            if (b2) {
                if (nch >= 2) __builtin_memcpy(dp + 16, &v1, 16);
                if (b3) {
                    if (nch >= 3) __builtin_memcpy(dp + 32, &v2, 16);
                    if (b4) {
                        if (nch >= 4) __builtin_memcpy(dp + 48, &v3, 16);
                    }
                }
            }
        }
        // the remaining r = 0..15 bytes, memcpy style: both ends of the rest with one size (the two
        // pieces overlap in the middle), so every lane is in exactly one of three classes
        const uint32_t r = mylen - i;
        const bool c8 = r >= 8, c4 = r >= 4 && r < 8, c1 = r >= 1 && r < 4;
        uint64_t a8 = 0, b8 = 0;
        uint32_t a4 = 0, b4 = 0;
        uint8_t a1 = 0, m1 = 0, b1 = 0;
        if (c8) { a8 = ld64(sp + i); b8 = ld64(sp + mylen - 8); }
        if (c4) { a4 = ld32(sp + i); b4 = ld32(sp + mylen - 4); }
        if (c1) { a1 = sp[i]; m1 = sp[i + (r >> 1)]; b1 = sp[mylen - 1]; }
        if (c8) { __builtin_memcpy(dp + i, &a8, 8); __builtin_memcpy(dp + mylen - 8, &b8, 8); }
        if (c4) { __builtin_memcpy(dp + i, &a4, 4); __builtin_memcpy(dp + mylen - 4, &b4, 4); }
        if (c1) { dp[i] = a1; dp[i + (r >> 1)] = m1; dp[mylen - 1] = b1; }
    }
    // near copies, in stream order (they may read what the steps above and each other wrote, and up to
    // kHist bytes of what was written before this window)
    for (uint64_t N = __ballot(near_copy); N; N &= N - 1) {
        const uint32_t e = ctz64(N);
        const uint32_t len = lane_get(olen_v, e), off = lane_get(aux_v, e), o = lane_get(o_rel, e);
        // (off >= len: plain copy; else the last `off` bytes repeat -- the division only then)
        const uint32_t k = off >= len ? lane : lane % off;
        if (lane < len)
            win[o + lane] = wbuf[kHist + o - off + k];  // (off < o + len <= o + 64: inside the buffer)
    }
    // the window's bytes go out: 16 per lane and step, the last < 16 one per lane
    for (uint32_t i = 16u * lane; i + 16u <= total; i += 1024u) {
        u32x4 v;
        __builtin_memcpy(&v, win + i, 16);
        __builtin_memcpy(out + op + i, &v, 16);
    }
    {
        const uint32_t t0 = total & ~15u;
        if (lane < total - t0)
            out[op + t0 + lane] = win[t0 + lane];
    }
    {   // the last kHist bytes written so far move to the front of the buffer (read by all lanes, then written)
        const uint8_t hb = wbuf[total + lane];
        wbuf[lane] = hb;
    }
}

// (The one-wave kernel keeps its window loop in one piece: split into parse_window + move_window as the two-wave
// kernel below needs it, it was 3 % slower on large launches -- the halves' hand-over through registers.)
__global__ __launch_bounds__(64) void uncompress_streams_kernel(
    const uint8_t *__restrict__ src, uint64_t src_stride, const uint32_t *__restrict__ src_len,
    const uint64_t *__restrict__ src_off, uint64_t total_ulen,
    uint32_t n_streams, uint8_t *dst, uint64_t dst_stride, uint32_t *__restrict__ out_len,
    int32_t *__restrict__ status)
{
    // Output of one window of elements, assembled in LDS before it goes to memory in whole 16-byte pieces:
    // [0, kHist) the last bytes written before this window (what an overlapping copy at the window's start can
    // reach), [kHist, kHist + kWinCap) this window's elements.  A window holds at most 21 three-byte copies of 64
    // bytes and one more element of <= 64: 1 408 bytes.
    __shared__ __attribute__((aligned(16))) uint8_t wbuf[kHist + kWinCap];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_streams)
        return;
    const uint8_t *__restrict__ in;
    uint8_t *out = dst + (uint64_t)b * dst_stride;
    uint32_t n;
    int32_t st = 0;
    uint32_t ip = 0, op = 0, ulen = 0;
    if (src_off) {
        const uint64_t o0 = src_off[b], o1 = src_off[b + 1];
        in = src + o0;
        const uint64_t len64 = o1 >= o0 ? o1 - o0 : ~0ull;
        n = uni((uint32_t)(len64 < 0x7fffffffull ? len64 : 0x7fffffffull));
        const uint64_t done = (uint64_t)b * kMaxBlock;
        const uint64_t left = total_ulen > done ? total_ulen - done : 0;
        ulen = uni((uint32_t)(left < kMaxBlock ? left : kMaxBlock));
        if (len64 >= 0x7fffffffull || left == 0)
            st = kStatusCorrupt;
        else if ((uint64_t)ulen > dst_stride)
            st = kStatusNoSpace;
    } else {
        in = src + (uint64_t)b * src_stride;
        n = uni(src_len[b]);
    }

    // preamble
    if (!src_off) {
        uint32_t shiftv = 0;
        bool ok = false;
        for (uint32_t i = 0; i < 5 && i < n; i++) {
            const uint32_t c = uni(in[i]);
            if (i == 4 && c > 0x0f)
                break;
            ulen |= (c & 0x7f) << shiftv;
            shiftv += 7;
            if (c < 0x80) {
                ok = true;
                ip = i + 1;
                break;
            }
        }
        if (!ok)
            st = kStatusCorrupt;
        else if ((uint64_t)ulen > dst_stride)
            st = kStatusNoSpace;
    }

    // ---- windowed decode: the tags of 64 candidate element starts are decoded at once (lane l looks
    // at compressed byte ip + l), then the elements that really start in the window are walked with
    // v_readlane -- no dependent tag load per element.  Needs ip + 72 <= n; the tail below is serial.
    uint64_t x_next = 0;     // the next window's bytes, requested as soon as it is known where it begins ...
    bool have_next = false;  // ... so that the load's latency passes under the rest of this window's work
    while (st == 0 && (uint64_t)ip + 72 <= n && n <= 0x7fffffffu) {  // (32-bit offset sums below)
        const uint64_t x = have_next ? x_next : ld64(in + ip + lane);
        have_next = false;
        const uint32_t tag = (uint32_t)x & 0xffu;
        const uint32_t extra = (uint32_t)(x >> 8);
        const uint32_t kind_v = tag & 3u;
        const uint32_t l6 = tag >> 2;
        // compressed size, produced bytes, (literal: header bytes | copy: offset) -- selects, no branches
        const bool is_lit_v = kind_v == 0;
        const uint32_t nb = (is_lit_v && l6 >= 60) ? l6 - 59u : 0u;                  // literal: length bytes
        const uint32_t lenm1 = nb == 0 ? l6 : (extra & (0xffffffffu >> (32u - 8u * nb)));
        const uint32_t c_off = kind_v == 1 ? (((tag >> 5) << 8) | (extra & 0xffu))
                                           : (kind_v == 2 ? (extra & 0xffffu) : extra);
        const uint32_t aux_v = is_lit_v ? 1u + nb : c_off;
        const uint32_t olen_v = is_lit_v ? lenm1 + 1u                                  // wraps to 0 for 2^32: rejected below
                                         : (kind_v == 1 ? (l6 & 7u) + 4u : l6 + 1u);
        const uint32_t size_v = is_lit_v ? aux_v + olen_v                              // may wrap: the bound checks use 64 bits
                                         : (kind_v == 1 ? 2u : (kind_v == 2 ? 3u : 5u));
        // ---- which lanes start an element?  Every lane knows where an element starting at it would
        // end (63 = outside the window, also for a literal that cannot be valid: lane 63 never starts
        // an element of this window, so it can stand for "out" and absorb the walk); the serial part
        // is one v_readlane per element.
        // (32-bit sums: ip + lane + 9 <= n < 2^31 here, aux_v <= 5, and a length is clamped before it is added)
        const uint32_t rem_v = n - ip - lane;  // bytes of the stream from this lane's byte on
        const bool bad_v = is_lit_v && (olen_v == 0 || olen_v > rem_v - aux_v);
        const uint32_t end_v = lane + (is_lit_v ? aux_v + (olen_v < 64u ? olen_v : 64u) : size_v);
        const uint32_t nx_v = (bad_v || end_v > 62 || lane == 63) ? 63u : end_v;
        // The chain 0 -> nx(0) -> nx(nx(0)) ... by pointer doubling: the lanes known to be on it send a mark to
        // where their 2^r-th successor starts (ds_permute; everybody else sends to lane 63), then every lane's
        // pointer is doubled (ds_bpermute).  An element has at least 2 bytes: at most 32 in a window, 5 rounds.
        // (A walk with one v_readlane per element was 3.5 instructions per element, 16 elements a window on text,
        // in a kernel that is bound by instruction issue.)
        uint64_t Pm;
        uint32_t j;
        {
            uint32_t jump = nx_v, on = lane == 0 ? 1u : 0u;
#pragma unroll
            for (int r = 0; r < 5; r++) {
                on |= (uint32_t)__builtin_amdgcn_ds_permute((int)((on ? jump : 63u) << 2), (int)on);
                jump = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jump << 2), (int)jump);
            }
            Pm = __ballot(on != 0) & ~(1ull << 63);
        }
        if (__ballot(bad_v) & Pm) {
            st = kStatusCorrupt;
            break;
        }
        // A literal of more than 64 bytes does not go through the window's buffer: the window ends in front of
        // it, and a window that begins with one is that literal alone, input to output by all lanes.
        const uint64_t BL = __ballot(kind_v == 0 && olen_v > 64) & Pm;
        if (BL & 1ull) {
            const uint32_t len = lane_get(olen_v, 0), hdr = lane_get(aux_v, 0);
            if ((uint64_t)op + len > ulen) {
                st = kStatusCorrupt;
                break;
            }
            wave_copy<6>(out + op, in + ip + hdr, len, lane);
            wbuf[lane] = in[ip + hdr + len - 64 + lane];  // (what a copy right behind it can reach)
            ip += hdr + len;
            op += len;
            continue;
        }
        if (BL) {
            j = ctz64(BL);
            Pm &= (1ull << j) - 1ull;
        } else {  // the last element may reach past the window: the next window starts where it really ends
            const uint32_t e_last = 63u - (uint32_t)__builtin_clzll(Pm);
            j = e_last + lane_get(size_v, e_last);  // (sizes of valid elements fit: n < 2^31)
        }
        if ((uint64_t)ip + j + 72 <= n) {
            x_next = ld64(in + ip + j + lane);
            have_next = true;
        }
        // ---- output offsets of the window's elements
        const bool is_elem = (Pm >> lane) & 1;
        const uint32_t olen = is_elem ? olen_v : 0u;
        const uint32_t incl = wave_inclusive_sum(olen);
        const uint32_t o_rel = incl - olen;  // offset inside this window's output
        const uint32_t total = lane_get(incl, 63);
        if ((uint64_t)op + total > ulen || total > kWinCap) {  // (more than kWinCap: not reached, see above)
            st = kStatusCorrupt;
            break;
        }
        const bool is_copy = is_elem && kind_v != 0;
        if (__ballot(is_copy && (aux_v == 0 || aux_v > op + o_rel)) != 0) {
            st = kStatusCorrupt;
            break;
        }
        // A copy whose source ends before this window's output begins ("far") depends on nothing that
        // is produced here, nor does a literal: those are all done at once, one element per lane, into the
        // window's buffer.  Copies that read this window's own output ("near") follow in order, LDS to LDS:
        // no trip to memory per copy.
        const bool far_copy = is_copy && aux_v >= o_rel + olen;
        const bool near_copy = is_copy && !far_copy;
        const bool small_lit = is_elem && kind_v == 0;
        uint8_t *const win = wbuf + kHist;
        {
            const bool par = far_copy || small_lit;
            const uint8_t *sp = small_lit ? in + ip + lane + aux_v : out + op + o_rel - aux_v;
            uint8_t *dp = win + o_rel;
            const uint32_t mylen = par ? olen : 0u;
            // 16 bytes per step while any element has that many left, then the rest in one step: few
            // predicated memory instructions for the short elements that dominate.
            // (an element is at most 64 bytes: up to four 16-byte pieces, all requested before the first one is stored
            // -- one trip to memory for the window, not one per piece)
            const uint32_t nch = mylen >> 4;
            const uint32_t i = nch << 4;
            if (__ballot(nch >= 1) != 0) {
                u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
                const uint64_t b2 = __ballot(nch >= 2), b3 = __ballot(nch >= 3), b4 = __ballot(nch >= 4);
                if (nch >= 1) v0 = ld128(sp);
                if (b2) {
                    if (nch >= 2) v1 = ld128(sp + 16);
                    if (b3) {
                        if (nch >= 3) v2 = ld128(sp + 32);
                        if (b4) {
                            if (nch >= 4) v3 = ld128(sp + 48);
                        }
                    }
                }
                if (nch >= 1) __builtin_memcpy(dp, &v0, 16);
                if (b2) {
                    if (nch >= 2) __builtin_memcpy(dp + 16, &v1, 16);
                    if (b3) {
                        if (nch >= 3) __builtin_memcpy(dp + 32, &v2, 16);
                        if (b4) {
                            if (nch >= 4) __builtin_memcpy(dp + 48, &v3, 16);
                        }
                    }
                }
            }
            // the remaining r = 0..15 bytes, memcpy style: both ends of the rest with one size (the two
            // pieces overlap in the middle), so every lane is in exactly one of three classes
            const uint32_t r = mylen - i;
            const bool c8 = r >= 8, c4 = r >= 4 && r < 8, c1 = r >= 1 && r < 4;
            uint64_t a8 = 0, b8 = 0;
            uint32_t a4 = 0, b4 = 0;
            uint8_t a1 = 0, m1 = 0, b1 = 0;
            if (c8) { a8 = ld64(sp + i); b8 = ld64(sp + mylen - 8); }
            if (c4) { a4 = ld32(sp + i); b4 = ld32(sp + mylen - 4); }
            if (c1) { a1 = sp[i]; m1 = sp[i + (r >> 1)]; b1 = sp[mylen - 1]; }
            if (c8) { __builtin_memcpy(dp + i, &a8, 8); __builtin_memcpy(dp + mylen - 8, &b8, 8); }
            if (c4) { __builtin_memcpy(dp + i, &a4, 4); __builtin_memcpy(dp + mylen - 4, &b4, 4); }
            if (c1) { dp[i] = a1; dp[i + (r >> 1)] = m1; dp[mylen - 1] = b1; }
        }
        // near copies, in stream order (they may read what the steps above and each other wrote, and up to
        // kHist bytes of what was written before this window)
        for (uint64_t N = __ballot(near_copy); N; N &= N - 1) {
            const uint32_t e = ctz64(N);
            const uint32_t len = lane_get(olen_v, e), off = lane_get(aux_v, e), o = lane_get(o_rel, e);
            // (off >= len: plain copy; else the last `off` bytes repeat -- the division only then)
            const uint32_t k = off >= len ? lane : lane % off;
            if (lane < len)
                win[o + lane] = wbuf[kHist + o - off + k];  // (off < o + len <= o + 64: inside the buffer)
        }
        // the window's bytes go out: 16 per lane and step, the last < 16 one per lane
        for (uint32_t i = 16u * lane; i + 16u <= total; i += 1024u) {
            u32x4 v;
            __builtin_memcpy(&v, win + i, 16);
            __builtin_memcpy(out + op + i, &v, 16);
        }
        {
            const uint32_t t0 = total & ~15u;
            if (lane < total - t0)
                out[op + t0 + lane] = win[t0 + lane];
        }
        {   // the last kHist bytes written so far move to the front of the buffer (read by all lanes, then written)
            const uint8_t hb = wbuf[total + lane];
            wbuf[lane] = hb;
        }
        op += total;
        if (st != 0)
            break;
        ip += j;
    }

    while (st == 0 && ip < n) {
        // tag + up to 4 trailing bytes in one go when they are in bounds
        uint32_t tag, extra = 0;
        if (ip + 8 <= n) {  // (the 8-byte load stays inside the stream: no padding is asked of the caller)
            const uint64_t x = ld64(in + ip);
            tag = uni((uint32_t)x & 0xff);
            extra = uni((uint32_t)(x >> 8));
        } else {
            tag = uni(in[ip]);
            for (uint32_t k = 1; k < 5 && ip + k < n; k++)
                extra |= uni((uint32_t)in[ip + k]) << (8 * (k - 1));
        }
        const uint32_t kind = tag & 3;
        if (kind == 0) {
            uint32_t len = (tag >> 2) + 1, hdr = 1;
            if (len > 60) {
                const uint32_t nb = len - 60;
                if (ip + 1 + nb > n) {
                    st = kStatusCorrupt;
                    break;
                }
                len = (nb == 4 ? extra : (extra & ((1u << (8 * nb)) - 1))) + 1;
                hdr = 1 + nb;
                if (len == 0) {  // 2^32 wrap
                    st = kStatusCorrupt;
                    break;
                }
            }
            if ((uint64_t)ip + hdr + len > n || (uint64_t)op + len > ulen) {
                st = kStatusCorrupt;
                break;
            }
            wave_copy<6>(out + op, in + ip + hdr, len, lane);
            ip += hdr + len;
            op += len;
        } else {
            uint32_t len, off, hdr;
            if (kind == 1) {
                hdr = 2;
                len = ((tag >> 2) & 7) + 4;
                off = ((tag >> 5) << 8) | (extra & 0xff);
            } else if (kind == 2) {
                hdr = 3;
                len = (tag >> 2) + 1;
                off = extra & 0xffff;
            } else {
                hdr = 5;
                len = (tag >> 2) + 1;
                off = extra;
            }
            if (ip + hdr > n || off == 0 || off > op || (uint64_t)op + len > ulen) {
                st = kStatusCorrupt;
                break;
            }
            // len <= 64: one byte per lane; an overlapping copy (off < len) repeats the last off
            // bytes, so lane i reads byte (i mod off) of the window [op-off, op).
            const uint32_t k = off >= len ? lane : lane % off;
            if (lane < len)
                out[op + lane] = out[op - off + k];
            ip += hdr;
            op += len;
        }
    }
    if (st == 0 && op != ulen)
        st = kStatusCorrupt;
    if (lane == 0) {
        out_len[b] = st == 0 ? op : 0;
        status[b] = st;
    }
}

__global__ __launch_bounds__(128) void uncompress_streams_pair_kernel(
    const uint8_t *__restrict__ src, uint64_t src_stride, const uint32_t *__restrict__ src_len,
    const uint64_t *__restrict__ src_off, uint64_t total_ulen,
    uint32_t n_streams, uint8_t *dst, uint64_t dst_stride, uint32_t *__restrict__ out_len,
    int32_t *__restrict__ status)
{
    // Output of one window of elements, assembled in LDS before it goes to memory in whole 16-byte pieces:
    // [0, kHist) the last bytes written before this window (what an overlapping copy at the window's start can
    // reach), [kHist, kHist + kWinCap) this window's elements.  A window holds at most 21 three-byte copies of 64
    // bytes and one more element of <= 64: 1 408 bytes.
    __shared__ __attribute__((aligned(16))) uint8_t wbuf[kHist + kWinCap];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = uni(threadIdx.x >> 6);
    __shared__ WinLane ring_lanes[2][64];
    __shared__ WinHead ring_head[2];
    const uint32_t b = blockIdx.x;
    if (b >= n_streams)
        return;
    const uint8_t *__restrict__ in;
    uint8_t *out = dst + (uint64_t)b * dst_stride;
    uint32_t n;
    int32_t st = 0;
    uint32_t ip = 0, op = 0, ulen = 0;
    if (src_off) {
        const uint64_t o0 = src_off[b], o1 = src_off[b + 1];
        in = src + o0;
        const uint64_t len64 = o1 >= o0 ? o1 - o0 : ~0ull;
        n = uni((uint32_t)(len64 < 0x7fffffffull ? len64 : 0x7fffffffull));
        const uint64_t done = (uint64_t)b * kMaxBlock;
        const uint64_t left = total_ulen > done ? total_ulen - done : 0;
        ulen = uni((uint32_t)(left < kMaxBlock ? left : kMaxBlock));
        if (len64 >= 0x7fffffffull || left == 0)
            st = kStatusCorrupt;
        else if ((uint64_t)ulen > dst_stride)
            st = kStatusNoSpace;
    } else {
        in = src + (uint64_t)b * src_stride;
        n = uni(src_len[b]);
    }

    // preamble
    if (!src_off) {
        uint32_t shiftv = 0;
        bool ok = false;
        for (uint32_t i = 0; i < 5 && i < n; i++) {
            const uint32_t c = uni(in[i]);
            if (i == 4 && c > 0x0f)
                break;
            ulen |= (c & 0x7f) << shiftv;
            shiftv += 7;
            if (c < 0x80) {
                ok = true;
                ip = i + 1;
                break;
            }
        }
        if (!ok)
            st = kStatusCorrupt;
        else if ((uint64_t)ulen > dst_stride)
            st = kStatusNoSpace;
    }

    // ---- two waves per stream: wave 0 parses window k + 1 (nothing it needs depends on the bytes of window k) while
    // wave 1 moves window k; one s_barrier per window, descriptors in two LDS slots.  A stream's time is its windows
    // one after the other: this halves it where there are too few streams to fill the GPU.
    if (wave == 0) {
        uint64_t x_next = 0;
        bool have_next = false;
        for (uint32_t k = 0;; k ^= 1u) {
            WinLane wl;
            WinHead wh;
            parse_window(in, n, ulen, lane, ip, op, st, x_next, have_next, wl, wh);
            ring_lanes[k][lane] = wl;
            if (lane == 0)
                ring_head[k] = wh;
            __syncthreads();
            if (wh.type == kWinEnd)
                return;
        }
    }
    for (uint32_t k = 0;; k ^= 1u) {
        __syncthreads();
        const WinLane wl = ring_lanes[k][lane];
        WinHead wh;
        wh.type = uni(ring_head[k].type);
        wh.ip = uni(ring_head[k].ip);
        wh.op = uni(ring_head[k].op);
        wh.total = uni(ring_head[k].total);
        wh.hdr = uni(ring_head[k].hdr);
        wh.st = (int32_t)uni((uint32_t)ring_head[k].st);
        if (wh.type == kWinEnd) {  // where the windows end: the serial tail and the verdict are this wave's
            ip = wh.ip;
            op = wh.op;
            st = wh.st;
            break;
        }
        move_window(in, out, wbuf, lane, wl, wh);
    }

    while (st == 0 && ip < n) {
        // tag + up to 4 trailing bytes in one go when they are in bounds
        uint32_t tag, extra = 0;
        if (ip + 8 <= n) {  // (the 8-byte load stays inside the stream: no padding is asked of the caller)
            const uint64_t x = ld64(in + ip);
            tag = uni((uint32_t)x & 0xff);
            extra = uni((uint32_t)(x >> 8));
        } else {
            tag = uni(in[ip]);
            for (uint32_t k = 1; k < 5 && ip + k < n; k++)
                extra |= uni((uint32_t)in[ip + k]) << (8 * (k - 1));
        }
        const uint32_t kind = tag & 3;
        if (kind == 0) {
            uint32_t len = (tag >> 2) + 1, hdr = 1;
            if (len > 60) {
                const uint32_t nb = len - 60;
                if (ip + 1 + nb > n) {
                    st = kStatusCorrupt;
                    break;
                }
                len = (nb == 4 ? extra : (extra & ((1u << (8 * nb)) - 1))) + 1;
                hdr = 1 + nb;
                if (len == 0) {  // 2^32 wrap
                    st = kStatusCorrupt;
                    break;
                }
            }
            if ((uint64_t)ip + hdr + len > n || (uint64_t)op + len > ulen) {
                st = kStatusCorrupt;
                break;
            }
            wave_copy<6>(out + op, in + ip + hdr, len, lane);
            ip += hdr + len;
            op += len;
        } else {
            uint32_t len, off, hdr;
            if (kind == 1) {
                hdr = 2;
                len = ((tag >> 2) & 7) + 4;
                off = ((tag >> 5) << 8) | (extra & 0xff);
            } else if (kind == 2) {
                hdr = 3;
                len = (tag >> 2) + 1;
                off = extra & 0xffff;
            } else {
                hdr = 5;
                len = (tag >> 2) + 1;
                off = extra;
            }
            if (ip + hdr > n || off == 0 || off > op || (uint64_t)op + len > ulen) {
                st = kStatusCorrupt;
                break;
            }
            // len <= 64: one byte per lane; an overlapping copy (off < len) repeats the last off
            // bytes, so lane i reads byte (i mod off) of the window [op-off, op).
            const uint32_t k = off >= len ? lane : lane % off;
            if (lane < len)
                out[op + lane] = out[op - off + k];
            ip += hdr;
            op += len;
        }
    }
    if (st == 0 && op != ulen)
        st = kStatusCorrupt;
    if (lane == 0) {
        out_len[b] = st == 0 ? op : 0;
        status[b] = st;
    }
}

// Which kernel decodes a launch: one wave per stream, or -- launches that leave the GPU's wave slots mostly empty --
// two (parser + mover: a stream's windows in half the time).  SZ_DECODE_KERNEL=single|pair forces one (tests, A/B).
static bool use_pair_kernel(uint32_t n_streams)
{
    const char *e = getenv("SZ_DECODE_KERNEL");  // (read per call, like SZ_KERNEL)
    if (e && !strcmp(e, "pair"))
        return true;
    if (e && !strcmp(e, "single"))
        return false;
    return n_streams <= 4096;  // (8 192 wave slots at eight waves per SIMD: both waves of every stream are resident)
}

hipError_t launch_uncompress_streams(const void *d_src, uint64_t src_stride, const uint32_t *d_src_len,
                                     uint32_t n_streams, void *d_dst, uint64_t dst_stride,
                                     uint32_t *d_out_len, int32_t *d_status, hipStream_t stream)
{
    if (n_streams == 0)
        return hipSuccess;
    if (use_pair_kernel(n_streams))
        hipLaunchKernelGGL(uncompress_streams_pair_kernel, dim3(n_streams), dim3(128), 0, stream,
                           static_cast<const uint8_t *>(d_src), src_stride, d_src_len,
                           static_cast<const uint64_t *>(nullptr), (uint64_t)0, n_streams,
                           static_cast<uint8_t *>(d_dst), dst_stride, d_out_len, d_status);
    else
        hipLaunchKernelGGL(uncompress_streams_kernel, dim3(n_streams), dim3(64), 0, stream,
                           static_cast<const uint8_t *>(d_src), src_stride, d_src_len,
                           static_cast<const uint64_t *>(nullptr), (uint64_t)0, n_streams,
                           static_cast<uint8_t *>(d_dst), dst_stride, d_out_len, d_status);
    return hipGetLastError();
}

hipError_t launch_uncompress_packed(const void *d_packed, const uint64_t *d_off, uint32_t n_blocks,
                                    uint64_t total_ulen, void *d_dst, uint32_t *d_out_len,
                                    int32_t *d_status, hipStream_t stream)
{
    if (n_blocks == 0)
        return hipSuccess;
    if (use_pair_kernel(n_blocks))
        hipLaunchKernelGGL(uncompress_streams_pair_kernel, dim3(n_blocks), dim3(128), 0, stream,
                           static_cast<const uint8_t *>(d_packed), (uint64_t)0,
                           static_cast<const uint32_t *>(nullptr), d_off, total_ulen, n_blocks,
                           static_cast<uint8_t *>(d_dst), (uint64_t)kMaxBlock, d_out_len, d_status);
    else
        hipLaunchKernelGGL(uncompress_streams_kernel, dim3(n_blocks), dim3(64), 0, stream,
                           static_cast<const uint8_t *>(d_packed), (uint64_t)0,
                           static_cast<const uint32_t *>(nullptr), d_off, total_ulen, n_blocks,
                           static_cast<uint8_t *>(d_dst), (uint64_t)kMaxBlock, d_out_len, d_status);
    return hipGetLastError();
}

}  // namespace szk
