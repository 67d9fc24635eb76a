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
//   2. the maps of 64 consecutive segments are composed into chunk maps (in parallel), one wave follows
//      the true parse from chunk to chunk, and the chunks' segments get their entries in parallel again
//      (details at the kernels: a chunk that the parse enters far from a segment's start is walked by
//      that one wave, runs of equal-sized elements 64 at a time).
//   3. one wave per segment follows the parse from that entry until it stands on a checkpoint that pass 1
//      left (one per KiB, of one of its hypothetical parses), looks every 64 KiB output boundary of the
//      segment up among the checkpoints and walks only the KiB in front of it; it records the compressed
//      offset of every element that starts on a boundary; an element that straddles a boundary
//      means the stream was not produced fragment by fragment: the caller then decodes serially.
// Streams of every stock Snappy compressor (64 KiB fragments) qualify.  Nothing here moves payload
// bytes; the bound is instruction issue of the tag decode (about 100 instructions per 64 bytes).
#include <stdlib.h>

#include "wave_utils.h"
#if defined(SZ_STREAM_DEBUG) || defined(SZ_DEBUG_SYNC)
#include <stdio.h>  // (diagnostic builds only: tools/streamdbg.py, the launch-by-launch synchronisation)
#endif

namespace szk {

// compressed bytes per segment: 4 to 64 KiB, chosen per stream (small streams want many segments for the
// parallel passes -- a wave walks its segment window by window --, large ones fewer)
static inline uint32_t seg_shift_for(uint64_t n)
{
    if (const char *e = getenv("SZ_TEST_SEG_SHIFT")) {  // test hook: every segment size on small streams
        const int v = atoi(e);
        if (v >= 12 && v <= 16)
            return (uint32_t)v;
    }
    // (measured on text streams after the serial pass stopped caring about the number of segments -- EXPERIMENTS.md
    // R3.3: 7 MB of stream 0.77 ms at 16 KiB segments, 0.65 at 4 KiB; 232 MB best at 16 KiB; 464 MB at 32 KiB)
    return n < (24ull << 20) ? 12u : (n < (128ull << 20) ? 13u : (n < (320ull << 20) ? 14u : (n < (1024ull << 20) ? 15u : 16u)));
}
static constexpr uint32_t kBad = 0xffffffffu;

// The element that would start at byte p < n: compressed size including payload, produced bytes;
// ok = false if it cannot be a valid element of a stream of n bytes.  Reads up to 8 bytes at p.
struct Elem {
    uint32_t size, olen;
    bool ok;
};
__device__ __forceinline__ Elem element_of(uint64_t x, uint32_t p, uint32_t n)  // x = the 8 bytes at p
{
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
__device__ __forceinline__ Elem element_at(const uint8_t *__restrict__ in, uint32_t p, uint32_t n)
{
    return element_of(ld64(in + p), p, n);
}

__device__ __forceinline__ uint32_t lane_shuffle(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src_lane & 63u) << 2), (int)v);
}

// minimum over the 64 lanes (every lane gets it): DPP inside the rows and across them -- six vector
// instructions and one v_readlane instead of six dependent trips through the LDS crossbar (ds_bpermute: ~57
// clocks each, tools/isacost.hip); lanes a DPP step has no source for keep the identity
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#define SZ_DPP_MIN(ctrl, rmask)                                                                                   \
    {                                                                                                             \
        const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, (ctrl), (rmask), 0xf, false); \
        v = t_ < v ? t_ : v;                                                                                      \
    }
    SZ_DPP_MIN(0x111, 0xf)  // row_shr:1
    SZ_DPP_MIN(0x112, 0xf)  // row_shr:2
    SZ_DPP_MIN(0x114, 0xf)  // row_shr:4
    SZ_DPP_MIN(0x118, 0xf)  // row_shr:8  -> lane 15 of every row holds the row's minimum
    SZ_DPP_MIN(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    SZ_DPP_MIN(0x143, 0xc)  // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's minimum
#undef SZ_DPP_MIN
    return lane_get(v, 63);
}

// Window of 64 bytes at wb: for a chain of elements that enters it at lane l (byte wb + l),
// E = offset from wb at which the chain leaves the window (>= 64), or (< 64) the offset of the first
// element start at or past `limit` (the end of the segment or of the stream: chains stop there, so
// that a segment's exit is exactly the next segment's entry), or kBad; O = bytes produced on the way.
// (`bytes` = the window's bytes where they are read from: `in + wb`, or a copy of that stretch of the stream in LDS)
template <typename BytePtr>
__device__ __forceinline__ void window_map_from(BytePtr bytes, uint32_t wb, uint32_t limit, uint32_t n,
                                                uint32_t lane, uint32_t &E, uint32_t &O)
{
    const uint32_t p = wb + lane;
    E = lane;  // at or past the limit: stays where it is
    O = 0;
    if (p < limit) {
        uint64_t x;
        __builtin_memcpy(&x, bytes + lane, 8);
        const Elem e = element_of(x, p, n);
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
__device__ __forceinline__ void window_map(const uint8_t *__restrict__ in, uint32_t wb, uint32_t limit, uint32_t n,
                                           uint32_t lane, uint32_t &E, uint32_t &O)
{
    window_map_from(in + wb, wb, limit, n, lane, E, O);
}

// ---- 1. segment maps
// Besides the 64 entry -> exit maps, the chain of hypothesis 0 leaves CHECKPOINTS: for every KiB mark of
// the segment the first element start at or past it, with the bytes produced up to there.  Parses that
// enter a segment anywhere fall into step with that chain within a few elements (the format
// resynchronises), so a parse that enters FAR from the segment's start (pass 2b) only has to be followed
// element by element until it stands on a checkpoint: from there on it is hypothesis 0.
static constexpr uint32_t kChkShift = 10;  // one checkpoint per KiB of compressed bytes
__global__ __launch_bounds__(64) void stream_segment_maps_kernel(const uint8_t *__restrict__ in, uint32_t n,
                                                                 uint32_t n_seg, uint32_t seg_shift,
                                                                 uint2 *__restrict__ maps, uint2 *__restrict__ chk,
                                                                 const uint32_t *__restrict__ covered)
{
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    if (s >= n_seg || *covered)
        return;
    const uint32_t seg_base = s * kSeg;
    const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
    const uint32_t n_chk = kSeg >> kChkShift;
    uint2 *my_chk = chk + (size_t)s * n_chk;
    for (uint32_t i = lane; i < n_chk; i += 64)
        my_chk[i] = make_uint2(kBad, 0);
    uint32_t pos = seg_base + lane, out = 0;  // hypothetical parse number `lane`
    uint32_t next_mark = 0;                   // the next KiB mark without a checkpoint
    uint32_t ref = 0;                         // the hypothesis whose checkpoints are being recorded
    // Within a few windows the parses that matter stand on ONE position and move together; the loop is bound by
    // instruction issue (seven waves per SIMD, ~150 instructions per window), so that common case skips what it
    // does not need: `pk` = the hindmost position when it is known without a reduction over the lanes (only the
    // parses there moved in the window before, and nobody lies below where they went), `ref_here` = the reference
    // is one of them -- then no mark is reached before (pk - seg_base) >> kChkShift says so.
    uint32_t pk = kBad;
    bool ref_here = false;
    for (;;) {
        const bool active = pos < seg_end;  // (kBad and exits are >= seg_end)
        if (__ballot(active) == 0)
            break;
        const uint32_t wb = pk != kBad ? pk : uni(wave_min(active ? pos : kBad));
        if (!(pk != kBad && ref_here && ((wb - seg_base) >> kChkShift) < next_mark)) {
            // Checkpoints of ONE hypothesis, the reference: at every KiB mark the first of its element starts that
            // this loop sees at or past the mark, the bytes it has produced by then, and its number (bits 26..31:
            // a segment produces < 64 MiB).  Hypothesis 0 is the reference while it lives; a hypothesis that runs
            // away (a stretch of payload read as a literal of megabytes leaves the segment at once: ~1 % of text
            // segments for any given one) is replaced by the lowest one still inside the segment.  Parses fall
            // into step within a few elements, so the reference is the true parse from its first or second
            // checkpoint on; a reader uses the checkpoints of one number only.
            const uint64_t live = __ballot(active);
            if (!((live >> ref) & 1))
                ref = (uint32_t)__builtin_ctzll(live);  // (live != 0 here)
            uint32_t rpos = lane_get(pos, ref);
            uint32_t mk = (rpos - seg_base) >> kChkShift;
            bool hold = false;
            if (mk >= next_mark && rpos >= wb + 64) {
                // the reference is about to leave a checkpoint far ahead of the hindmost parse: one of the two is
                // reading payload as elements (a fake literal of kilobytes that stays inside the segment, or the
                // parses that started inside a real one).  The one more hypotheses have joined is the better
                // reference either way: where it is wrong, the true parse is one long element and needs no checkpoint.
                // Until more parses stand at the hindmost position than with the reference, nothing is recorded: the
                // hindmost ones reach the marks in between later, and early in a segment -- 64 parses, none joined yet
                // -- a reference that read payload as a literal of 27 KiB would otherwise leave its end as the
                // checkpoint of 27 marks (one text segment in a hundred: its boundaries pass then walked 455 windows).
                const uint64_t hind = __ballot(active && pos == wb);
                if (__builtin_popcountll(hind) > __builtin_popcountll(__ballot(active && pos == rpos))) {
                    ref = (uint32_t)__builtin_ctzll(hind);
                    rpos = wb;
                    mk = (rpos - seg_base) >> kChkShift;
                } else {
                    hold = true;
                }
            }
            if (mk >= next_mark && !hold) {
                const uint32_t rout = lane_get(out, ref);
                if (lane == 0)
                    for (uint32_t j = next_mark; j <= mk; j++)
                        my_chk[j] = make_uint2(rpos, (rout & 0x3ffffffu) | (ref << 26));
                next_mark = mk + 1;
            }
            ref_here = rpos == wb;
        }
        uint32_t E, O;
        window_map(in, wb, seg_end, n, lane, E, O);
        const uint32_t rel = pos - wb;
        const uint32_t e = lane_shuffle(E, rel), o = lane_shuffle(O, rel);
        const bool moves = active && rel < 64;  // (an element is >= 2 bytes: every parse in the window moves on)
        const uint64_t others = ballot64(moves && rel != 0);
        if (moves) {
            out += o;
            pos = e == kBad ? kBad : wb + e;
        }
        pk = kBad;
        if (others == 0) {  // only the parses at wb moved: to wb + E(0), all of them
            const uint32_t ex = lane_get(E, 0);
            const uint32_t np = ex == kBad ? kBad : wb + ex;
            if (np < seg_end && ballot64(pos < np) == 0)  // (parses that are out of the segment lie above every np)
                pk = np;
        }
    }
    maps[(size_t)s * 64 + lane] = make_uint2(pos, out);
}

// ---- 1b. entries FAR from a segment's start
// A parse enters a segment more than 63 bytes behind its start when an element (a literal of hundreds of bytes or
// more) reaches that far into it: no map covers such an entry.  From there the parse is followed with 64-byte
// windows (2 KiB of the stream at a time in LDS: one trip to memory, not one per window) until it stands on a
// checkpoint of the segment -- then that hypothesis' exit is its exit -- or leaves the segment.
// p, out: the parse (wave-uniform); false: a malformed element.  map_of(rel) = the segment's map entry.
static constexpr uint32_t kFarChunks = 128;  // 16-byte pieces of the stream kept in LDS while windows are walked
template <typename MapFn>
__device__ __forceinline__ bool follow_far_entry(const uint8_t *__restrict__ in, uint32_t n, uint32_t seg_base,
                                                 uint32_t seg_end, uint32_t n_chk, const uint2 *__restrict__ seg_chk,
                                                 MapFn map_of, uint8_t *far_bytes, uint32_t &p, uint32_t &out,
                                                 uint32_t lane)
{
    // the segment's checkpoints, one per lane (n_chk <= 64): the parse has met a hypothesis of pass 1 when one of
    // them is an element start of its own (a checkpoint is AN element start of its hypothesis shortly behind its mark,
    // not the first one behind it: comparing only the first element behind a mark can miss for KiB after KiB)
    (void)seg_base;
    const uint2 kc = lane < n_chk ? seg_chk[lane] : make_uint2(kBad, 0);
    uint32_t b0 = kBad, bend = 0;  // far_bytes holds stream bytes [b0, bend)
    while (p < seg_end) {
        if (!(p >= b0 && p + 72u <= bend) && n + 8u - p >= 72u) {  // (the caller's buffer is readable to n + 8)
            const uint32_t avail = n + 8u - p;
            const uint32_t nck = avail / 16u < kFarChunks ? avail / 16u : kFarChunks;
            __syncthreads();  // (single-wave workgroup: orders the LDS accesses of its lanes)
#pragma unroll
            for (uint32_t k = 0; k < kFarChunks; k += 64) {
                if (k + lane < nck) {
                    const u32x4 v = ld128(in + p + 16u * (k + lane));
                    __builtin_memcpy(far_bytes + 16u * (k + lane), &v, 16);
                }
            }
            __syncthreads();
            b0 = p;
            bend = p + 16u * nck;
        }
        // the elements that start in [p, p + 62] on the chain from p
        const uint32_t q = p + lane;
        Elem e;
        e.size = 0;
        e.olen = 0;
        e.ok = true;
        if (q < n) {
            uint64_t x;
            if (p >= b0 && p + 72u <= bend)
                __builtin_memcpy(&x, far_bytes + (q - b0), 8);
            else
                x = ld64(in + q);
            e = element_of(x, q, n);
        }
        const uint32_t end = lane + e.size;
        const uint32_t nx = (q >= n || !e.ok || end > 62 || lane == 63) ? 63u : end;
        uint64_t Mm;
        {
            uint32_t jump = nx, on = lane == 0 ? 1u : 0u;  // (pointer doubling, as the decoder finds a window's elements)
#pragma unroll
            for (int r = 0; r < 5; r++) {
                on |= (uint32_t)__builtin_amdgcn_ds_permute((int)((on ? jump : 63u) << 2), (int)on);
                jump = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jump << 2), (int)jump);
            }
            Mm = ballot64(on != 0 && lane != 63 && q < seg_end);  // only the elements that start inside the segment
        }
        if (ballot64(((Mm >> lane) & 1) && !e.ok) != 0)
            return false;
        const bool mine = (Mm >> lane) & 1;
        const uint32_t e_last = 63u - (uint32_t)__builtin_clzll(Mm);  // (lane 0 is always on the chain: Mm != 0)
        const uint32_t olen = mine ? e.olen : 0u;
        const uint32_t incl = wave_inclusive_sum(olen);
        const uint32_t start = out + (incl - olen);
        {
            const uint32_t relc = kc.x - p;
            const uint64_t meet = ballot64(kc.x != kBad && kc.x >= p && relc < 64u && ((Mm >> (relc & 63u)) & 1ull));
            if (meet) {
                const uint32_t jj = ctz64(meet);  // the checkpoint's hypothesis: its exit, the bytes it has produced by then and by here
                const uint32_t cy = lane_get(kc.y, jj);
                const uint2 m = map_of(cy >> 26);
                if (uni(m.x) == kBad)
                    return false;
                out = lane_get(start, lane_get(relc, jj)) + uni(m.y) - (cy & 0x3ffffffu);
                p = uni(m.x);
                return true;
            }
        }
        out += lane_get(incl, 63);
        p += e_last + lane_get(e.size, e_last);
    }
    return true;
}

// Pass 1b, one wave per segment: WHERE parses can enter a segment far from its start is known from the maps -- the
// exits of the hypothetical parses of the segments before it, and the true parse's exit is one of them.  Every
// segment follows the exit that most of its 64 parses share (they fall into step within a few elements: in a
// compressible stream that is the true parse) into the segment where it lands, if it lands far from that one's
// start, and leaves (entry, exit, bytes produced, ok) in THAT segment's far map; passes 2a-2c look entries up there
// instead of leaving them to the one wave of the serial pass (24 windows each, ~1 us a window).  An entry nobody
// foresaw (a minority's exit, more than kFarSlots entries into one segment) is walked by the serial pass as before.
static constexpr uint32_t kFarSlots = 4;
struct FarEntry {
    uint32_t pos, exit, out, st;  // st: 0 = followed, 1 = malformed element on the way
};
__global__ __launch_bounds__(64) void stream_far_maps_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t n_seg,
                                                             uint32_t seg_shift, const uint2 *__restrict__ maps,
                                                             const uint2 *__restrict__ chk, FarEntry *__restrict__ far,
                                                             uint32_t *__restrict__ far_cnt,
                                                             const uint32_t *__restrict__ covered)
{
    __shared__ __attribute__((aligned(16))) uint8_t far_bytes[kFarChunks * 16];
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    if (s >= n_seg || *covered)
        return;
    const uint32_t q = maps[(size_t)s * 64 + lane].x;  // (an exit lies at or behind this segment's end)
    const bool far_q = q != kBad && q < n && (q & (kSeg - 1u)) >= 64u;
    uint64_t todo = ballot64(far_q);
    uint32_t best = kBad, best_cnt = 0;
    for (uint32_t i = 0; i < 8 && todo; i++) {  // (the parses have joined into a few packs by the segment's end)
        const uint32_t q0 = lane_get(q, ctz64(todo));
        const uint64_t m = ballot64(far_q && q == q0);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        if (cnt > best_cnt) {
            best_cnt = cnt;
            best = q0;
        }
        todo &= ~m;
    }
    if (best_cnt == 0)
        return;
    const uint32_t t = best >> seg_shift;  // the segment the exit lands in (t > s)
    const uint32_t seg_base = t * kSeg;
    const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
    const uint32_t n_chk = kSeg >> kChkShift;
    uint32_t p = best, out = 0;
    const bool ok = follow_far_entry(in, n, seg_base, seg_end, n_chk, chk + (size_t)t * n_chk,
                                     [&](uint32_t rel) { return maps[(size_t)t * 64 + rel]; }, far_bytes, p, out, lane);
    uint32_t k = 0;
    if (lane == 0)
        k = atomicAdd(far_cnt + t, 1u);
    k = uni(k);
    if (k < kFarSlots && lane == 0)
        far[(size_t)t * kFarSlots + k] = FarEntry{best, p, out, ok ? 0u : 1u};
}
// the far map's answer for an entry at `pos` into segment s: slot index, or kFarSlots if it has none
__device__ __forceinline__ uint32_t far_slot(const FarEntry *__restrict__ far, uint32_t s, uint32_t pos)
{
    for (uint32_t k = 0; k < kFarSlots; k++)
        if (far[(size_t)s * kFarSlots + k].pos == pos)
            return k;
    return kFarSlots;
}

// ---- 2. the true parse through the segments, in three short passes instead of one long serial walk
// A segment map is a function "entry offset (< 64) -> exit position, bytes produced", and such functions
// compose.  2a: one wave per CHUNK of 64 segments composes its segments' maps for all 64 entry offsets at
// once (64 dependent look-ups, every chunk in parallel).  2b: ONE wave follows the true parse from chunk
// to chunk through those chunk maps (a few hundred dependent steps for a GiB, not one per segment).
// 2c: one wave per chunk walks its segments again from the now known entry and writes entries[s].
// A hypothesis stops where the parse enters a segment FAR from its start (behind a literal that reaches
// more than 64 bytes into it); a chunk in which the true parse does that is walked by 2b itself:
// segment maps where it enters near a segment's start, single elements elsewhere -- and runs of
// equal-sized elements (a stream of incompressible fragments is one: ~64 KiB literals) 64 at a time,
// lane k trying the element at p + k * size.
// entries[s] = (position, bytes produced) where the true parse enters segment s, (kBad, 0) where no
// element starts in the segment; result[0] = 0 if the stream parses cleanly to its end.
static constexpr uint32_t kChunkSegs = 64;
static constexpr uint32_t kStRun = 0, kStFar = 1, kStBad = 2;

struct ChunkMap {
    uint32_t pos, out, st, pad;
};

__global__ __launch_bounds__(64) void stream_chunk_maps_kernel(uint32_t n, uint32_t n_seg, uint32_t seg_shift,
                                                               const uint2 *__restrict__ maps,
                                                               ChunkMap *__restrict__ cmaps,
                                                               const FarEntry *__restrict__ far,
                                                               const uint32_t *__restrict__ covered)
{
    if (*covered)
        return;
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    const uint32_t s0 = c * kChunkSegs, s1 = s0 + kChunkSegs < n_seg ? s0 + kChunkSegs : n_seg;
    uint32_t pos = s0 * kSeg + lane, out = 0, st = kStRun;  // hypothesis: the parse enters the chunk at byte `lane`
    for (uint32_t s = s0; s < s1; s++) {
        const uint32_t seg_base = s * kSeg;
        const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
        if (st == kStRun && pos < seg_end) {  // (pos >= seg_end: the segment lies inside an element)
            const uint32_t rel = pos - seg_base;
            if (rel < 64) {
                const uint2 m = maps[(size_t)s * 64 + rel];
                if (m.x == kBad) {
                    st = kStBad;
                } else {
                    pos = m.x;
                    out += m.y;
                }
            } else {  // far into the segment: pass 1b has followed the entries it could foresee
                const uint32_t k = far_slot(far, s, pos);
                if (k == kFarSlots) {
                    st = kStFar;  // not this one: the serial pass walks this chunk
                } else {
                    const FarEntry f = far[(size_t)s * kFarSlots + k];
                    if (f.st != 0) {
                        st = kStBad;
                    } else {
                        pos = f.exit;
                        out += f.out;
                    }
                }
            }
        }
    }
    cmaps[(size_t)c * 64 + lane] = ChunkMap{pos, out, st, 0};
}

#ifdef SZ_STREAM_DEBUG
__device__ uint32_t g_bnd_dbg;  // (debug builds: the boundaries pass's slowest wave, reported by stream_debug_report_kernel)
__global__ void stream_debug_report_kernel()
{
    printf("boundaries pass: most windows walked by one wave %u (segment %u mod 8192)\n", g_bnd_dbg >> 13, g_bnd_dbg & 8191u);
    g_bnd_dbg = 0;
}
__device__ unsigned long long g_res_dbg[16];  // (debug builds: counts and clocks of the serial pass; tools/streamdbg.py)
#define RES_COUNT(i, v) do { if (lane == 0) g_res_dbg[i] += (v); } while (0)
#define RES_CLOCK(t) unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#else
#define RES_COUNT(i, v) do { } while (0)
#define RES_CLOCK(t) do { } while (0)
#endif
// The parse from p (bytes produced: out) up to the first element that starts at or past `until`,
// recording in entries[] the first element that starts in each segment.  One wave, wave-uniform state.
// `rows` (LDS) takes the maps of segments [row_s0, row_s0 + row_cnt) at the first look-up of one of them (all rows
// in flight at once): every look-up would otherwise be a dependent trip to memory of this one wave.  (A stream of
// incompressible fragments never looks a map up: its rows are not fetched.)
__device__ __forceinline__ bool walk_segments(const uint8_t *__restrict__ in, uint32_t n, uint32_t seg_shift,
                                              const uint2 *__restrict__ maps, const uint2 *__restrict__ chk,
                                              uint2 *__restrict__ entries, uint32_t &p, uint32_t &out,
                                              uint32_t until, uint32_t lane, uint2 *rows, uint32_t row_s0,
                                              uint32_t row_cnt, uint8_t *far_bytes, uint32_t &run_size,
                                              const FarEntry *__restrict__ far)
{
    const uint32_t kSeg = 1u << seg_shift;
    uint32_t last_seg = kBad;
    bool rows_here = false;
    auto map_of = [&](uint32_t s, uint32_t rel) -> uint2 {  // (s, rel wave-uniform)
        if (s - row_s0 >= row_cnt)
            return maps[(size_t)s * 64 + rel];
        if (!rows_here) {
            RES_CLOCK(tf0);
            __syncthreads();  // (single-wave workgroup: orders the LDS accesses of its lanes)
#pragma unroll 16
            for (uint32_t i = 0; i < row_cnt; i++)
                rows[i * 64 + lane] = maps[(size_t)(row_s0 + i) * 64 + lane];
            __syncthreads();
            rows_here = true;
#ifdef SZ_STREAM_DEBUG
            RES_CLOCK(tf1);
            RES_COUNT(7, tf1 - tf0);
#endif
        }
        uint2 m = rows[(s - row_s0) * 64 + rel];
        // (kept apart from the load of the other branch: merged, the two become ONE load through a generic pointer,
        // which waits like a load from memory -- and for the store to entries[] in front of it: 930 clocks a step)
        asm volatile("" : "+v"(m.x), "+v"(m.y));
        return m;
    };
    while (p < until) {
        const uint32_t s = p >> seg_shift;
        const uint32_t seg_base = s * kSeg;
        if (s != last_seg) {
            if (lane == 0)
                entries[s] = make_uint2(p, out);
            last_seg = s;
        }
        const uint32_t rel = p - seg_base;
        if (rel < 64) {  // the segment's map covers the rest of the segment
            RES_COUNT(0, 1);
            const uint2 m = map_of(s, rel);
            const uint32_t ex = uni(m.x), os = uni(m.y);
            if (ex == kBad)
                return false;
            p = ex;
            out += os;
            continue;
        }
        // far from the segment's start: the element at p, and -- lane k -- the ones at p + k * size,
        // as long as they have the same size (what a run of equal literals or copies looks like)
        // (lane k looks at p + k * the size of the run before right away: when this element has that size again --
        // a stream of incompressible fragments is one ~64 KiB literal after the other -- the run costs one trip to
        // memory instead of two)
        const uint64_t qs64 = (uint64_t)p + (uint64_t)lane * run_size;
        const bool ins = qs64 < until;
        const uint32_t qs = ins ? (uint32_t)qs64 : p;
        const Elem es = element_at(in, qs, n);
        const uint32_t size0 = lane_get(es.size, 0);
        if (!lane_get(es.ok ? 1u : 0u, 0))
            return false;
        if (size0 < 64) {
            // small elements (the inside of a compressible fragment): pass 1b's answer if it has one for this
            // entry, else 64-byte window steps until the parse stands on a checkpoint of the segment
            const uint32_t seg_end = seg_base + kSeg < until ? seg_base + kSeg : until;
            const uint32_t n_chk = kSeg >> kChkShift;
            RES_COUNT(1, 1);
            const uint32_t fk = seg_end == seg_base + kSeg || seg_end == n ? far_slot(far, s, p) : kFarSlots;
            if (fk != kFarSlots) {
                const FarEntry f = far[(size_t)s * kFarSlots + fk];
                if (uni(f.st) != 0)
                    return false;
                p = uni(f.exit);
                out += uni(f.out);
                continue;
            }
            if (!follow_far_entry(in, n, seg_base, seg_end, n_chk, chk + (size_t)s * n_chk,
                                  [&](uint32_t r) { return map_of(s, r); }, far_bytes, p, out, lane))
                return false;
            continue;
        }
        RES_COUNT(3, 1);
        const uint64_t q64 = (uint64_t)p + (uint64_t)lane * size0;
        const bool inside = q64 < until;  // (until <= n)
        const uint32_t q = inside ? (uint32_t)q64 : p;
        Elem e = es;  // (size0 == run_size: q64 == qs64, the elements are the ones already looked at)
        if (size0 != run_size)
            e = element_at(in, q, n);
        run_size = size0;
        const uint64_t good = ballot64(inside && e.ok && e.size == size0);
        const uint32_t cnt = good == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~good);  // >= 1: lane 0 is the element at p
        const bool mine = lane < cnt;
        const uint32_t olen = mine ? e.olen : 0u;
        const uint32_t incl = wave_inclusive_sum(olen);
        const uint32_t my_seg = q >> seg_shift;
        const uint32_t prev_seg = lane == 0 ? last_seg : (uint32_t)((q64 - size0) >> seg_shift);
        if (mine && my_seg != prev_seg)
            entries[my_seg] = make_uint2(q, out + (incl - olen));
        asm volatile("");  // (keeps the per-lane store's join apart from the loop's uniform state)
        last_seg = (uint32_t)(((uint64_t)p + (uint64_t)(cnt - 1) * size0) >> seg_shift);
        out += lane_get(incl, 63);
        p += cnt * size0;  // (< 2^32: every consumed element ends inside the stream)
    }
    return true;
}

__global__ __launch_bounds__(64) void stream_resolve_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t body,
                                                            uint32_t ulen, uint32_t n_seg, uint32_t seg_shift,
                                                            const uint2 *__restrict__ maps, const uint2 *__restrict__ chk,
                                                            const ChunkMap *__restrict__ cmaps,
                                                            uint2 *__restrict__ entries, uint2 *__restrict__ centries,
                                                            uint32_t *__restrict__ result,
                                                            const FarEntry *__restrict__ far,
                                                            const uint32_t *__restrict__ covered)
{
    if (*covered)
        return;  // (result[0] stays 0: the stream parsed cleanly to its end in pass 0)
    // One wave, and every step's look-up depends on the step before: the tables it reads are brought into LDS in
    // whole rows first (all 64 entries of a chunk map / of a segment map, many rows in flight at once), so that a
    // step costs an LDS read instead of a trip to memory (~2 us each: the rows were written by other XCDs).
    constexpr uint32_t kCmGroup = 16;
    __shared__ ChunkMap cm_rows[kCmGroup * 64];  // chunk maps of kCmGroup chunks
    __shared__ uint2 seg_rows[kChunkSegs * 64];  // segment maps of the chunk that is being walked
    __shared__ __attribute__((aligned(16))) uint8_t far_bytes[kFarChunks * 16];
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_chunk = (n_seg + kChunkSegs - 1) / kChunkSegs;
    uint32_t p = body, out = 0;
    uint32_t run_size = 0;  // size of the elements of the last run of equal-sized elements (0: none yet)
    bool ok = true;
    RES_CLOCK(tk0);
    for (uint32_t c = 0; c < n_chunk && ok; c++) {
        if (c % kCmGroup == 0) {
            __syncthreads();  // (the rows of the group before have been read)
            // (all rows requested before the first one is stored: rows past the last chunk repeat it, unread)
            ChunkMap row[kCmGroup];
#pragma unroll
            for (uint32_t i = 0; i < kCmGroup; i++) {
                const uint32_t ci = c + i < n_chunk ? c + i : n_chunk - 1;
                row[i] = cmaps[(size_t)ci * 64 + lane];
            }
#pragma unroll
            for (uint32_t i = 0; i < kCmGroup; i++)
                cm_rows[i * 64 + lane] = row[i];
            __syncthreads();
        }
        const uint64_t cb64 = (uint64_t)c * kChunkSegs * kSeg;
        const uint64_t ce64 = cb64 + (uint64_t)kChunkSegs * kSeg;
        const uint32_t chunk_base = (uint32_t)cb64, chunk_end = ce64 < n ? (uint32_t)ce64 : n;
        uint2 ce = make_uint2(kBad, 0);  // nothing for pass 2c to do in this chunk
        if (p < chunk_end) {
            bool regular = false;
            if (p - chunk_base < 64) {
                const ChunkMap cm = cm_rows[(c % kCmGroup) * 64 + (p - chunk_base)];
                if (uni(cm.st) == kStRun) {  // the whole chunk through near entries: pass 2c fills its entries
                    ce = make_uint2(p, out);
                    p = uni(cm.pos);
                    out += uni(cm.out);
                    regular = true;
                }
            }
            RES_COUNT(regular ? 4 : 5, 1);
            if (!regular) {
                const uint32_t s0 = c * kChunkSegs, cnt = n_seg - s0 < kChunkSegs ? n_seg - s0 : kChunkSegs;
                RES_CLOCK(tw0);
                ok = walk_segments(in, n, seg_shift, maps, chk, entries, p, out, chunk_end, lane, seg_rows, s0, cnt, far_bytes, run_size, far);
#ifdef SZ_STREAM_DEBUG
                RES_CLOCK(tw1);
                RES_COUNT(6, tw1 - tw0);
#endif
            }
        }
        if (lane == 0)
            centries[c] = ce;
    }
    ok = ok && p == n && out == ulen;
#ifdef SZ_STREAM_DEBUG
    {
        RES_CLOCK(tk1);
        if (lane == 0) {
            printf("serial pass: %llu clocks; chunks through their map %llu, walked %llu (%llu clocks); in walked chunks: segment maps "
                   "%llu, entries far from a segment's start %llu (%llu windows, %llu clocks), runs of equal elements %llu; rows into LDS %llu clocks\n",
                   tk1 - tk0, g_res_dbg[4], g_res_dbg[5], g_res_dbg[6], g_res_dbg[0], g_res_dbg[1], g_res_dbg[2], g_res_dbg[8], g_res_dbg[3], g_res_dbg[7]);
            for (int i = 0; i < 16; i++) g_res_dbg[i] = 0;
        }
    }
#endif
    if (lane == 0)
        result[0] = ok ? 0u : 1u;
}

// 2c: the segments of the chunks that pass 2b crossed through their chunk map
__global__ __launch_bounds__(64) void stream_chunk_entries_kernel(uint32_t n, uint32_t n_seg, uint32_t seg_shift,
                                                                  const uint2 *__restrict__ maps,
                                                                  const uint2 *__restrict__ centries,
                                                                  uint2 *__restrict__ entries,
                                                                  const FarEntry *__restrict__ far,
                                                                  const uint32_t *__restrict__ covered)
{
    if (*covered)
        return;
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    const uint2 ce = centries[c];
    uint32_t p = uni(ce.x), out = uni(ce.y);
    if (p == kBad)
        return;
    const uint32_t s0 = c * kChunkSegs, s1 = s0 + kChunkSegs < n_seg ? s0 + kChunkSegs : n_seg;
    for (uint32_t s = s0; s < s1; s++) {
        const uint32_t seg_base = s * kSeg;
        const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
        if (p >= seg_end)
            continue;  // inside an element: entries[s] stays (kBad, 0)
        if (p < seg_base)
            return;  // (not reached)
        if (lane == 0)
            entries[s] = make_uint2(p, out);
        if (p - seg_base >= 64u) {  // a far entry: pass 2a came through the far map here
            const uint32_t k = far_slot(far, s, p);
            if (k == kFarSlots)
                return;  // (not reached: pass 2a ran this very path)
            const FarEntry f = far[(size_t)s * kFarSlots + k];
            p = uni(f.exit);
            out += uni(f.out);
            continue;
        }
        const uint2 m = maps[(size_t)s * 64 + (p - seg_base)];  // (near entry, not kBad: pass 2a ran this path)
        p = uni(m.x);
        out += uni(m.y);
    }
}

// ---- 3. fragment boundaries inside each segment
// off[m] := compressed offset of the element that starts at output position m * 64 KiB (0 < m < number
// of fragments); flag |= 2 if an element straddles such a position, |= 1 on a malformed element.
// A segment is NOT walked from end to end (round 2 did: 1.9 ms of a 1 GiB text stream's 8.0): the true parse is
// followed from its entry only until it stands on a checkpoint of the segment's hypothesis 0 (pass 1 left one per
// KiB; parses fall into step within a few elements) -- from there on the checkpoints ARE the true parse, shifted
// by a constant number of output bytes, so every 64 KiB output boundary inside the segment is looked up among
// them and only the KiB in front of it is walked.  (Elements on the true parse are well-formed: passes 1-2 have
// followed it to the end of the stream, result[0].)
__global__ __launch_bounds__(64) void stream_boundaries_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t ulen,
                                                               uint32_t n_seg, uint32_t seg_shift,
                                                               const uint2 *__restrict__ entries,
                                                               const uint2 *__restrict__ maps, const uint2 *__restrict__ chk,
                                                               unsigned long long *__restrict__ off,
                                                               uint32_t *__restrict__ flag,
                                                               const uint32_t *__restrict__ covered)
{
    const uint32_t kSeg = 1u << seg_shift;
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    if (s >= n_seg || *covered)
        return;
    const uint2 en = entries[s];
    uint32_t p = uni(en.x), out = uni(en.y);
    if (p == kBad)
        return;
#ifdef SZ_STREAM_DEBUG
    uint32_t dbg_windows = 0;
    unsigned long long dbg_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t0) :: "memory");
#endif
    const uint32_t seg_base = s * kSeg;
    const uint32_t seg_end = seg_base + kSeg < n ? seg_base + kSeg : n;
    const uint32_t n_chk = kSeg >> kChkShift;
    const uint2 *my_chk = chk + (size_t)s * n_chk;
    // The segment's checkpoints, one per lane (n_chk <= 64): position, bytes produced in its hypothesis' frame, hypothesis
    const uint2 kc = lane < n_chk ? my_chk[lane] : make_uint2(kBad, 0);
    const uint32_t kx = kc.x, kout = kc.y & 0x3ffffffu, kid = kc.y >> 26;
    uint32_t delta = 0, jm = kBad;  // true output position of checkpoint j = kout(j) + delta, for j >= jm with kid(j) == rid
    uint32_t rid = 0;               // the hypothesis the parse has met
    bool remet = false;             // the last window changed (rid, delta, jm)
    // one window of the walk: the elements that start in [p, p + 62] and inside the segment; records the ones that
    // start on a boundary; false: a flag was raised (the caller decodes serially)
    auto window = [&]() -> bool {
#ifdef SZ_STREAM_DEBUG
        dbg_windows++;
#endif
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
            return false;
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
            return false;
        }
        // Does this parse stand on a checkpoint in this window -- is a checkpoint's position one of its element
        // starts?  (A checkpoint is AN element start of its hypothesis shortly behind its KiB mark.)  Then the parse IS
        // that hypothesis from there on, and what lies ahead is looked up among that hypothesis' checkpoints.  Tested
        // in every window, also behind the first meeting: pass 1 changes the hypothesis it records when the one it
        // followed stands far ahead of most others (a long literal), and a walk that started from the old one's last
        // checkpoint would otherwise go on to the segment's end (455 windows in 1 % of a text stream's segments).
        {
            const uint32_t relc = kx - p;
            const bool at = kx != kBad && kx >= p && relc < 64u && ((Mm >> (relc & 63u)) & 1ull);
            const uint64_t meet = ballot64(at);
            if (meet) {
                const uint32_t jj = 63u - (uint32_t)__builtin_clzll(meet);  // the last of them
                const uint32_t id = lane_get(kid, jj);
                if (jm == kBad || id != rid) {
                    rid = id;
                    jm = jj;
                    delta = lane_get(start, lane_get(relc, jj)) - lane_get(kout, jj);
                    remet = true;
                }
            }
        }
        out += lane_get(incl, 63);
        p += e_last + lane_get(e.size, e_last);
        return true;
    };
    // ---- from the entry until the parse stands on a checkpoint (or leaves the segment)
    while (jm == kBad && p < seg_end) {
        if (!window())
            return;
    }
    if (jm == kBad)
        return;  // the whole segment was walked
    // ---- the boundaries behind the meeting point: look each one up among the checkpoints, walk the KiB before it
    for (uint32_t B = (out + 0xffffu) & ~0xffffu;;) {
        const uint2 m0 = maps[(size_t)s * 64 + rid];  // that hypothesis: (exit, bytes produced) -- bytes at the exit = m0.y + delta
        const uint32_t out_exit = uni(m0.y) + delta;
        if (!(B < out_exit && B < ulen))
            break;
        if (B == 0) {
            B += 0x10000u;
            continue;
        }
        // the last checkpoint of that hypothesis at or behind the meeting point whose element starts at or before B
        const uint32_t cy = kout + delta;
        const uint64_t okm = ballot64(lane >= jm && kx != kBad && kid == rid && cy <= B);
        if (okm != 0) {
            const uint32_t js = 63u - (uint32_t)__builtin_clzll(okm);
            const uint32_t pj = lane_get(kx, js);
            if (pj > p) {  // (never back: the walk may already stand behind it)
                p = pj;
                out = lane_get(cy, js);
            }
        }
        remet = false;
        while (p < seg_end && out <= B && !remet) {
            if (!window())
                return;
        }
        if (!remet)
            B += 0x10000u;  // (else: the same boundary again, among the checkpoints of the hypothesis just met)
    }
#ifdef SZ_STREAM_DEBUG
    {
        unsigned long long dbg_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t1) :: "memory");
        if (lane == 0)
            atomicMax(&g_bnd_dbg, (dbg_windows << 13) | (s & 8191u));  // most windows walked by a wave, its segment (never into the API's result words)
        (void)dbg_t0;
    }
#endif
}

// off[0] = body, off[n_frag] = n, everything between "not found"; entries[] = "no element starts here";
// result[0..2] = 0
__global__ __launch_bounds__(256) void stream_index_init_kernel(unsigned long long *__restrict__ off, uint32_t n_frag,
                                                                uint32_t body, uint32_t n, uint2 *__restrict__ entries,
                                                                uint32_t n_seg, uint32_t *__restrict__ result,
                                                                uint32_t *__restrict__ covered,
                                                                uint2 *__restrict__ centries, uint32_t n_chunk,
                                                                FarEntry *__restrict__ far, uint32_t *__restrict__ far_cnt)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_seg) {
        far_cnt[i] = 0;
        for (uint32_t k = 0; k < kFarSlots; k++)
            far[(size_t)i * kFarSlots + k] = FarEntry{kBad, kBad, 0, 1};
    }
    if (i <= n_frag)
        off[i] = i == 0 ? body : (i == n_frag ? n : ~0ull);
    if (i < n_seg)
        entries[i] = make_uint2(kBad, 0);
    if (i < n_chunk)
        centries[i] = make_uint2(kBad, 0);  // (pass 2b stops at the first malformed element: the chunks behind it stay so)
    if (i < 3)
        result[i] = 0;
    if (i == 0)
        *covered = 0;
}

// ---- 0. a stream of big elements needs no index
// One wave follows the parse from the preamble as long as it consists of runs of equal-sized elements of 64 bytes or
// more (lane k tries the element at p + k * size, four such rows in flight): what a stream of incompressible fragments
// looks like -- one ~64 KiB literal per fragment.  If that reaches the end of the stream, every boundary is known
// (*covered = 1) and passes 1-3 return at once: 0.2 ms instead of 2.9 ms for 1 GiB.  At the first small element the
// wave gives up (a compressible stream: after one load); the boundaries it has recorded by then are true ones.
__global__ __launch_bounds__(64) void stream_runs_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t body,
                                                         uint32_t ulen, unsigned long long *__restrict__ off,
                                                         uint32_t *__restrict__ covered)
{
    constexpr int kRows = 4;
    const uint32_t lane = threadIdx.x;
    uint32_t p = body, out = 0, size = 0;
    bool ok = true;
    while (ok && p < n) {
        if (size == 0) {  // the size of the next run: the element at p
            const Elem e0 = element_at(in, p, n);
            size = lane_get(e0.size, 0);
            if (!lane_get(e0.ok ? 1u : 0u, 0) || size < 64u) {
                ok = false;
                break;
            }
        }
        Elem e[kRows];
        uint64_t q[kRows];
        bool inside[kRows];
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            q[r] = (uint64_t)p + (uint64_t)(64u * r + lane) * size;
            inside[r] = q[r] < n;
            e[r] = element_at(in, inside[r] ? (uint32_t)q[r] : p, n);
        }
        uint32_t taken = 0;  // elements of this step that belong to the run
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            if (taken != 64u * r)
                break;  // (the run ended in the row before)
            const uint64_t good = ballot64(inside[r] && e[r].ok && e[r].size == size);
            const uint32_t cnt = good == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~good);
            const bool mine = lane < cnt;
            const uint32_t olen = mine ? e[r].olen : 0u;
            const uint32_t incl = wave_inclusive_sum(olen);
            const uint32_t start = out + (incl - olen);
            // (sums of this row stay below 2^32 where it matters: an element that would pass ulen ends the attempt)
            if (ballot64(mine && ((uint64_t)start + olen > ulen || (start >> 16) != ((start + olen - 1u) >> 16))) != 0) {
                ok = false;  // past the announced size, or across a 64 KiB boundary: not for this pass to judge
                break;
            }
            if (mine && (start & 0xffffu) == 0 && start != 0 && start < ulen)
                off[start >> 16] = (uint32_t)q[r];
            asm volatile("");  // (keeps the per-lane store's join apart from the loop's uniform state)
            out += lane_get(incl, 63);
            taken += cnt;
        }
#ifdef SZ_STREAM_DEBUG
        if (lane == 0 && p < (64u << 20))
            printf("runs: p %u size %u taken %u ok %d out %u\n", p, size, taken, (int)ok, out);
#endif
        if (!ok)
            break;
        p += taken * size;  // (< 2^32: every element taken ends inside the stream)
        if (taken != 64u * kRows)
            size = 0;  // the run is over (or the stream is): look at the next element's size
        if (taken == 0) {
            ok = false;  // (cannot happen: the element at p was valid a moment ago)
            break;
        }
    }
#ifdef SZ_STREAM_DEBUG
    if (lane == 0)
        printf("runs: end p %u n %u out %u ulen %u ok %d\n", p, n, out, ulen, (int)ok);
#endif
    if (ok && p == n && out == ulen)
        *covered = 1;
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
    const uint64_t seg = 1ull << seg_shift_for(n);  // (the same choice as the launch makes, test hook included)
    const size_t n_seg = (size_t)((n + seg - 1) / seg);
    const size_t n_chunk = (n_seg + kChunkSegs - 1) / kChunkSegs;
    return n_seg * 64 * sizeof(uint2) + n_seg * sizeof(uint2) + n_chunk * (64 * sizeof(ChunkMap) + sizeof(uint2)) +
           (size_t)((n + 1023) / 1024 + 64) * sizeof(uint2) +  // maps, entries, chunk maps + entries, checkpoints
           n_seg * (kFarSlots * sizeof(FarEntry) + sizeof(uint32_t)) + 16 + 64;  // far maps + their counters, pass 0's flag
}

// d_in: the whole raw stream (n <= SZ_STREAM_INDEX_MAX_BYTES = 2^32 - 256 KiB: positions are 32-bit, 0xffffffff is "none", and
// seg_base + segment, p + 72 and n + 8 must not wrap; readable up to n + 8), body = size of its preamble,
// ulen = the decoded size it announces, n_frag = ceil(ulen / 64 KiB).  Fills d_off[0 .. n_frag]
// (boundaries that are not found stay UINT64_MAX) and d_result[0..1] = {parse status, boundary
// flags}; d_result[2] := 0 (see launch_count_failed).
#ifdef SZ_DEBUG_SYNC
#define SZ_DBG_SYNC(name) do { fprintf(stderr, "launched %s\n", name); hipError_t e_ = hipStreamSynchronize(stream); fprintf(stderr, "  done %s: %d\n", name, (int)e_); } while (0)
#else
#define SZ_DBG_SYNC(name) do { } while (0)
#endif
hipError_t launch_stream_index(const void *d_in, uint64_t n, uint32_t body, uint64_t ulen, void *d_scratch,
                               uint64_t *d_off, uint32_t *d_result, hipStream_t stream)
{
    const uint32_t seg_shift = seg_shift_for(n);
    const uint32_t n_seg = (uint32_t)((n + (1ull << seg_shift) - 1) >> seg_shift);
    const uint32_t n_frag = (uint32_t)((ulen + kMaxBlock - 1) / kMaxBlock);
    const uint32_t n_chunk = (n_seg + kChunkSegs - 1) / kChunkSegs;
    uint2 *maps = static_cast<uint2 *>(d_scratch);
    uint2 *entries = maps + (size_t)n_seg * 64;
    ChunkMap *cmaps = reinterpret_cast<ChunkMap *>(entries + n_seg);
    uint2 *centries = reinterpret_cast<uint2 *>(cmaps + (size_t)n_chunk * 64);
    uint2 *chk = centries + n_chunk;  // (1 << seg_shift) / 1 KiB checkpoints per segment
    // (far maps behind the checkpoints' area as it is sized -- for the smallest segments --, 16-byte aligned)
    const size_t chk_area = (size_t)((n + 1023) / 1024 + 64) * sizeof(uint2);
    const size_t far_off = ((reinterpret_cast<uint8_t *>(chk) - static_cast<uint8_t *>(d_scratch)) + chk_area + 15) & ~(size_t)15;
    FarEntry *far = reinterpret_cast<FarEntry *>(static_cast<uint8_t *>(d_scratch) + far_off);
    uint32_t *far_cnt = reinterpret_cast<uint32_t *>(far + (size_t)n_seg * kFarSlots);
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    unsigned long long *off = reinterpret_cast<unsigned long long *>(d_off);
    const uint32_t n_init = n_frag + 1 > n_seg ? n_frag + 1 : n_seg;
    // (the last 64 bytes of the scratch area are nobody's: the flag of pass 0 lives there)
    uint32_t *covered = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_scratch) + stream_index_scratch_bytes(n) - 16);
    hipLaunchKernelGGL(stream_index_init_kernel, dim3(n_init / 256 + 1), dim3(256), 0, stream, off, n_frag, body,
                       (uint32_t)n, entries, n_seg, d_result, covered, centries, n_chunk, far, far_cnt);
    SZ_DBG_SYNC("stream_index_init_kernel");
#ifndef SZ_NO_RUNS_PASS
    hipLaunchKernelGGL(stream_runs_kernel, dim3(1), dim3(64), 0, stream, in, (uint32_t)n, body, (uint32_t)ulen, off, covered);
    SZ_DBG_SYNC("stream_runs_kernel");
#endif
    hipLaunchKernelGGL(stream_segment_maps_kernel, dim3(n_seg), dim3(64), 0, stream, in, (uint32_t)n, n_seg, seg_shift, maps,
                       chk, covered);
    SZ_DBG_SYNC("stream_segment_maps_kernel");
    hipLaunchKernelGGL(stream_far_maps_kernel, dim3(n_seg), dim3(64), 0, stream, in, (uint32_t)n, n_seg, seg_shift, maps, chk,
                       far, far_cnt, covered);
    SZ_DBG_SYNC("stream_far_maps_kernel");
    hipLaunchKernelGGL(stream_chunk_maps_kernel, dim3(n_chunk), dim3(64), 0, stream, (uint32_t)n, n_seg, seg_shift, maps,
                       cmaps, far, covered);
    SZ_DBG_SYNC("stream_chunk_maps_kernel");
    hipLaunchKernelGGL(stream_resolve_kernel, dim3(1), dim3(64), 0, stream, in, (uint32_t)n, body, (uint32_t)ulen,
                       n_seg, seg_shift, maps, chk, cmaps, entries, centries, d_result, far, covered);
    SZ_DBG_SYNC("stream_resolve_kernel");
    hipLaunchKernelGGL(stream_chunk_entries_kernel, dim3(n_chunk), dim3(64), 0, stream, (uint32_t)n, n_seg, seg_shift,
                       maps, centries, entries, far, covered);
    SZ_DBG_SYNC("stream_chunk_entries_kernel");
    hipLaunchKernelGGL(stream_boundaries_kernel, dim3(n_seg), dim3(64), 0, stream, in, (uint32_t)n, (uint32_t)ulen,
                       n_seg, seg_shift, entries, maps, chk, off, d_result + 1, covered);
    SZ_DBG_SYNC("stream_boundaries_kernel");
#ifdef SZ_STREAM_DEBUG
    hipLaunchKernelGGL(stream_debug_report_kernel, dim3(1), dim3(1), 0, stream);
#endif
    return hipGetLastError();
}

hipError_t launch_count_failed(const int32_t *d_status, uint32_t n_frag, uint32_t *d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(count_failed_kernel, dim3((n_frag + 255) / 256), dim3(256), 0, stream, d_status, n_frag, d_result);
    return hipGetLastError();
}

}  // namespace szk
