// snappy_compress.hip -- hand-written gfx950 (CDNA4) kernels of the Snappy block compressor.
//
// Hot path: hash table (a2) + match finder (a3) + pointer advance (a4) +
// match extension (a5) + copy element encoder (a6) + literal emission (a7) + preamble (a8) +
// result length (a9).  Output is byte-identical to software Snappy 1.1.8 (the parity target of
// the reference's tests/swcompression.cc:45), not to the RTL's own stream.
//
// Mapping to the machine (details at compress_one_block and the two kernels below)
//   * one 64-lane wavefront per <=64 KiB block; the 16384 x uint16 hash table in LDS (32 KiB: four
//     blocks per CU = one wave per SIMD) or, for the extra waves of the multi-wave kernel, in a
//     table of the same shape in global memory, never cleared: a bitmap in LDS names the buckets a block has written;
//   * the greedy parse is serial in the reference; here 64 positions are probed at once -- 64
//     consecutive bytes ("dense window", scalar walk over the ballot mask, prefix-sum packing of
//     the elements) or the next 64 probes of the skip schedule ("strided batch") -- and the table
//     ends up exactly as the serial parse leaves it;
//   * long matches extend up to a 16-byte boundary of the input, then 4 KiB of both strings per step
//     (one OR-reduction and one ballot); literal bytes and split copy elements are written by all
//     lanes; incompressible blocks stream through the register file.
//
// No MFMA: this is byte/integer work bounded by HBM ("Roofline").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "wave_utils.h"

namespace szk {

// Placed behind a per-lane `if` that would otherwise be the last statement of a wave-uniform branch
// arm: keeps the point where the lanes meet again a block of its own.  Without it that point is merged
// with the point where the wave-uniform arms meet, and the compiler then has to assume that every value
// merged there differs between lanes (vector registers, 64-bit vector arithmetic for the masks).
#define UNIFORM_JOIN() asm volatile("")

// KiB of each string one step of a long match extension keeps in flight (match_length)
static constexpr int kLdsWavePrio = 3;
static constexpr int kMatchPiecesDefault = 4;

#include "diag.h"  // MARK / TRACE / PROF_* / PE_*: nothing in the product build

static constexpr uint32_t kHashMul = 0x1e35a7bdu;  // HashTable.scala:51-53, Snappy HashBytes
static constexpr uint32_t kMaxTable = 16384u;
static constexpr uint32_t kInputMargin = 15u;  // CompressionAccelerator.scala:76-78
// (Variants that were measured and lost live as patches under tools/patches/: heads of the next tile issued ahead of the
// walk -- SZ_SPEC_MODE 1..3, rounds 2 and 5 -- and a second visit's candidates re-read from memory -- SZ_REVISIT_REGS=0,
// rounds 2-4 --: spec_heads_and_revisit_from_memory.patch; EXPERIMENTS.md R5.4.)
static constexpr uint32_t kFastMargin = 128u;  // the fast dense path stays this far from the last probe position
#ifndef SZ_SKIP_DEAD_LANES
#define SZ_SKIP_DEAD_LANES 1
#endif
static constexpr bool kSkipDeadLanes = SZ_SKIP_DEAD_LANES != 0;  // global-table waves: no look-up, no candidate string for the lanes below a tile's entry

// Probe schedule of the scan loop after a (re)start: off[k] = distance of the k-th probe from the
// first one.  skip_0 = 32, off_{k+1} = off_k + (skip_k >> 5), skip_{k+1} = skip_k + (skip_k >> 5).
// 266 probes span 64 KiB; entries past that saturate so they always fail the limit check.
static constexpr int kProbeEntries = 448;
struct ProbeSchedule {
    uint32_t off[kProbeEntries];
};
static constexpr ProbeSchedule make_probe_schedule()
{
    ProbeSchedule s{};
    uint32_t skip = 32, pos = 0;
    for (int k = 0; k < kProbeEntries; k++) {
        s.off[k] = pos;
        uint32_t step = skip >> 5;
        skip += step;
        pos += step;
        if (pos > (1u << 20)) {
            pos = 1u << 20;
            skip = 32;
        }
    }
    return s;
}
__device__ const ProbeSchedule g_probe = make_probe_schedule();

// Number of leading equal bytes (4..16) of two 16-byte strings whose first 4 bytes are equal.
__device__ __forceinline__ uint32_t matchlen16(uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b1,
                                               uint32_t b2, uint32_t b3)
{
    const uint32_t d1 = a1 ^ b1, d2 = a2 ^ b2, d3 = a3 ^ b3;
    uint32_t m = 16;
    if (d3) m = 12 + (__builtin_ctz(d3) >> 3);
    if (d2) m = 8 + (__builtin_ctz(d2) >> 3);
    if (d1) m = 4 + (__builtin_ctz(d1) >> 3);
    return m;
}

// Same over 32 bytes (second halves a2/b2); 4..32.  Per dword: bit index of the first difference
// (v_ffbl_b32 of 0 is 0xffffffff and the saturating add keeps it there); minimum over the dwords.
__device__ __forceinline__ uint32_t diff_bit(uint32_t x, uint32_t y, uint32_t base)
{
    uint32_t f, r;
    asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x ^ y));
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(f), "v"(base));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
}
__device__ __forceinline__ uint32_t matchlen32(const u32x4 &a, const u32x4 &a2, const u32x4 &b, const u32x4 &b2)
{
    uint32_t m = umin3(diff_bit(a.y, b.y, 32), diff_bit(a.z, b.z, 64), diff_bit(a.w, b.w, 96));
    m = umin3(m, diff_bit(a2.x, b2.x, 128), diff_bit(a2.y, b2.y, 160));
    m = umin3(m, diff_bit(a2.z, b2.z, 192), diff_bit(a2.w, b2.w, 224));
    return (m < 256u ? m : 256u) >> 3;
}

// Literal element (a7): tag (len-1)<<2 for len<=60 (CompressionAccelerator.scala:172,231-233);
// software Snappy tags 60/61 + 1/2 little-endian bytes of len-1 for longer runs (len<=65536).
template <int kDepth = 16>
__device__ __forceinline__ uint32_t emit_literal(uint8_t *__restrict__ out, uint32_t op,
                                                 const uint8_t *__restrict__ s, uint32_t len,
                                                 uint32_t lane)
{
    const uint32_t v = len - 1;
    uint32_t hdr, word;
    if (v < 60) {
        hdr = 1;
        word = v << 2;
    } else if (v < 256) {
        hdr = 2;
        word = (60u << 2) | (v << 8);
    } else {
        hdr = 3;
        word = (61u << 2) | (v << 8);
    }
    if (lane < hdr)
        out[op + lane] = (uint8_t)(word >> (8 * lane));
    wave_copy<kDepth>(out + op + hdr, s, len, lane);
    return op + hdr + len;
}

// Copy elements (a6), byte layouts of CopyStreamFormer (CopyCompress.scala:236-250): 2-byte form
// iff len<=11 and offset<2048, else 3-byte form.  A long match is cut 64,64,...,(60),rest.
__device__ __forceinline__ uint32_t emit_copy(uint8_t *__restrict__ out, uint32_t op, uint32_t offset,
                                              uint32_t len, uint32_t lane)
{
    const uint32_t lo = offset & 0xffu, hi = offset >> 8;
    uint8_t *o = out + op;
    uint32_t bytes = 0;
    uint32_t rest = len;
    if (len > 64) {
        const uint32_t k64 = (len - 4) >> 6;  // number of leading 64-byte elements
        rest = len - (k64 << 6);              // 4..67
        // the 3-byte element [fe lo hi] repeated k64 times: written as (unaligned) dwords of the
        // period-3 byte pattern -- dword j starts at pattern phase j mod 3 -- plus up to 3 tail bytes
        const uint32_t w64 = 0xfeu | (lo << 8) | (hi << 16);  // 2 | (63<<2)
        const uint32_t total = 3 * k64, nd = total >> 2;
        const uint32_t p0 = w64 | (w64 << 24);                          // fe lo hi fe
        const uint32_t p1 = (w64 >> 8) | (w64 << 16);                   // lo hi fe lo
        const uint32_t q2 = (hi) | (0xfeu << 8) | (lo << 16) | (hi << 24);  // hi fe lo hi
        for (uint32_t j = lane; j < nd; j += 64) {
            const uint32_t ph = j % 3u;
            const uint32_t v = ph == 0 ? p0 : (ph == 1 ? p1 : q2);
            __builtin_memcpy(o + 4 * (size_t)j, &v, 4);
        }
        const uint32_t tail = total & 3u;
        if (lane < tail) {
            const uint32_t jb = (nd << 2) + lane;
            o[jb] = (uint8_t)(w64 >> (8 * (jb % 3u)));
        }
        o += total;
        bytes = total;
        if (rest > 64) {
            const uint32_t w60 = 0xeeu | (lo << 8) | (hi << 16);  // 2 | (59<<2)
            if (lane < 3)
                o[lane] = (uint8_t)(w60 >> (8 * lane));
            o += 3;
            bytes += 3;
            rest -= 60;
        }
    }
    uint32_t word, nb;
    if (rest < 12 && offset < 2048) {
        word = 1u | ((rest - 4) << 2) | ((offset >> 8) << 5) | (lo << 8);
        nb = 2;
    } else {
        word = 2u | ((rest - 1) << 2) | (lo << 8) | (hi << 16);
        nb = 3;
    }
    if (lane < nb)
        o[lane] = (uint8_t)(word >> (8 * lane));
    return op + bytes + nb;
}

// Match extension (a5): `m` bytes are already known equal; compares on up to the block end n.
template <int kPieces>
__device__ __forceinline__ uint32_t match_length(const uint8_t *__restrict__ in, uint32_t cand,
                                                 uint32_t ip, uint32_t n, uint32_t lane, uint32_t m)
{
    ML_START();
    {
        // 49..64 bytes, so that the 16-byte loads of the ip string below are aligned
        const uint32_t l1 = 64 - ((ip + m) & 15u);
        const uint32_t i = m + lane;
        const bool eq = lane >= l1 || ((ip + i < n) && in[cand + i] == in[ip + i]);
        const uint64_t ne = ~__ballot(eq);
        ML_LAP(0);
        if (ne)
            return m + ctz64(ne);
        m += l1;
    }
    // long match: 16 B per lane and KiB; 4 KiB of both strings in flight per step while that much is left
    auto eq16 = [&](const u32x4 &a, const u32x4 &b) -> uint32_t {  // equal leading bytes of two 16-byte strings
        const uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
        uint32_t eqb = 16;
        if (d3) eqb = 12 + (__builtin_ctz(d3) >> 3);
        if (d2) eqb = 8 + (__builtin_ctz(d2) >> 3);
        if (d1) eqb = 4 + (__builtin_ctz(d1) >> 3);
        if (d0) eqb = (__builtin_ctz(d0) >> 3);
        return eqb;
    };
    // Every step below is one dependent round trip to memory, and bytes of this block that nobody has
    // touched yet come from HBM (2-3 us under load): after the first whole 4 KiB matched, one line
    // per lane of the REST of the block's own input is requested at once, so that the steps that
    // follow find their bytes on the way in or in the L2 (the other string is this same data d bytes
    // earlier).  Nothing is fetched that the block would not read anyway.
    // Every step is one dependent round trip to memory (several thousand cycles under load, more for
    // bytes of this block nobody has touched yet), so a step keeps kPieces KiB of both strings in
    // flight; what a step costs in instructions is small beside that: one OR-reduction and one ballot,
    // the place of the difference is only worked out in the step that has one.
    auto locate = [&](const u32x4 (&a)[kPieces], const u32x4 (&b)[kPieces]) -> uint32_t {
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            const uint32_t eqb = eq16(a[k], b[k]);
            const uint64_t ne = ballot64(eqb < 16);
            if (ne) {
                const uint32_t f = ctz64(ne);
                return m + 1024 * k + 16 * f + lane_get(eqb, f);
            }
        }
        return m;  // (not reached: the caller saw a difference)
    };
    auto differ = [&](const u32x4 (&a)[kPieces], const u32x4 (&b)[kPieces]) -> bool {
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < kPieces; k++)
            d |= (a[k].x ^ b[k].x) | (a[k].y ^ b[k].y) | (a[k].z ^ b[k].z) | (a[k].w ^ b[k].w);
        return ballot64(d != 0) != 0;
    };
    // (round 5) A match whose offset IS the step -- data that repeats with a period of kPieces KiB: pages, records, the
    // low-entropy workload's 4 KiB pattern -- compares every step's bytes with the step before: the candidate string of a
    // step is what the step before loaded as ITS string, so only one of the two strings is read from memory (half the
    // loads of the one phase of such a block that is bound by bandwidth: every wave of the launch is in it at once).
    // A loop of its own: two sets of registers live at a time (this step's bytes and the last step's), as in the general one.
    if (ip - cand == 1024u * kPieces && ip + m + 2048 * kPieces <= n) {
        u32x4 pb[kPieces];
        {
            u32x4 a[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; k++) {
                a[k] = ld128(in + cand + m + 1024 * k + 16 * lane);
                pb[k] = ld128(in + ip + m + 1024 * k + 16 * lane);
            }
            if (differ(a, pb)) {
                ML_LAP(2);
                return locate(a, pb);
            }
            m += 1024 * kPieces;
        }
        while (ip + m + 1024 * kPieces <= n) {
            u32x4 b[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; k++)
                b[k] = ld128(in + ip + m + 1024 * k + 16 * lane);
            if (differ(pb, b)) {
                ML_LAP(2);
                return locate(pb, b);
            }
            ML_LAP(1);
#pragma unroll
            for (int k = 0; k < kPieces; k++)
                pb[k] = b[k];
            m += 1024 * kPieces;
        }
    }
    while (ip + m + 1024 * kPieces <= n) {
        u32x4 a[kPieces], b[kPieces];
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            a[k] = ld128(in + cand + m + 1024 * k + 16 * lane);
            b[k] = ld128(in + ip + m + 1024 * k + 16 * lane);
        }
        if (differ(a, b)) {
            ML_LAP(2);
            return locate(a, b);
        }
        ML_LAP(1);
        m += 1024 * kPieces;
    }
    // the last < kPieces KiB: the 16-byte pieces that lie wholly inside the block and the < 16 bytes
    // after them, all in flight together
    {
        const uint32_t rem = n - (ip + m), whole = rem & ~15u, r15 = rem & 15u;
        u32x4 a[kPieces], b[kPieces];
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            a[k] = b[k] = u32x4{0, 0, 0, 0};
            if (1024u * k + 16u <= rem) {
                if (1024u * k + 16u * lane + 16u <= rem) {
                    a[k] = ld128(in + cand + m + 1024 * k + 16 * lane);
                    b[k] = ld128(in + ip + m + 1024 * k + 16 * lane);
                }
                UNIFORM_JOIN();
            }
        }
        uint32_t ca = 0, cb = 1;  // lanes at and past r15 differ: the block ends there
        if (lane < r15) {
            ca = in[cand + m + whole + lane];
            cb = in[ip + m + whole + lane];
        }
        UNIFORM_JOIN();
        ML_LAP(3);
        if (differ(a, b))
            return locate(a, b);
        return m + whole + ctz64(ballot64(ca != cb));
    }
}

// Wave-uniform by construction, said so to the compiler (free where it already knows: a readfirstlane
// of a scalar value folds away; where it does not, this is what brings the value back from vector
// registers into scalar ones)
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}
__device__ __forceinline__ bool unib(bool v) { return uni(v ? 1u : 0u) != 0; }

// select among six registers by a wave-uniform index (kept as scalars on purpose: an array here
// is turned into a scratch lookup by the compiler)
__device__ __forceinline__ uint32_t sel6(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4,
                                         uint32_t r5, uint32_t b)
{
    uint32_t v = r0;
    v = (b == 1) ? r1 : v;
    v = (b == 2) ? r2 : v;
    v = (b == 3) ? r3 : v;
    v = (b == 4) ? r4 : v;
    v = (b == 5) ? r5 : v;
    return v;
}

// ------------------------------------------------------------------------ compress kernel
//
// State between steps (all wave-uniform): P = next position to probe; chain = P is the end of a
// copy (the reference then inserts P-1 and probes P, snappy's "copy chain"); otherwise a scan is in
// progress and sc = index of its next probe (probe k sits at scan_base + off[k]); next_emit = start
// of the pending literal.
//
// DENSE WINDOW (sc <= 31: probes 0..31 of a scan advance by one byte).  Lane l takes position W0 + l.  All 64
// positions are hashed, looked up and compared against their candidate at once; which of them the
// serial parse actually visits is then a short scalar walk over the hit mask (a hit at lane j with
// local match length ml sends the parse to lane j+ml, a miss to lane j+1), and the elements of the
// window are packed by a wave prefix sum.  The lookups are exact as long as no two INSERTED lanes
// share a bucket; that is verified after the fact (commit, read back) and repaired by a redo.
//
// STRIDED BATCH (sc >= 32: the skip heuristic has accelerated).  Lane k takes the k-th future probe
// of the schedule; the first hit wins and probes behind it are rolled back.

// The hash table of one block, in one of two homes:
//  * LDS, 16384 x uint16, zeroed per block (32 KiB: four of them fit a CU, one per SIMD);
//  * global memory (L2 / Infinity Cache), the same 16384 x uint16, NEVER cleared (round 5): which buckets a block has
//    written is kept in a 2 KiB bitmap in LDS, the others read as empty.  Used by the lone waves of the team kernel
//    and the extra waves of the multi-wave kernel, for which no LDS table is left.  (Round 1 had uint32 entries tagged
//    with a generation, rounds 2-4 a zeroing pass per block; half the footprint was worth more than the tags.)
// NOT volatile: volatile LDS accesses are not address-space-inferred and compile to FLAT
// loads/stores with a vmcnt(0) wait after each.  The places that need "really re-read memory"
// (other lanes of the same store may have won the bucket) use a compiler barrier.
static constexpr int kGlobBmLog2 = 13;  // bits of the bucket number that index a global-table wave's scratch bitmap
template <bool in_lds>
struct HashTab {
    uint16_t *lds;
    uint16_t *glob;
    uint32_t *bm;   // global home only: 8192-bit scratch bitmap in LDS (all zero between uses), over 13 bits of the bucket number
    uint32_t *occ;  // global home only: 16384-bit bitmap in LDS, bit h = bucket h has been written in THIS block
    static constexpr bool kInLds = in_lds;

    // (relaxed atomics = plain loads/stores that the compiler may neither forward nor merge: other
    // lanes of this wave write the same entries.  LDS: wave scope, the LDS executes a wave's accesses in
    // order.  Global memory: the look-up is an AGENT-scope load -- `global_load_ushort ... sc1`, served by
    // the L2 and never by a line the CU's vector L1 may still hold from before this wave's store; the
    // wave's stores (write-through) and loads of one address reach the L2 in issue order.  Measured
    // against the wave-scope load: no difference in rate (these lines miss the L1 anyway: 8 x 32 KiB of
    // tables per CU).  tests/test_abi.py checks the instruction.)
    // Global home (round 5): a bucket nobody has written in this block is EMPTY -- it reads as position 0, as in
    // the reference -- and which buckets have been written is kept in 2 KiB of LDS (`occ`).  A look-up goes to the
    // table in memory only for the lanes whose bucket is occupied (a tile of text looks up 64 positions of which
    // ~14 are probed; a third to a half of them find an empty bucket: that many fewer 64-byte reads over the fabric,
    // where every look-up of these tables is a miss in the L2), and the table in memory is never cleared: what an
    // earlier block left in a bucket is not read before this block has written the bucket.
    // (`wanted` = false: a lane whose look-up nobody will use -- it reads as an empty bucket, without a request)
    __device__ __forceinline__ uint32_t get(uint32_t h, bool wanted = true) const
    {
        if (in_lds)
            return __hip_atomic_load(lds + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t w = __hip_atomic_load(occ + (h >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t v = 0;
        if (((w >> (h & 31u)) & 1u) != 0 && wanted)
            v = __hip_atomic_load(glob + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    }
    __device__ __forceinline__ void put(uint32_t h, uint32_t v) const
    {
        if (in_lds) {
            __hip_atomic_store(lds + h, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            __hip_atomic_store(glob + h, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_or(occ + (h >> 5), 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    // (a2) table of T = 2^log2T entries, all empty
    __device__ __forceinline__ void clear(uint32_t log2T, uint32_t lane) const
    {
        u32x4 z = {0, 0, 0, 0};
        if (in_lds) {
            const uint32_t nvec = (1u << log2T) >> 3;  // 16-byte chunks
            for (uint32_t i = lane; i < nvec; i += 64)
                reinterpret_cast<u32x4 *>(lds)[i] = z;
        } else {
            reinterpret_cast<u32x4 *>(occ)[lane] = z;  // 2 KiB: no bucket is occupied
            reinterpret_cast<u32x4 *>(occ)[64 + lane] = z;
        }
    }
    // global home: the scratch bitmap's word and bit of bucket h (13 of its 14 bits: two buckets share a bit, and a
    // lane that finds its bit set by the other bucket is told apart by the full hashes where the cut is worked out)
    __device__ __forceinline__ uint32_t *bm_word(uint32_t h) const { return bm + ((h & ((1u << kGlobBmLog2) - 1u)) >> 5); }
};
static constexpr uint32_t kGlobBmWords = (1u << kGlobBmLog2) / 32u;  // scratch bitmap of a global-table wave: 8192 bits
static constexpr uint32_t kGlobOccWords = 512;  // occupancy bitmap: 16384 bits
static constexpr uint32_t kGlobLdsWords = kGlobBmWords + kGlobOccWords;  // per global-table wave: 3 KiB of LDS
// (team kernel) behind them, per lone wave: the adoption request, the wave's state as its team sees it, the record handed over
static constexpr uint32_t kLoneReqWord = kGlobLdsWords, kLoneStateWord = kGlobLdsWords + 1, kLoneRecWord = kGlobLdsWords + 2;
static constexpr uint32_t kLoneLdsWords = kGlobLdsWords + 8;

// ------------------------------------------------------------------------ wave teams
//
// TEAM KERNEL (round 3): kWaves wavefronts of one workgroup compress ONE block together and share ONE hash
// table in LDS -- no table leaves the CU.  The greedy parse is serial, so what the waves share is a
// pipeline, not the data: the parse state travels from wave to wave as a TOKEN (one 64-bit LDS word:
// position, chain flag, scan index, pending literal start, turn number) and each wave in turn runs the
// serial part of one tile (stale check of its lookups, walk, table commit).  Everything else a tile costs --
// the element writer of the tile it has just parsed and the head of its next tile (bytes, hashes, table
// lookups, candidate loads, compares) -- runs while the other waves hold the token.  Lookups made ahead of
// the token are speculative: they are re-read from the table once the token arrives, and only the lanes
// whose bucket was written in between get their candidate again (same rule as the single-wave kSpec path;
// tests/tile_model.py).  The output position travels on a second token behind the first (a writer needs to
// know where the elements of the tile before end, nothing else).
struct NoTeam {
    static constexpr int kWaves = 1;
};
template <int W>
struct WaveTeam {
    static constexpr int kWaves = W;
    uint64_t *ptok;  // LDS: parse token
    uint64_t *otok;  // LDS: output-position token
    uint32_t wave;   // this wave's place in the team (0 starts every block)
    unsigned long long *trace;  // (-DSZ_TRACE builds: this wave's stamps for this block, or null)
    uint8_t *scr;    // LDS: 256 bytes of scratch of this wave's own (lanes that share a bucket inside a tile)
    uint32_t *bar;   // LDS: arrivals at the team's barrier (the workgroup may hold waves that are not of the team)
    // ADOPTION (round 6; see "the end of a launch" at the team kernel; the instance with AdoptT = TeamAdopt): the block is
    // one that a lone wave of this workgroup has handed over in mid-parse -- its parse state is in the kAdoptRec* words
    // at `arec` (LDS), its hash table has been copied into the team's by the kernel
    const uint32_t *arec;
};
// What a lone wave does about adoption inside compress_one_block: nothing (every kernel but the team kernel), or it
// looks at `req` once per tile of the fast dense path and, asked to, hands the block over where a tile ends.
struct NoAdopt {
    static constexpr bool kLone = false, kResume = false;
};
struct TeamAdopt {  // a team goes on with a block that a lone wave has handed over
    static constexpr bool kLone = false, kResume = true;
};
struct LoneAdopt {  // (the request word and the record lie behind the wave's bitmaps in LDS, at kLoneReqWord / kLoneRecWord
                    // from `tab.occ`: no pointer of their own to keep in a register file that has none to spare)
    static constexpr bool kLone = true, kResume = false;
};
static constexpr uint32_t kAdoptIdle = 0, kAdoptAsked = 1, kAdoptTaken = 2, kAdoptWithdrawn = 3;  // a request's life
static constexpr uint32_t kLoneRunning = 0, kLoneLeft = 1, kLoneHanded = 2, kLoneAdopted = 3;     // a lone wave's state, as its team sees it
static constexpr uint32_t kAdoptRecBlock = 0, kAdoptRecState = 1, kAdoptRecEmit = 2, kAdoptRecOp = 3;  // (words of the record, from kLoneRecWord)
static constexpr uint32_t kAdoptMinBlock = 16384;  // blocks this long or longer are handed over (their table has all 16384 entries) ...
static constexpr uint32_t kAdoptMinLeft = 4096;    // ... while this much of them is still to be parsed
static constexpr uint32_t kTeamDoneTurn = 0x3fffffu;  // turn field of a token: the block is finished
static constexpr uint32_t kTeamNotMine = 0xfffffffcu; // return value of the waves that did not finish the block
static constexpr uint32_t kTeamLost = 0xfffffffbu;    // return value of a wave that gave up waiting for its team (never stored: the kernel poisons the block)
#ifndef SZ_TEAM_SPIN_CAP
#define SZ_TEAM_SPIN_CAP (1u << 24)  // polls before a waiting wave gives the block up (seconds: only a lost wave gets there)
#endif
static constexpr int kTeamBasePrio = 2;  // issue priority of a team's waves off the token's path (the lone waves of a workgroup have 0; round 5: 2 / 3 against 1 / 2: low-entropy +2.7 %, text and mix +0.4 %)
static constexpr int kTeamPollSleep = 3;  // s_sleep units (64 clocks) between two polls of a token
static constexpr int kTeamTokenPrio = 3;  // issue priority of the wave that holds the parse token
static constexpr int kTeamPollBackoffMult = 4;
static constexpr int kTeamPollBackoff = 8;  // polls after which a waiting wave polls four times less often (0: never)

__device__ __forceinline__ uint64_t team_pack_parse(uint32_t turn, uint32_t P2, uint32_t chain2, uint32_t sc2,
                                                    uint32_t pm1, uint32_t next_emit)
{
    return (uint64_t)(P2 | (chain2 << 17) | (sc2 << 18) | (pm1 << 24)) | ((uint64_t)next_emit << 25) |
           ((uint64_t)turn << 42);
}
// every lane stores the same value to the same address (one LDS write); the table commits of this wave are
// ahead of it in the LDS queue and the wait makes that explicit
__device__ __forceinline__ void team_publish(uint64_t *slot, uint64_t v)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_wakeup" ::: "memory");
}
// Barrier among the kWaves waves of a team (s_barrier would wait for every wave of the workgroup, and the hybrid
// kernel's workgroups hold waves that compress other blocks): a counter of arrivals that only grows; a wave waits
// until the count has passed the multiple of kWaves its own arrival belongs to.  LDS accesses of a wave made
// before its arrival are complete before it (the wait), those of the others are visible after it.
template <int W, bool kWatchDead>
__device__ __forceinline__ uint32_t team_barrier(uint32_t *bar, const uint32_t *dead, uint32_t lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t mine = 0;
    if (lane == 0)
        mine = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    mine = uni(mine);
    const uint32_t target = (mine / W + 1u) * W;
    uint32_t lost = 1;  // the cap reached, or the team given up by another wave: the caller leaves the persistent loop
    for (uint32_t spin = 0; spin < SZ_TEAM_SPIN_CAP; spin++) {
        const uint32_t now = uni(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int32_t)(now - target) >= 0) {
            lost = 0;
            break;
        }
        if (kWatchDead && uni(__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0)
            break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return lost;
}

// waits until the token in `slot` carries `turn` (or "done"); returns the token; lost = 1: gave up
__device__ __forceinline__ uint64_t team_wait(const uint64_t *slot, uint32_t turn, uint32_t &lost)
{
    // (flags are integers and every value is brought back to scalar registers at once: a boolean that the
    // compiler keeps as a lane mask, or a counter it keeps in a vector register, costs the loop a round trip
    // between the vector and the scalar unit per poll -- ~25 clocks each, tools/isacost.hip)
    uint64_t v = 0;
    uint32_t spin = 0;
    lost = 0;
    for (;;) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
        v = ((uint64_t)hi << 32) | lo;
        const uint32_t tt = hi >> 10;
        if (tt == turn)
            break;
        if (tt == kTeamDoneTurn)
            break;
        spin = uni(spin + 1u);
        if (spin > SZ_TEAM_SPIN_CAP) {
            lost = 1;
            break;
        }
        // (a wave that has polled in vain a few times is waiting for something long -- the holder follows a long
        // match, or runs the general code --: it polls four times less often from then on, its issue slots are the
        // other waves')
        if (spin > (uint32_t)kTeamPollBackoff)
            __builtin_amdgcn_s_sleep(kTeamPollBackoffMult * kTeamPollSleep);
        else
            __builtin_amdgcn_s_sleep(kTeamPollSleep);
    }
    asm volatile("" ::: "memory");  // nothing of what follows is read before the token
    return v;
}

// One block by one wavefront.  kChunks = 1 KiB chunks of the block kept in registers.
// kFast: with the fast dense path (compressible data).  The single kernel's instance for launches that
// a sample found incompressible is built without it: its presence alone costs that instance 2 % (register
// allocation of the strided path), and such blocks do not use it.
template <int kBlkChunks, bool kLean, typename Tab, bool kFast = true,
          int kMatchPieces = kMatchPiecesDefault, typename TeamT = NoTeam, typename AdoptT = NoAdopt>
__device__ __forceinline__ uint32_t compress_one_block(const Tab &tab, const uint8_t *__restrict__ in,
                                                       uint32_t n, uint8_t *__restrict__ out, uint32_t flags,
                                                       uint32_t lane, const TeamT &team = TeamT{},
                                                       const AdoptT & = AdoptT{})
{
    PE_START();
    constexpr int kTeamW = TeamT::kWaves;  // 1: this wave compresses the block alone
    // team mode: wave 0 starts the block (it holds the parse token of turn 0 and the output position);
    // the others wait for a token.  Blocks too short for the fast dense path are wave 0's alone.
    bool holder = true, own_op = true, team_gone = false, team_lost = false;
    uint32_t myturn = 0;
    TRACE_DECL;
    if constexpr (kTeamW > 1) {
        holder = own_op = team.wave == 0;
        myturn = team.wave;
        if (n < 1024 && team.wave != 0)
            return kTeamNotMine;
    }
    // (a8) varint32(n), n <= 65536 -> at most 3 bytes.  (`op` is a select of wave-uniform values, not a
    // value merged behind the per-lane `if`: the compiler keeps the whole parse state in scalar
    // registers only while it can prove it uniform -- `opt -passes='print<uniformity>'`.)
    const uint32_t nb = (flags & kFlagBlockVarint) ? (n < 128 ? 1u : (n < 16384 ? 2u : 3u)) : 0u;
    constexpr bool adopted = AdoptT::kResume;  // team mode: the block comes from a lone wave in mid-parse (its preamble is written, its table is taken over)
    static_assert(!adopted || kTeamW > 1, "only a team adopts");
    if constexpr (kTeamW > 1) {
        if (holder && lane < nb && !adopted)
            out[lane] = (uint8_t)(((n >> (7 * lane)) & 0x7f) | (lane + 1 < nb ? 0x80 : 0));
        UNIFORM_JOIN();
    } else {
        if (lane < nb)
            out[lane] = (uint8_t)(((n >> (7 * lane)) & 0x7f) | (lane + 1 < nb ? 0x80 : 0));
    }
    uint32_t op = nb;
    if (n == 0)
        return op;

    uint32_t next_emit = 0;
    bool poisoned = false;
    if (n >= kInputMargin) {
        const uint32_t ip_limit = n - kInputMargin;  // positions <= n-16 may be probed

        // probe schedule in registers: lane l holds off[64*j + l] and off[64*j + l + 1]
        // (lean variant, for the multi-wave kernel's compressible data: the schedule stays in memory and
        // the first scan gets no prefetch, which frees ~20 registers per lane for a higher occupancy)
        const uint32_t *po = g_probe.off + (kLean ? 0u : lane);
        const uint32_t Pr1 = po[64], Pr2 = po[128], Pr3 = po[192], Pr4 = po[256], Pr5 = po[320];
        const uint32_t Pn1 = po[65], Pn2 = po[129], Pn3 = po[193], Pn4 = po[257], Pn5 = po[321];
        const uint32_t Pr0b = po[32], Pn0b = po[33];  // probes 32..63 in lanes 0..31 (first strided batch)
        // The first scan of a block starts at 1: fetch the shingles of its far probe positions now
        // (one HBM latency for the whole block instead of one per batch when nothing matches).
        uint32_t pv0b = 0, pv1 = 0, pv2 = 0, pv3 = 0, pv4 = 0;
        // Bytes of the dense windows.  Lane l of a window starting at W needs bytes [W+l, W+l+32): 64
        // overlapping 32-byte strings out of ~100 distinct bytes.  Loading them per lane costs the L1
        // ~190 tag lookups per window (the vector-memory pipeline is what bounds compressible data:
        // profiles/r01_text_pmc.txt), so instead 256-byte stretches of the block are kept in
        // registers -- buffer k = bytes [128k, 128k+256), one dword per lane, fetched by one coalesced
        // load, the next one always prefetched -- and each window is cut out of its buffer with lane
        // shuffles (ds_bpermute: LDS crossbar, no memory access) and v_alignbyte.
        const uint32_t last16 = n - 16;  // n >= 15 here; (n == 15: no position is probed at all)
        // An empty bucket reads as position 0 (as in the reference), so a probe whose bucket is empty is compared with
        // the block's first four bytes: they are kept in a scalar register, and a batch of probes in which no lane has
        // a real candidate and no lane's shingle equals them -- every batch of an incompressible block -- does without
        // its candidate loads, i.e. without a memory round trip of its own (an incompressible 64 KiB block had 7
        // dependent round trips, ~3 us each under load; about two are left: C3 +2 %, EXPERIMENTS.md R4.2).
        const bool zok = n >= 64;
        uint32_t z0v = 0;  // (every lane the same four bytes; brought to a scalar register where it is used: a
                           // readfirstlane right here would make the wave wait for the load before it requests the block)
        if (zok)
            z0v = ld32(in);
        u32x4 a = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
        uint32_t wcur = 0, wnxt = 0, wnxt_sh = 0, kcur = 0;  // buffers kcur and kcur + 1
        // (the prefetched buffer stays as loaded until it becomes the current one: the shift that puts
        // the last bytes of a block where they belong is applied then, not behind the load -- a use
        // right behind the load would make the wave wait for it)
        auto fetch_buf = [&](uint32_t k, uint32_t &sh8) -> uint32_t {
            const uint32_t o = (k << 7) + 4u * lane;
            const uint32_t oc = o < n - 4u ? o : n - 4u;  // stay inside the block ...
            const uint32_t sh = o - oc < 3u ? o - oc : 3u;
            sh8 = 8u * sh;                                 // ... and keep its last bytes where they belong
            return ld32(in + oc);
        };
        auto load_buf = [&](uint32_t k) -> uint32_t {
            uint32_t sh8;
            const uint32_t v = fetch_buf(k, sh8);
            return v >> sh8;
        };
        auto advance_bufs = [&](uint32_t k) {  // buffers k and k + 1 become current / prefetched
            wcur = k == kcur + 1 ? wnxt >> wnxt_sh : load_buf(k);
            wnxt = fetch_buf(k + 1, wnxt_sh);
            kcur = k;
        };
        auto cut_window = [&](uint32_t W) {  // a, a2 := bytes [W+lane, W+lane+32); needs kcur == W >> 7
            const uint32_t s0 = (W & 127u) + lane;  // <= 190: dwords s0/4 .. s0/4+8 <= 55 of the buffer
            const uint32_t i0 = (s0 >> 2) << 2, sh = s0 & 3u;
            uint32_t d[9];
#pragma unroll
            for (int k = 0; k < 9; k++)
                d[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i0 + 4 * k), (int)wcur);
            a.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
            a.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
            a.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
            a.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
            a2.x = __builtin_amdgcn_alignbyte(d[5], d[4], sh);
            a2.y = __builtin_amdgcn_alignbyte(d[6], d[5], sh);
            a2.z = __builtin_amdgcn_alignbyte(d[7], d[6], sh);
            a2.w = __builtin_amdgcn_alignbyte(d[8], d[7], sh);
        };
        auto load_window = [&](uint32_t W) {
            const uint32_t k = W >> 7;
            if (k != kcur)
                advance_bufs(k);
            cut_window(W);
        };
        if (kTeamW > 1 && !holder) {
            kcur = 0xfffffffeu;  // no buffer yet: the first head of this wave loads its own
        } else if (n >= 512) {
            // (buffer 0 lies inside the block: no shift, so nothing here uses the loaded value -- `load_buf` would,
            // and the wave would wait a round trip for it before it requests the block)
            wcur = ld32(in + 4u * lane);
            wnxt = fetch_buf(1, wnxt_sh);
        } else if (n >= 16) {
            wcur = load_buf(0);
            wnxt = fetch_buf(1, wnxt_sh);
        }

        // Whole-block register prefetch (blocks >= 16 KiB), issued right behind the probe loads above
        // (measured: waiting for those to land first makes HBM reads exactly 1.00x the block but costs
        // 18% of the C3 rate; streaming at once re-fetches the ~270 probed lines): with one wave per SIMD the register file
        // is otherwise idle, so all HBM reads of the block are put in flight now and the match
        // finding below runs under them.  If the block turns out to start with a long literal (it
        // does when the data is incompressible), that literal is stored straight from these
        // registers.  Chunk c = 64*u + lane holds literal bytes [bhead + 16c, +16), laid out so that
        // the stores are 16-byte aligned for the predicted position of the literal (3-byte tag).
        const bool use_blk = kBlkChunks > 0 && n >= 16384;
        const uint32_t bhead = (16u - (uint32_t)((uintptr_t)(out + op + 3) & 15u)) & 15u;
        uint32_t nchunk = use_blk ? (n - bhead) >> 10 : 0;  // whole 1 KiB chunks (wave-uniform)
        nchunk = nchunk < (uint32_t)kBlkChunks ? nchunk : (uint32_t)kBlkChunks;
        u32x4 blk[kBlkChunks > 0 ? kBlkChunks : 1];
        {
            const uint8_t *bp = in + bhead + 16u * lane;
#pragma unroll
            for (int u = 0; u < kBlkChunks; u++) {
                blk[u] = u32x4{0, 0, 0, 0};
                if ((uint32_t)u < nchunk)
                    // (plain loads, not non-temporal ones: the lines the probe loads below touch are then
                    // served by the same fetch instead of being fetched twice -- C3 0.86 -> 0.82 ms)
                    blk[u] = ld128(bp + 1024 * u);
            }
        }
        const uint32_t blk_op = op;  // the literal must start here for the prediction to hold
        PE_LAP(28);

        // (BEHIND the block stream, and behind the wait for it: the same five gathers requested AHEAD of the stream cost
        // 27 % of the incompressible rate, twice measured -- each touches up to ten pages that nothing has touched yet,
        // and the stream queues behind their address translation; behind it they hit translated pages and lines in the L2)
        if (!kLean) {
            if (lane < 32 && 1 + Pn0b <= ip_limit) pv0b = ld32(in + 1 + Pr0b);
            if (1 + Pn1 <= ip_limit) pv1 = ld32(in + 1 + Pr1);
            if (1 + Pn2 <= ip_limit) pv2 = ld32(in + 1 + Pr2);
            if (1 + Pn3 <= ip_limit) pv3 = ld32(in + 1 + Pr3);
            if (1 + Pn4 <= ip_limit) pv4 = ld32(in + 1 + Pr4);
        }

        // (a2) zeroed uint16 table, T = clamp(pow2 >= n, 256, 16384)
        uint32_t log2T = 8;
        while (log2T < 14 && (1u << log2T) < n)
            log2T++;
        const uint32_t shift = 32 - log2T;
        if constexpr (kTeamW > 1) {
            if (n >= 1024) {  // the whole team: a slice each, then every wave sees an empty table
                if constexpr (!adopted) {  // (an adopted block: the kernel has copied its lone wave's table into this one)
                const uint32_t nvec = (1u << log2T) >> 3;
                for (uint32_t i = team.wave * 64u + lane; i < nvec; i += 64u * kTeamW)
                    reinterpret_cast<u32x4 *>(tab.lds)[i] = u32x4{0, 0, 0, 0};
                }
                // (a lost barrier: the table is not known to be clear -- the main loop is left at its first test and
                // the wave returns kTeamLost, the kernel poisons the block)
                team_lost = uni(team_barrier<kTeamW, false>(team.bar, nullptr, lane)) != 0;
                team_gone = team_lost;
                poisoned = team_lost;
            } else {
                tab.clear(log2T, lane);
                UNIFORM_JOIN();  // (the per-lane loop's join is not this arm's join with the one above: team_gone stays provably uniform)
            }
        } else {
            tab.clear(log2T, lane);
        }

        // literal emission; a long block-prefix literal is only accounted for here and stored from
        // the prefetched registers after the main loop (one copy of that code, at a known address)
        uint32_t big_len = 0;
        auto emit_lit = [&](uint32_t from, uint32_t len) {
            if (use_blk && from == 0 && op == blk_op && len >= 4096) {
                big_len = len;
                op += 3 + len;
            } else {
                op = emit_literal<(kLean ? 0 : 16)>(out, op, in + from, len, lane);
            }
        };

        // Head of the next dense window (it starts at Wn; chainN: lane 0 is the end of a copy), run one
        // step ahead: its bytes, hash, the insert of P-1 and the table lookup are issued as soon as the
        // window's start is known, and complete under the stores of the elements of the window before.
        uint32_t hN = 0, candN = 0;
        auto prep = [&](uint32_t Wn, bool chainN) {
            const uint32_t firstN = chainN ? 1u : 0u;
            if (Wn + firstN >= ip_limit)
                return;  // nothing is probed there: the main loop ends
            load_window(Wn);
            const uint32_t posN = Wn + lane;
            hN = (a.x * kHashMul) >> shift;
            if (chainN && lane == 0)
                tab.put(hN, posN);  // the reference inserts P-1 before it looks P up
            candN = 0;
            if (posN < ip_limit && lane >= firstN)
                candN = tab.get(hN);  // table as of the start of the window
            UNIFORM_JOIN();
        };
        // (requested where the next window's start becomes known, run at the top of the general dense
        // window: the fast path below has its own, tile-based head)
        uint32_t prepW = 1;
        bool prepC = false, prepped = false;
        auto want_prep = [&](uint32_t Wn, bool chainN) {
            prepW = Wn;
            prepC = chainN;
            prepped = false;
        };

        // fast dense path: constants per lane, and "the general code takes the next window"
        const uint32_t lane4 = (lane >> 2) << 2, lane3 = lane & 3u;
        const uint32_t run32 = lane <= 32 ? 32u : 0xffffffffu;
        bool hold_fast = false;

        uint32_t P = 1;          // next position to probe
        bool chain = false;      // P ends a copy: insert P-1, probe P
        uint32_t sc = 0;         // scan: index of the next probe
        uint32_t scan_base = 1;  // scan: position of probe 0 (skip == 32 there)
        bool first_scan = !kLean;  // pv[] belongs to the scan that starts at 1
        if constexpr (kTeamW > 1) {
            if constexpr (adopted) {
                // an adopted block: the state its lone wave left the fast dense path with, exactly as that path hands a
                // state to the general code (every wave of the team reads it; it matters to wave 0, which holds the token)
                const uint32_t st = uni(team.arec[kAdoptRecState]);
                P = st & 0x1ffffu;
                chain = ((st >> 17) & 1u) != 0;
                sc = (st >> 18) & 63u;
                next_emit = uni(team.arec[kAdoptRecEmit]);
                op = uni(team.arec[kAdoptRecOp]);
                if (!chain)
                    scan_base = P - sc;
                prepW = P - (chain ? 1u : 0u);
                prepC = chain;
                first_scan = false;
            }
        }
        PE_LAP(29);
        PROF_DECL
        PROF_START();
        // (every trip of this loop moves P forward or ends it; the cap turns a state that does neither
        // -- memory that does not hold what this wave wrote -- into an error marker instead of a hang)
        uint32_t trips = 0;
        for (;;) {
            // The parse state is wave-uniform by construction; say so at every trip.  Inside a trip the
            // compiler keeps it in scalar registers only while it can PROVE it uniform, and the general
            // window code below has joins behind per-lane branches where it cannot (checked with
            // `opt -passes='print<uniformity>'`): without this the whole state, this trip's and every later
            // one's, lives in vector registers and the mask arithmetic runs as 64-bit vector instructions.
            P = uni(P); chain = unib(chain); sc = uni(sc); scan_base = uni(scan_base); next_emit = uni(next_emit);
            op = uni(op); kcur = uni(kcur); trips = uni(trips); poisoned = unib(poisoned); hold_fast = unib(hold_fast);
            prepW = uni(prepW); prepC = unib(prepC); prepped = unib(prepped); first_scan = unib(first_scan);
            big_len = uni(big_len);
            if (poisoned || ++trips > 2u * kMaxBlock) {
                poisoned = true;
                break;
            }
            ABLATION_BREAK_IF(flags & 0x200u);  // (-DSZ_ABLATION timing builds: no match finding at all, one literal)

            if constexpr (kTeamW > 1) {
                // (blocks under 1 KiB are wave 0's alone -- the others have left --: general code only)
                if (!holder || (n >= 1024 && chain && !hold_fast && P + kFastMargin <= ip_limit)) {
                    // =============================================== TEAM FAST DENSE PATH (fixed tiles, one TURN per tile)
                    // Same tiles, visits and masks as the single-wave fast dense path below (tests/tile_model.py);
                    // what differs is who does what when.  A turn = [token] -> stale check of the lookups made
                    // ahead -> visits (walk, exit state, table commit) -> [token passed on] -> element writer of
                    // this tile (waits for the output position of the tile before) -> head of this wave's next
                    // tile (kTeamW - 1 tiles past the one the next wave takes) -> wait.  A state the fast path does
                    // not take is kept by the wave that meets it: it runs the general code below (it owns the
                    // output position then) and comes back here, or ends the block.
                    first_scan = false;
                    uint32_t t = 0xffffffffu, e = 0, sc_in = 0, c_in = 1;
                    uint32_t a0 = 0, th = 0, tcand = 0, tmlw = 0, X = 0, Y = 0, Z = 0;
                    uint32_t Glo = 0, Ghi = 0;  // lanes of a shared bucket: the EARLIER lanes of this tile with this lane's hash
                    uint64_t Hraw = 0, farmask = 0;
                    u32x4 na = {0, 0, 0, 0}, na2 = {0, 0, 0, 0};
                    // The head of a tile (its bytes, hashes, table lookups, candidate strings, compares, jumps, shared
                    // buckets) is made in ONE place, at the top of the turn loop: ahead of the token for the tile this
                    // wave expects next, or -- the parse went elsewhere, or this wave comes from the general code -- with
                    // the token in hand for the tile of P2.
                    uint32_t have_head = 0, need_head = holder ? 1u : 0u;  // (flags as integers: see team_wait)
                    uint32_t tpois = 0;  // poison met inside this path (folded into `poisoned` where the path is left)
                    uint32_t thead = P >> 6;
                    uint32_t P2 = P, sc2 = 0, chain2 = 1, pm1_done = 0, quit = 0;
                    const uint32_t lb_lo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu;  // bits of the lanes below this one
                    const uint32_t lb_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
                    // Bytes of the block near the tile in registers (one dword per lane, one coalesced load): buffer of
                    // tile tt = bytes [64 tt - 128, 64 tt + 128) -- the tile + 32 for its own strings, and the two tiles
                    // before it, where the candidates are that a lookup made ahead of the token can have missed.
                    uint32_t tbuf = 0, tb_base = 0, tb_tile = 0xffffffffu;
                    uint32_t tnbuf = 0, tn_sh = 0, tn_tile = 0xffffffffu;  // prefetched (as loaded; shift applied when it becomes current)
                    auto tb_fetch = [&](uint32_t tt, uint32_t &sh8) -> uint32_t {
                        const uint32_t o = (tt >= 2 ? (tt << 6) - 128u : 0u) + 4u * lane;
                        const uint32_t oc = o < n - 4u ? o : n - 4u;  // stay inside the block ...
                        const uint32_t sh = o - oc < 3u ? o - oc : 3u;
                        sh8 = 8u * sh;                                 // ... and keep its last bytes where they belong
                        return ld32(in + oc);
                    };
                    // Where the parse goes from each lane, with what the exit of a visit needs to know about the last
                    // jump beside it, so that the walk's last v_readlane IS the exit state:
                    //   X: bits 0..5 target lane (next hit at or after this lane, then past its match), bit 6 the jump
                    //      leaves the tile (bits 0..5 then name the lane itself), bit 7 no hit from here on (the scan stays open), bit 8 that hit's match reaches 32
                    //      bytes (long), bits 10..15 lane of that hit, bits 16..23 where the parse stands afterwards
                    //      relative to the tile;
                    //   Y: the state a visit that ends with this lane's jump hands on, as it travels in the token
                    //      (position, chain flag, "copy ends with the tile", scan index); Z: the pending literal's start.
                    auto jumps = [&]() {
                        const uint64_t tl = Hraw >> lane;
                        uint32_t dlo, dhi;
                        asm("v_ffbl_b32 %0, %1" : "=v"(dlo) : "v"((uint32_t)tl));
                        asm("v_ffbl_b32 %0, %1" : "=v"(dhi) : "v"((uint32_t)(tl >> 32)));
                        const uint32_t dist = umin3(dlo, dhi | 32u, 64u);  // 64: no hit from here on
                        const uint32_t nh_l = lane + dist;
                        const uint32_t m_nh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nh_l << 2), (int)tmlw);
                        const bool open = dist >= 64u;
                        const bool lng = !open && m_nh == 64u;
                        const uint32_t sum = nh_l + m_nh;
                        uint32_t tgt = sum > lane ? sum : lane + 1;  // (always so for a hit: ml >= 4; keeps the walk finite whatever memory holds)
                        tgt = (open || tgt > 64u) ? 64u : tgt;
                        const uint32_t p2off = open ? 64u : (lng ? nh_l : sum);
                        const uint32_t base_ = t << 6;
                        // (a lane whose jump leaves the tile points at itself, bit 6 set: the walk below may then take
                        // steps past its end -- they change nothing -- and needs no branch per element)
                        X = (tgt >= 64u ? lane | 64u : tgt) | (open ? 1u << 7 : 0u) | (lng ? 1u << 8 : 0u) | ((nh_l & 63u) << 10) |
                            ((p2off & 0xffu) << 16);
                        Y = (base_ + p2off) | (open ? 0u : 1u << 17) | ((!open && !lng && sum == 64u) ? 1u << 18 : 0u) |
                            (open ? (63u - lane) << 19 : 0u);
                        Z = open ? base_ + lane : base_ + p2off;
                        // lanes from which a scan could reach probe 32 inside the tile (32 hit-free lanes in a row)
                        farmask = ballot64(dist >= run32);
                    };
                    for (;;) {  // ================================================ one turn per trip
                        P2 = uni(P2); sc2 = uni(sc2); chain2 = uni(chain2); pm1_done = uni(pm1_done); quit = uni(quit);
                        holder = unib(holder); own_op = unib(own_op); myturn = uni(myturn); next_emit = uni(next_emit);
                        op = uni(op); tpois = uni(tpois); have_head = uni(have_head); need_head = uni(need_head);
                        t = uni(t); thead = uni(thead); tb_base = uni(tb_base); tb_tile = uni(tb_tile); tn_tile = uni(tn_tile);
                        TRACE(0); MARK("T0");
                        if (need_head) {
                            // ---- HEAD of tile thead
                            const uint32_t tt = thead;
                            if (tt != tb_tile) {
                                if (tt == tn_tile) {
                                    tbuf = tnbuf >> tn_sh;
                                } else {
                                    uint32_t sh8;
                                    const uint32_t v = tb_fetch(tt, sh8);
                                    tbuf = v >> sh8;
                                }
                                tb_tile = tt;
                                tb_base = tt >= 2 ? (tt << 6) - 128u : 0u;
                            }
                            {
                                const uint32_t i0 = ((tt << 6) - tb_base) + lane4;
                                uint32_t d[9];
#pragma unroll
                                for (int q = 0; q < 9; q++)
                                    d[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i0 + 4 * q), (int)tbuf);
                                na.x = __builtin_amdgcn_alignbyte(d[1], d[0], lane3);
                                na.y = __builtin_amdgcn_alignbyte(d[2], d[1], lane3);
                                na.z = __builtin_amdgcn_alignbyte(d[3], d[2], lane3);
                                na.w = __builtin_amdgcn_alignbyte(d[4], d[3], lane3);
                                na2.x = __builtin_amdgcn_alignbyte(d[5], d[4], lane3);
                                na2.y = __builtin_amdgcn_alignbyte(d[6], d[5], lane3);
                                na2.z = __builtin_amdgcn_alignbyte(d[7], d[6], lane3);
                                na2.w = __builtin_amdgcn_alignbyte(d[8], d[7], lane3);
                            }
                            th = (na.x * kHashMul) >> shift;
                            a0 = na.x;
                            t = tt;
                            tcand = tab.get(th);  // (ahead of the token: checked against the table when it comes)
                            const u32x4 nc = ld128(in + tcand);  // tcand < pos, pos + 32 <= n here
                            const u32x4 nc2 = ld128(in + tcand + 16);
                            if (((tt + kTeamW) << 6) < n) {  // this wave's likely tile after this one: its bytes are on their way
                                tnbuf = tb_fetch(tt + kTeamW, tn_sh);
                                tn_tile = tt + kTeamW;
                            }
                            {
                                // Lanes of this tile that share a bucket: G(lane) = the earlier lanes with this lane's hash.
                                // Two INSERTED lanes of one bucket end a visit before the second one (the later lane should
                                // have seen the earlier one); which lanes are inserted is only known on the token, but with
                                // these masks the test is two ANDs and a ballot there: no read-back of the table, nothing to
                                // roll back.  (256 bytes of scratch in LDS find the candidates: every lane writes its number
                                // under its hash's low byte and reads the entry back; a lane that reads another number shares
                                // its entry -- a shared bucket or just a shared low byte, told apart by the full hashes.)
                                Glo = 0;
                                Ghi = 0;
                                uint8_t *slot = team.scr + (th & 255u);
                                __hip_atomic_store(slot, (uint8_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                asm volatile("" ::: "memory");
                                const uint32_t r = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                uint64_t losers = ballot64(r != lane);
                                while (losers) {
                                    const uint32_t hv = lane_get(th, ctz64(losers));
                                    const bool mine = th == hv;
                                    const uint64_t G = ballot64(mine);
                                    losers &= ~G;
                                    Glo = mine ? (uint32_t)G & lb_lo : Glo;
                                    Ghi = mine ? (uint32_t)(G >> 32) & lb_hi : Ghi;
                                }
                            }
                            {  // candidates' strings against the lanes' own: hit mask, match lengths for the walk (4..31, or 64: goes on)
                                const uint32_t ml = matchlen32(na, na2, nc, nc2);
                                Hraw = ballot64(nc.x == na.x);
                                tmlw = ml == 32 ? 64u : ml;
                            }
                            jumps();
                            have_head = 1;
                            need_head = 0;
                        }
                        TRACE(10); MARK("T10");
                        if (!holder) {
                            uint32_t lost;
                            const uint64_t tok = team_wait(team.ptok, myturn, lost);
                            if ((lost | ((uint32_t)(tok >> 42) == kTeamDoneTurn ? 1u : 0u)) != 0) {
                                team_gone = true;
                                team_lost = lost != 0;
                                break;
                            }
                            P2 = (uint32_t)tok & 0x1ffffu;
                            chain2 = ((uint32_t)tok >> 17) & 1u;
                            sc2 = ((uint32_t)tok >> 18) & 63u;
                            pm1_done = ((uint32_t)tok >> 24) & 1u;
                            next_emit = (uint32_t)(tok >> 25) & 0x1ffffu;
                            holder = true;
                        }
                        TRACE(1); MARK("T1");
                        if (kTeamTokenPrio)
                            __builtin_amdgcn_s_setprio(kTeamTokenPrio);
                        // ---- enter the tile of P2
                        const uint32_t t2 = P2 >> 6, e2 = P2 & 63u;
                        if (__builtin_expect(((have_head ^ 1u) | (t2 ^ t)) != 0, 0)) {  // not the tile this wave has a head of: make it now
                            thead = t2;
                            need_head = 1;
                            have_head = 0;
                            continue;
                        }
                        have_head = 0;
                        const uint32_t acc_emit = next_emit;   // start of the literal that is open on entry (if any)
                        const bool acc_open = chain2 == 0;
                        const uint32_t base = t << 6;
                        const uint32_t pos = base + lane;
                        // The reference inserts P-1 before it looks P up; then the head's lookups are checked against the
                        // table as it stands now (the lanes the parse can still reach): a lane whose bucket was written
                        // since -- by the tiles in between, or by P-1 just now -- gets the new candidate.
                        if ((chain2 & (pm1_done ^ 1u)) != 0) {
                            // (every lane stores: lane e2 - 1 into its bucket, the others into a scratch halfword of this
                            // wave's -- no EXEC mask to set up, i.e. no trip to the scalar unit and back)
                            uint32_t hp = th;
                            if (__builtin_expect(e2 == 0, 0))
                                hp = (ld32(in + P2 - 1) * kHashMul) >> shift;
                            uint16_t *slot = (lane == ((e2 - 1u) & 63u)) ? tab.lds + hp : reinterpret_cast<uint16_t *>(team.scr + 254);
                            __hip_atomic_store(slot, (uint16_t)(P2 - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            asm volatile("" ::: "memory");
                        }
                        uint32_t newc = tab.get(th);
                        uint64_t sel = ballot64(newc != tcand) & ~below63(e2);
                        TRACE(2); MARK("T2");
                        ABLATION_PAD(flags, 0x4000u);
                        ABLATION_PAD_S(flags, 0x40000u);
                        uint64_t Vacc = 0, Hpacc = 0;
                        uint64_t Cacc = 0;  // lanes of this tile whose table entry this turn has written
                        uint32_t redo_head = 0;
                        uint32_t long_lane = 64;
                        e = e2;
                        c_in = chain2;
                        sc_in = sc2;
                        for (;;) {  // ============================================ visits of this tile
                            if (sel) {
                                // ---- PATCH: new candidates for the lanes in `sel`: their strings (out of the register buffer:
                                // no memory access on the token's path, unless a candidate lies before it), hit bit and match
                                // length merged in, jumps redone
                                const bool on = __builtin_amdgcn_inverse_ballot_w64(sel);
                                if (__builtin_expect((ballot64(newc < tb_base) & sel) != 0, 0)) {
                                    // a candidate that lies before the register buffer (the lookups were made long ago: this
                                    // wave came back from the general code, or the parse jumped): a fresh head of this tile
                                    // instead -- no memory access in the patch itself, so that nothing on the token's path ever
                                    // waits for vector memory (a wait here would also wait for the prefetch of the next buffer)
                                    thead = t;
                                    need_head = 1;
                                    redo_head = 1;
                                    break;
                                }
                                u32x4 c, c2;
                                {
                                    const uint32_t r = on ? newc - tb_base : 0u;  // (every lane takes part in the shuffles)
                                    const uint32_t i0 = (r >> 2) << 2, sh = r & 3u;
                                    uint32_t d[9];
#pragma unroll
                                    for (int q = 0; q < 9; q++)
                                        d[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i0 + 4 * q), (int)tbuf);
                                    c.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
                                    c.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
                                    c.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
                                    c.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
                                    c2.x = __builtin_amdgcn_alignbyte(d[5], d[4], sh);
                                    c2.y = __builtin_amdgcn_alignbyte(d[6], d[5], sh);
                                    c2.z = __builtin_amdgcn_alignbyte(d[7], d[6], sh);
                                    c2.w = __builtin_amdgcn_alignbyte(d[8], d[7], sh);
                                }
                                const uint32_t ml = matchlen32(na, na2, c, c2);
                                tcand = on ? newc : tcand;
                                tmlw = on ? (ml == 32 ? 64u : ml) : tmlw;
                                Hraw = (Hraw & ~sel) | (ballot64(c.x == na.x) & sel);
                                jumps();
                            }
                            TRACE(3); MARK("T3");
                            const uint64_t keep = ~below63(e);  // the lanes the parse can still reach
                            const uint64_t Hm = Hraw & keep;
                            uint32_t x = lane_get(X, e);
                            {
                                // ---- states left to the general code: a scan could reach probe 32 (32 hit-free lanes in
                                // a row, or the misses carried in plus those before the first hit)
                                const uint32_t lead = (x & 0x80u) ? 63u - e : ((x >> 10) & 63u) - e;
                                const uint32_t lead_bad = (c_in ^ 1u) & ((31u - (sc_in + lead)) >> 31);  // sc_in + lead > 31
                                const uint64_t far = farmask & keep;
                                if (__builtin_expect(((uint32_t)far | (uint32_t)(far >> 32) | lead_bad) != 0, 0)) {  // (P2, chain2, sc2 are this very state)
                                    hold_fast = true;
                                    quit = 1;
                                    break;
                                }
                            }
                            TRACE(11);
                            // ---- the serial part: one v_readlane per element; the last one is the exit state
                            // (four steps at a time, no branch in between: s_lshl_b64 and v_readlane use bits 0..5 of their
                            // lane operand; two wait states more before a v_readlane may select by what the one before read)
                            uint64_t S = 1ull << e;
                            do {
                                uint64_t bit;
                                asm volatile(
                                    "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                    "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                    "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                    "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1"
                                    : "+s"(S), "+s"(x), "=&s"(bit)
                                    : "v"(X)
                                    : "scc");  // (s_lshl_b64 and s_or_b64 write SCC)
                            } while (!(x & 64u));
                            TRACE(12);
                            const uint32_t s_l = x & 63u;  // the last start: the lane whose jump left the tile
                            uint64_t V, Hp, C;
                            uint32_t ll, emit2;
                            {
                                const uint32_t y = lane_get(Y, s_l), z = lane_get(Z, s_l);
                                const uint32_t open = (x >> 7) & 1u;
                                // the scan carried in found nothing: (s_l == e) as an integer, ((s_l ^ e) - 1) >> 31
                                const uint32_t carried = (((s_l ^ e) - 1u) >> 31) & (c_in ^ 1u) & open;
                                P2 = y & 0x1ffffu;
                                chain2 = (y >> 17) & 1u;
                                pm1_done = (y >> 18) & 1u;
                                sc2 = carried ? sc_in + 64u - e : (y >> 19) & 63u;
                                emit2 = carried ? next_emit : z;
                                ll = (x & 0x100u) ? (x >> 10) & 63u : 64u;
                                Hp = Hm & ~(Hm - S);  // first hit of every segment [start, next start)
                                V = (Hp << 1) - S;    // [start, hit] of every segment; an open last one runs to lane 63
                                C = V | ((S & ~(1ull << e)) >> 1) | ((uint64_t)pm1_done << 63);  // + the lane before each chain probe
                            }
                            TRACE(13);
                            {
                                // two inserted lanes of one bucket: the visit ends before the second one
                                const uint64_t conf = ballot64(((Glo & (uint32_t)C) | (Ghi & (uint32_t)(C >> 32))) != 0) & C;
                                if (__builtin_expect(conf != 0, 0)) {
                                    // the same exit for a visit that ends before lane `cut`: the last start below it, and
                                    // either its hit lies below the cut too (then nothing changes for it) or the scan is open there
                                    const uint32_t cut = ctz64(conf);
                                    tpois |= (e - cut) >> 31 ^ 1u;  // cut <= e (cannot happen: lane e has no earlier lane in C)
                                    const uint64_t bl = below63(cut);
                                    const uint64_t Sx = S & bl, Hm_t = Hm & bl;
                                    const uint32_t sl = 63u - (uint32_t)__builtin_clzll(Sx | 1ull);
                                    const uint32_t xs = lane_get(X, sl);
                                    const uint32_t open = ((xs >> 7) & 1u) | ((((xs >> 10) & 63u) - cut) >> 31 ^ 1u);  // no hit, or it lies at or past the cut
                                    const uint32_t carried = (((sl ^ e) - 1u) >> 31) & (c_in ^ 1u) & open;
                                    const uint32_t y = lane_get(Y, sl), z = lane_get(Z, sl);
                                    P2 = open ? base + cut : y & 0x1ffffu;
                                    chain2 = open ^ 1u;
                                    pm1_done = 0;
                                    sc2 = open ? (carried ? sc_in + cut - e : cut - sl - 1u) : 0u;
                                    emit2 = open ? (carried ? next_emit : base + sl) : z;
                                    ll = ((open ^ 1u) & (xs >> 8) & 1u) ? (xs >> 10) & 63u : 64u;
                                    Hp = Hm_t & ~(Hm_t - Sx);
                                    V = (((Hp << 1) | (open ? 1ull << cut : 0ull)) - Sx) & bl;
                                    C = V | ((Sx & ~(1ull << e)) >> 1);
                                }
                                TRACE(14);
                                if (__builtin_amdgcn_inverse_ballot_w64(C))
                                    tab.put(th, pos);  // (the lanes of C share no bucket now)
                                UNIFORM_JOIN();
                                Cacc |= C;
                            }
                            TRACE(15);
                            next_emit = emit2;
                            Vacc |= V;
                            Hpacc |= Hp;
                            long_lane = ll;  // (64: none)
                            if (__builtin_expect((tpois | (ll ^ 64u) | ((P2 >> 6) ^ t)) != 0, 1))
                                break;
                            // the same tile again (after a cut): insert P-1, then the lanes whose bucket this turn has
                            // written get the last writer as their candidate (known from the masks: no table access)
                            const uint32_t e2s = P2 & 63u;
                            if (chain2) {
                                if (lane == e2s - 1)
                                    tab.put(th, pos);
                                UNIFORM_JOIN();
                                Cacc |= 1ull << ((e2s - 1) & 63u);
                            }
                            {
                                const uint32_t mlo = Glo & (uint32_t)Cacc, mhi = Ghi & (uint32_t)(Cacc >> 32);
                                const uint32_t topl = mhi ? 63u - (uint32_t)__builtin_clz(mhi) : 31u - (uint32_t)__builtin_clz(mlo | 1u);
                                newc = base + topl;
                                sel = ballot64((mlo | mhi) != 0) & ballot64(newc != tcand) & ~below63(e2s);
                            }
                            e = e2s;
                            c_in = chain2;
                            sc_in = sc2;
                        }
                        if (__builtin_expect(redo_head != 0, 0))  // (only ever on the first visit: nothing of this turn is committed yet)
                            continue;
                        TRACE(4); MARK("T4");
                        // ---- a match that runs past its 32 local bytes: its end is the next state
                        uint32_t Lip = 0, Lcnd = 0, Lm = 0;
                        const uint32_t has_long = long_lane ^ 64u;  // (non-zero: there is one)
                        if (__builtin_expect(has_long != 0, 0)) {
                            Lip = base + long_lane;
                            Lcnd = lane_get(tcand, long_lane);
                            Lm = match_length<kMatchPieces>(in, Lcnd, Lip, n, lane, 32);
                            P2 = Lip + Lm;
                            next_emit = P2;
                            chain2 = 1;
                            sc2 = 0;
                            pm1_done = 0;
                        }
                        TRACE(5); MARK("T5");
                        // ---- the token goes on (or stays: the block's tail, a state for the general code, poison)
                        const uint32_t stop = quit | tpois | ((ip_limit - (P2 + kFastMargin)) >> 31);  // ... or P2 + kFastMargin > ip_limit
                        const bool cont = stop == 0;
                        if (cont)
                            team_publish(team.ptok, team_pack_parse(myturn + 1, P2, chain2, sc2, pm1_done, next_emit));
                        if (kTeamTokenPrio)
                            __builtin_amdgcn_s_setprio(kTeamBasePrio);
                        TRACE(6); MARK("T6");
                        // ---- element writer of this tile: where do its elements begin?
                        if (!own_op) {
                            uint32_t lost;
                            const uint64_t ot = team_wait(team.otok, myturn, lost);
                            op = (uint32_t)ot & 0x1ffffu;
                            tpois |= lost | (((uint32_t)ot >> 17) & 1u);
                            own_op = true;
                        }
                        TRACE(7); MARK("T7");
                        if (__builtin_expect(Hpacc != 0, 1)) {
                            const uint32_t jfirst = ctz64(Hpacc);
                            const uint32_t jlast = 63u - (uint32_t)__builtin_clzll(Hpacc);
                            uint64_t Lit = Vacc & ~Hpacc & below63(jlast);
                            if (acc_open) {
                                // a literal was open on entry: it ends at the first hit (one tag byte, at most 33
                                // bytes by construction); its bytes of the tiles before come from memory
                                const uint32_t len = base + jfirst - acc_emit;
                                const uint32_t nprev = base - acc_emit;
                                const uint32_t cur = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane - nprev) << 2), (int)a0);
                                uint32_t lbyte = cur;
                                if (lane < nprev && lane < len)
                                    lbyte = in[acc_emit + lane];
                                if (lane == 0)
                                    out[op] = (uint8_t)((len - 1) << 2);
                                if (lane < len)
                                    out[op + 1 + lane] = (uint8_t)lbyte;
                                UNIFORM_JOIN();
                                tpois |= (60u - len) >> 31;  // len > 60
                                op += 1 + len;
                                Lit &= ~below63(jfirst);
                            }
                            const uint64_t RS = Lit & ~(Lit << 1);
                            const uint64_t HC = has_long ? Hpacc & ~(1ull << (long_lane & 63u)) : Hpacc;
                            const bool is_lit = __builtin_amdgcn_inverse_ballot_w64(Lit);
                            const bool run_start = __builtin_amdgcn_inverse_ballot_w64(RS);
                            const bool is_hit = __builtin_amdgcn_inverse_ballot_w64(HC);
                            const uint32_t runlen = (uint32_t)__builtin_ctzll((Hpacc >> lane) | (1ull << 63));
                            const uint32_t abyte = a0 & 0xffu;
                            const uint32_t off = pos - tcand;
                            const bool short_form = tmlw < 12 && off < 2048;
                            uint32_t cnt = 0, word = 0;
                            if (is_lit) {
                                cnt = run_start ? 2 : 1;
                                word = run_start ? (((runlen - 1) << 2) | (abyte << 8)) : abyte;
                            } else if (is_hit) {
                                cnt = short_form ? 2 : 3;
                                word = short_form ? (1u | ((tmlw - 4) << 2) | ((off >> 8) << 5) | ((off & 0xffu) << 8))
                                                  : (2u | ((tmlw - 1) << 2) | (off << 8));
                            }
                            const uint32_t incl = wave_inclusive_sum(cnt);
                            uint8_t *o = out + (op + (incl - cnt));
                            if (cnt >= 1) o[0] = (uint8_t)word;
                            if (cnt >= 2) o[1] = (uint8_t)(word >> 8);
                            if (cnt >= 3) o[2] = (uint8_t)(word >> 16);
                            UNIFORM_JOIN();
                            op += lane_get(incl, 63);
                        }
                        if (__builtin_expect(has_long != 0, 0))
                            op = emit_copy(out, op, Lip - Lcnd, Lm, lane);
                        TRACE(8); MARK("T8");
                        ABLATION_PAD(flags, 0x2000u);
                        ABLATION_PAD_S(flags, 0x20000u);
                        if (!cont) {  // this wave goes on alone (general code below, or the end of the block)
                            poisoned = poisoned || tpois != 0;
                            P = P2;
                            chain = chain2 != 0;
                            sc = sc2;
                            break;
                        }
                        team_publish(team.otok, (uint64_t)(op | (tpois << 17)) | ((uint64_t)(myturn + 1) << 42));
                        own_op = false;
                        holder = false;
                        myturn += kTeamW;
                        // ---- this wave's next tile, if the parse goes on tile by tile: the next wave takes the tile of P2
                        thead = (P2 >> 6) + (kTeamW - 1);
                        need_head = ((ip_limit - ((thead << 6) + kFastMargin)) >> 31) ^ 1u;  // (thead << 6) + kFastMargin <= ip_limit
                        TRACE(9); MARK("T9");
                    }
                    if (team_gone)
                        break;
                    if (!chain)
                        scan_base = P - sc;
                    want_prep(P - (chain ? 1u : 0u), chain);
                    continue;
                }
            }
            if (kFast && kTeamW == 1 && chain && !hold_fast && P + kFastMargin <= ip_limit) {
                // =================================================== FAST DENSE PATH (fixed tiles)
                // Compressible data spends its time here.  The block is cut into fixed tiles of 64
                // positions (lane l of tile t = position 64 t + l), so the bytes of a tile come out of
                // the register buffers with constant shuffles, and a tile keeps its per-lane results
                // (hash, candidate, local match length) while the parse visits it once or several
                // times: a VISIT walks the lanes [e, lim) from the state (chain, sc) and commits their
                // table inserts.  Two inserted lanes that share a bucket end the visit before the second
                // one (lim = cut); the same tile is then visited again from there, and only the lanes
                // whose bucket was written in between get a new candidate.  The elements of a tile are
                // written once, when the parse leaves it (masks of all its visits accumulated).
                // States this path does not take (a scan that may reach probe 32, the last ~128
                // bytes of the block) go back to the general code with (P, chain, sc, next_emit) as they
                // stand.  tests/tile_model.py states the same algebra in plain integers (checked on the CPU).
                first_scan = false;
                // current tile: first dword of each position, hash, candidate (as validated), match
                // length for the walk (4..31, or 64: reaches 32 bytes and goes on), lanes whose candidate matches
                uint32_t t = 0, e = 0, sc_in = 0;
                uint32_t c_in = 1;  // (flags are wave-uniform integers: the compiler keeps bools as 64-bit lane masks)
                uint32_t a0 = 0, th = 0, tcand = 0, tmlw = 0;
                uint64_t Hraw = 0;
                // head of a tile whose candidate loads are in flight
                u32x4 na = {0, 0, 0, 0}, na2 = {0, 0, 0, 0}, nc = {0, 0, 0, 0}, nc2 = {0, 0, 0, 0};
                uint32_t nh = 0, ncand = 0, tn = 0;
                // visits of the current tile accumulated for the element writer
                uint64_t Vacc = 0, Hpacc = 0;
                bool acc_open = false;
                uint32_t acc_emit = next_emit;
                uint32_t carry = 0;  // lane j: byte acc_emit + j of an open literal that began in the tile before
                // where the parse stands when a tile is left / entered
                uint32_t P2 = P, sc2 = 0, long_lane = 64;
                uint32_t chain2 = 1, pm1_done = 0, quit = 0;
                auto head = [&](uint32_t tt) {  // bytes and hashes of tile tt
                    const uint32_t k = tt >> 1;  // buffer k = bytes [128 k, 128 k + 256) holds the tile + 32
                    if (k != kcur)
                        advance_bufs(k);
                    const uint32_t i0 = ((tt & 1u) << 6) + lane4;
                    uint32_t d[9];
#pragma unroll
                    for (int q = 0; q < 9; q++)
                        d[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i0 + 4 * q), (int)wcur);
                    na.x = __builtin_amdgcn_alignbyte(d[1], d[0], lane3);
                    na.y = __builtin_amdgcn_alignbyte(d[2], d[1], lane3);
                    na.z = __builtin_amdgcn_alignbyte(d[3], d[2], lane3);
                    na.w = __builtin_amdgcn_alignbyte(d[4], d[3], lane3);
                    na2.x = __builtin_amdgcn_alignbyte(d[5], d[4], lane3);
                    na2.y = __builtin_amdgcn_alignbyte(d[6], d[5], lane3);
                    na2.z = __builtin_amdgcn_alignbyte(d[7], d[6], lane3);
                    na2.w = __builtin_amdgcn_alignbyte(d[8], d[7], lane3);
                    nh = (na.x * kHashMul) >> shift;
                    tn = tt;
                };
                auto head_loads = [&]() {
                    // (plain loads: the same two as non-temporal loads for the global-table waves -- so that their candidates
                    // would not push the teams' blocks out of the L2 -- cost 11 % of the text rate, EXPERIMENTS.md R5.3)
                    nc = ld128(in + ncand);  // ncand < pos, pos + 32 <= n here
                    nc2 = ld128(in + ncand + 16);
                };
                // Where the parse goes from each lane, with what the exit of a visit needs to know about the last jump
                // beside it, so that the walk's last v_readlane IS the exit state (the team path's form, round 5 here too):
                //   X: bits 0..5 target lane (next hit at or after this lane, then past its match), bit 6 the jump leaves the
                //      tile (bits 0..5 then name the lane itself), bit 7 no hit from here on (the scan stays open), bit 8 that
                //      hit's match reaches 32 bytes (long), bits 10..15 lane of that hit, bits 16..23 where the parse stands
                //      afterwards relative to the tile;
                //   Y: the state a visit that ends with this lane's jump hands on (position, chain flag, "copy ends with the
                //      tile", scan index); Z: the pending literal's start.
                constexpr bool kFastX = true;
                uint32_t X = 0, Y = 0, Z = 0;
                uint64_t farmask = 0;
                auto jumps = [&]() {
                    const uint64_t tl = Hraw >> lane;
                    uint32_t dlo, dhi;
                    asm("v_ffbl_b32 %0, %1" : "=v"(dlo) : "v"((uint32_t)tl));
                    asm("v_ffbl_b32 %0, %1" : "=v"(dhi) : "v"((uint32_t)(tl >> 32)));
                    const uint32_t dist = umin3(dlo, dhi | 32u, 64u);  // 64: no hit from here on
                    const uint32_t nh_l = lane + dist;
                    const uint32_t m_nh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nh_l << 2), (int)tmlw);
                    const bool open = dist >= 64u;
                    const bool lng = !open && m_nh == 64u;
                    const uint32_t sum = nh_l + m_nh;
                    uint32_t tgt = sum > lane ? sum : lane + 1;  // (always so for a hit: ml >= 4; keeps the walk finite whatever memory holds)
                    tgt = (open || tgt > 64u) ? 64u : tgt;
                    const uint32_t p2off = open ? 64u : (lng ? nh_l : sum);
                    const uint32_t base_ = t << 6;
                    X = (tgt >= 64u ? lane | 64u : tgt) | (open ? 1u << 7 : 0u) | (lng ? 1u << 8 : 0u) | ((nh_l & 63u) << 10) |
                        ((p2off & 0xffu) << 16);
                    Y = (base_ + p2off) | (open ? 0u : 1u << 17) | ((!open && !lng && sum == 64u) ? 1u << 18 : 0u) |
                        (open ? (63u - lane) << 19 : 0u);
                    Z = open ? base_ + lane : base_ + p2off;
                    // lanes from which a scan could reach probe 32 inside the tile (32 hit-free lanes in a row)
                    farmask = ballot64(dist >= run32);
                };

                // the same for candidates that lie in the current tile (a second visit after a cut): both strings out of registers
                auto patch_regs = [&](uint64_t sel, uint32_t cnow) {
                    const bool on = __builtin_amdgcn_inverse_ballot_w64(sel);
                    const uint32_t r = on ? cnow - (kcur << 7) : 0u;  // (every lane takes part in the shuffles)
                    const uint32_t i0 = (r >> 2) << 2, sh = r & 3u;
                    uint32_t d[9];
#pragma unroll
                    for (int q = 0; q < 9; q++)
                        d[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i0 + 4 * q), (int)wcur);
                    u32x4 c, c2;
                    c.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
                    c.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
                    c.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
                    c.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
                    c2.x = __builtin_amdgcn_alignbyte(d[5], d[4], sh);
                    c2.y = __builtin_amdgcn_alignbyte(d[6], d[5], sh);
                    c2.z = __builtin_amdgcn_alignbyte(d[7], d[6], sh);
                    c2.w = __builtin_amdgcn_alignbyte(d[8], d[7], sh);
                    const uint32_t ml = matchlen32(na, na2, c, c2);
                    tcand = on ? cnow : tcand;
                    tmlw = on ? (ml == 32 ? 64u : ml) : tmlw;
                    Hraw = (Hraw & ~sel) | (ballot64(c.x == na.x) & sel);
                };

                PROF_LAP(14);
                auto enter_head = [&](uint32_t P2v, uint32_t e2v) {  // bytes, hashes, lookups, candidate loads of P2v's tile
                    head(P2v >> 6);
                    if (chain2 && !pm1_done) {  // the reference inserts P-1 before it looks P up
                        if (e2v >= 1) {
                            if (lane == e2v - 1)
                                tab.put(nh, P2v - 1);
                        } else if (lane == 0) {
                            tab.put((ld32(in + P2v - 1) * kHashMul) >> shift, P2v - 1);
                        }
                        asm volatile("" ::: "memory");
                    }
                    if constexpr (!Tab::kInLds && kSkipDeadLanes) {
                        // (round 6) A table in memory: the lanes below the one the parse enters the tile at -- the tail of
                        // the copy that crossed into it: 5.9 lanes of 64 on text -- are never probed, and that is known
                        // here: no look-up and no candidate string for them (9 % of a lone wave's requests over the
                        // fabric, which is what bounds it: EXPERIMENTS.md R6.7).  Their hashes are still needed (the
                        // insert of P-1, the commit); their hit bits are masked wherever the tile's are used (`keep`).
                        // (one predicate with the "bucket is occupied" test of the look-up; a dead lane's candidate is
                        // position 0 like an empty bucket's: its string is the block's first line, which every such lane shares)
                        ncand = tab.get(nh, lane >= e2v);
                        head_loads();
                    } else {
                        ncand = tab.get(nh);
                        head_loads();
                    }
                };
                for (;;) {  // ================================================ one tile per trip
                    // ---- (A) the elements closed inside the tile that is being left (all its visits)
                    MARK("flush");
                    ABLATION_PAD(flags, 0x1000u);
                    ABLATION_PAD_S(flags, 0x10000u);
                    if (__builtin_expect(Hpacc != 0, 1)) {
                        const uint32_t jfirst = ctz64(Hpacc);
                        const uint32_t jlast = 63u - (uint32_t)__builtin_clzll(Hpacc);
                        uint64_t Lit = Vacc & ~Hpacc & below63(jlast);
                        if (acc_open) {
                            // a literal was open when the accumulation began: it ends at the first hit.  It
                            // began at a failed chain probe of this path and ends within the 32 probes a scan
                            // makes here: one tag byte, at most 33 bytes
                            // (its bytes come out of registers -- those of the tile before were put aside when
                            // this tile was entered --: a load here would stall the wave on everything in flight)
                            const uint32_t len = (t << 6) + jfirst - acc_emit;
                            const uint32_t nprev = (t << 6) - acc_emit;
                            const uint32_t cur = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane - nprev) << 2), (int)a0);
                            const uint32_t lbyte = lane < nprev ? carry : cur;
                            if (lane == 0)
                                out[op] = (uint8_t)((len - 1) << 2);
                            if (lane < len)
                                out[op + 1 + lane] = (uint8_t)lbyte;
                            UNIFORM_JOIN();
                            poisoned = poisoned || len > 60;
                            op += 1 + len;
                            Lit &= ~below63(jfirst);
                        }
                        const uint64_t RS = Lit & ~(Lit << 1);
                        const uint64_t HC = long_lane < 64 ? Hpacc & ~(1ull << long_lane) : Hpacc;
                        const bool is_lit = __builtin_amdgcn_inverse_ballot_w64(Lit);
                        const bool run_start = __builtin_amdgcn_inverse_ballot_w64(RS);
                        const bool is_hit = __builtin_amdgcn_inverse_ballot_w64(HC);
                        const uint32_t runlen = (uint32_t)__builtin_ctzll((Hpacc >> lane) | (1ull << 63));
                        const uint32_t abyte = a0 & 0xffu;
                        const uint32_t off = (t << 6) + lane - tcand;
                        const bool short_form = tmlw < 12 && off < 2048;
                        uint32_t cnt = 0, word = 0;
                        if (is_lit) {
                            cnt = run_start ? 2 : 1;
                            word = run_start ? (((runlen - 1) << 2) | (abyte << 8)) : abyte;
                        } else if (is_hit) {
                            cnt = short_form ? 2 : 3;
                            word = short_form ? (1u | ((tmlw - 4) << 2) | ((off >> 8) << 5) | ((off & 0xffu) << 8))
                                              : (2u | ((tmlw - 1) << 2) | (off << 8));
                        }
                        const uint32_t incl = wave_inclusive_sum(cnt);
                        uint8_t *o = out + (op + (incl - cnt));
                        if (cnt >= 1) o[0] = (uint8_t)word;
                        if (cnt >= 2) o[1] = (uint8_t)(word >> 8);
                        if (cnt >= 3) o[2] = (uint8_t)(word >> 16);
                        UNIFORM_JOIN();
                        op += lane_get(incl, 63);
                    }
                    PROF_LAP(21);
                    if (__builtin_expect(long_lane < 64, 0)) {  // the last hit's match runs past its 32 local bytes: finish it
                        const uint32_t ip = (t << 6) + long_lane, cnd = lane_get(tcand, long_lane);
                        PROF_LAPV(26);
                        const uint32_t m = match_length<kMatchPieces>(in, cnd, ip, n, lane, 32);
                        PROF_LAPV(27);
                        op = emit_copy(out, op, ip - cnd, m, lane);
                        P2 = ip + m;
                        next_emit = P2;
                        chain2 = 1;
                        sc2 = 0;
                        pm1_done = 0;
                        long_lane = 64;
                        PROF_LAPV(18);
                    }
                    // ---- (B) enter the tile of P2 -- or leave: the block's tail, a state handed back (quit: P2,
                    // chain2, sc2 are that state), a poisoned block (one way out: every exit of its own costs copies)
                    MARK("enter");
                    if constexpr (AdoptT::kLone) {
                        // (a lone wave of the team kernel) The workgroup's team has run out of blocks and asks for this one:
                        // it is handed over here, where a tile ends -- its elements are written, (P2, chain2, sc2, next_emit,
                        // op) is the whole parse state, as for a state handed to the general code -- if enough of it is left
                        // to be worth a table's copy.  One LDS word read per tile; the claim is an LDS compare-and-swap (the
                        // team may withdraw its request: exactly one side gets it).
                        uint32_t *const req = tab.occ + kLoneReqWord, *const rec = tab.occ + kLoneRecWord;
                        if (__builtin_expect(uni(__hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == kAdoptAsked, 0) &&
                            ip_limit >= kAdoptMinBlock && P2 + kAdoptMinLeft <= ip_limit && quit == 0 && !poisoned) {
                            uint32_t was = kAdoptIdle;
                            if (lane == 0) {
                                uint32_t expect = kAdoptAsked;
                                (void)__hip_atomic_compare_exchange_strong(req, &expect, kAdoptTaken, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                                was = expect;
                            }
                            UNIFORM_JOIN();
                            if (uni(was) == kAdoptAsked) {
                                // everything this wave has stored -- elements, table entries -- is in the L2 before the team is
                                // told (workgroup-scope release: the team's waves run on this CU)
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                if (lane == 0) {
                                    rec[kAdoptRecState] = P2 | (chain2 << 17) | (sc2 << 18);
                                    rec[kAdoptRecEmit] = next_emit;
                                    rec[kAdoptRecOp] = op;
                                }
                                UNIFORM_JOIN();
                                // (no exit of its own -- every exit of this loop costs the allocator a set of copies, measured
                                // in rounds 3-5 --: the block leaves by the way a poisoned one does, with nothing left to emit; the
                                // kernel tells the two apart by the request word, which reads kAdoptTaken from here on)
                                next_emit = n;
                                poisoned = true;
                            }
                        }
                    }
                    if (__builtin_expect((quit != 0) | poisoned | (P2 + kFastMargin > ip_limit), 0)) {
                        P = P2;
                        chain = chain2 != 0;
                        sc = sc2;
                        break;
                    }
                    const uint32_t e2 = P2 & 63u;
                    enter_head(P2, e2);
                    PROF_LAP(20);
                    MARK("compare");
                    {  // the head becomes the current tile: compare -> hit mask, match lengths
                        const uint32_t ml = matchlen32(na, na2, nc, nc2);
                        Hraw = ballot64(nc.x == na.x);
                        tmlw = ml == 32 ? 64u : ml;
                        // an open literal comes along (then tn == t + 1): its bytes so far are a0 bytes of the tile being left
                        carry = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((next_emit - (t << 6) + lane) << 2), (int)a0) & 0xffu;
                        a0 = na.x;
                        th = nh;
                        tcand = ncand;
                        t = tn;
                    }
                    PROF_LAPV(22);
                    PROF_LAPV(24);
                    Vacc = 0;
                    Hpacc = 0;
                    acc_open = chain2 == 0;
                    acc_emit = next_emit;
                    e = e2;
                    c_in = chain2;
                    sc_in = sc2;
                    const uint32_t base = t << 6;
                    const uint32_t pos = base + lane;
                    PROF_LAP(25);
                    if (kFastX)
                        jumps();
                    for (;;) {  // ============================================ visits of this tile
                        // (every visit ends beyond its entry lane, every tile beyond the one before: no trip count)
                        MARK("visit_top");
                        const uint64_t keep = ~below63(e);  // the lanes the parse can still reach
                        const uint64_t Hm = Hraw & keep;
                        // ---- where the parse goes from each lane (X, Y, Z: made once per tile and after a patch), and the
                        // states left to the general code: a scan could reach probe 32 (32 hit-free lanes in a row, or the
                        // misses carried in plus those before the first hit)
                        uint32_t x = lane_get(X, e);
                        {
                            const uint32_t lead = (x & 0x80u) ? 63u - e : ((x >> 10) & 63u) - e;
                            const uint32_t lead_bad = (c_in ^ 1u) & ((31u - (sc_in + lead)) >> 31);  // sc_in + lead > 31
                            const uint64_t far = farmask & keep;
                            if (__builtin_expect(((uint32_t)far | (uint32_t)(far >> 32) | lead_bad) != 0, 0)) {  // (P2, chain2, sc2 are this very state: base + e, c_in, sc_in)
                                hold_fast = true;
                                quit = 1;
                                break;
                            }
                        }
                        PROF_LAP(15);
                        MARK("walk_loop");
                        // ---- the serial part: one v_readlane per element; the last one is the exit state
                        // (four steps at a time, no branch in between: s_lshl_b64 and v_readlane use bits 0..5 of their
                        // lane operand; a lane whose jump leaves the tile points at itself with bit 6 set, so steps past
                        // the end change nothing)
                        uint64_t S = 1ull << e;
                        do {
                            uint64_t bit;
                            asm volatile(
                                "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1\n\t"
                                "s_lshl_b64 %2, 1, %1\n\ts_or_b64 %0, %0, %2\n\ts_nop 1\n\tv_readlane_b32 %1, %3, %1"
                                : "+s"(S), "+s"(x), "=&s"(bit)
                                : "v"(X)
                                : "scc");  // (s_lshl_b64 and s_or_b64 write SCC)
                        } while (!(x & 64u));
                        PROF_LAP(16);
                        MARK("exit_commit");
                        // ---- exit state, commit, check.  Common case first (no two inserted lanes share a bucket); a
                        // shared bucket repeats the exit computation for a visit that ends before lane `cut`.
                        const uint32_t s_l = x & 63u;  // the last start: the lane whose jump left the tile
                        uint64_t V, Hp, C;
                        uint32_t ll, emit2;
                        {
                            const uint32_t y = lane_get(Y, s_l), z = lane_get(Z, s_l);
                            const uint32_t open = (x >> 7) & 1u;
                            // the scan carried in found nothing: (s_l == e) as an integer, ((s_l ^ e) - 1) >> 31
                            const uint32_t carried = (((s_l ^ e) - 1u) >> 31) & (c_in ^ 1u) & open;
                            P2 = y & 0x1ffffu;
                            chain2 = (y >> 17) & 1u;
                            pm1_done = (y >> 18) & 1u;
                            sc2 = carried ? sc_in + 64u - e : (y >> 19) & 63u;
                            emit2 = carried ? next_emit : z;
                            ll = (x & 0x100u) ? (x >> 10) & 63u : 64u;
                            Hp = Hm & ~(Hm - S);  // first hit of every segment [start, next start)
                            V = (Hp << 1) - S;    // [start, hit] of every segment; an open last one runs to lane 63
                            C = V | ((S & ~(1ull << e)) >> 1) | ((uint64_t)pm1_done << 63);  // + the lane before each chain probe
                        }
                        {
                            const bool inC = __builtin_amdgcn_inverse_ballot_w64(C);
                            uint64_t L;
                            if constexpr (Tab::kInLds) {
                                uint32_t w = pos & 0xffffu;
                                if (inC) {
                                    tab.put(th, pos);
                                    asm volatile("" ::: "memory");  // read back from LDS, not from `pos`
                                    w = tab.get(th);
                                }
                                L = ballot64(inC && w != (pos & 0xffffu));
                            } else {
                                const uint32_t bit = 1u << (th & 31u);
                                uint32_t old = 0;
                                if (inC)
                                    old = __hip_atomic_fetch_or(tab.bm_word(th), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                if (inC)
                                    __hip_atomic_store(tab.bm_word(th), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                L = ballot64(inC && (old & bit) != 0);
                            }
                            if (__builtin_expect(L != 0, 0)) {  // two inserted lanes share a bucket: end the visit before the second one
                                if constexpr (Tab::kInLds) {
                                    asm volatile("" ::: "memory");
                                    if (inC)
                                        tab.put(th, tcand);  // roll back (lanes of one bucket hold the same old entry)
                                }
                                uint32_t cut = 64;
                                while (L) {
                                    const uint32_t hv = lane_get(th, ctz64(L));
                                    const uint64_t G = ballot64(inC && th == hv);
                                    L &= ~G;
                                    const uint64_t G2 = G & (G - 1);
                                    const uint32_t second = G2 ? ctz64(G2) : 64u;  // (global home: a lane alone in G shared its BIT with another bucket)
                                    cut = second < cut ? second : cut;
                                }
                                // (cut <= e, or with the table in LDS no cut at all, cannot happen: a lost write that no
                                // shared bucket explains.  The block is poisoned instead of looping on it.  Global home: no
                                // cut at all = only bits of the scratch bitmap were shared, no bucket: the visit stands.)
                                const bool no_cut = !Tab::kInLds && cut >= 64;
                                const bool bad_cut = !no_cut && (cut <= e || cut >= 64);
                                poisoned = poisoned || bad_cut;
                                if (!no_cut) {
                                    // the same exit for a visit that ends before lane `cut`: the last start below it, and either
                                    // its hit lies below the cut too (then nothing changes for it) or the scan is open there
                                    cut = bad_cut ? (e < 62u ? e + 1u : 63u) : cut;  // (<= 63: `1ull << cut` and below63(cut) below; the block is poisoned anyway)
                                    const uint64_t bl = below63(cut);
                                    const uint64_t Sx = S & bl, Hm_t = Hm & bl;
                                    const uint32_t sl = 63u - (uint32_t)__builtin_clzll(Sx | 1ull);
                                    const uint32_t xs = lane_get(X, sl);
                                    const uint32_t open = ((xs >> 7) & 1u) | ((((xs >> 10) & 63u) - cut) >> 31 ^ 1u);  // no hit, or it lies at or past the cut
                                    const uint32_t carried = (((sl ^ e) - 1u) >> 31) & (c_in ^ 1u) & open;
                                    const uint32_t y = lane_get(Y, sl), z = lane_get(Z, sl);
                                    P2 = open ? base + cut : y & 0x1ffffu;
                                    chain2 = open ^ 1u;
                                    pm1_done = 0;
                                    sc2 = open ? (carried ? sc_in + cut - e : cut - sl - 1u) : 0u;
                                    emit2 = open ? (carried ? next_emit : base + sl) : z;
                                    ll = ((open ^ 1u) & (xs >> 8) & 1u) ? (xs >> 10) & 63u : 64u;
                                    Hp = Hm_t & ~(Hm_t - Sx);
                                    V = (((Hp << 1) | (open ? 1ull << cut : 0ull)) - Sx) & bl;
                                    C = V | ((Sx & ~(1ull << e)) >> 1);
                                }
                                if (__builtin_amdgcn_inverse_ballot_w64(C))
                                    tab.put(th, pos);  // lanes below a cut share no bucket
                                UNIFORM_JOIN();
                            } else if constexpr (!Tab::kInLds) {
                                if (inC)
                                    tab.put(th, pos);
                                UNIFORM_JOIN();
                            }
                        }
                        PROF_LAP(17);
                        MARK("after_commit");
                        next_emit = emit2;
                        Vacc |= V;
                        Hpacc |= Hp;
                        long_lane = ll;  // (64: none)
                        // (one way out of the visits of a tile: every exit of its own costs a set of copies)
                        if (__builtin_expect(poisoned | (ll < 64) | ((P2 >> 6) != t), 1))
                            break;
                        // the same tile again (after a cut): insert P-1, then give the lanes whose bucket
                        // was written since their lookup the new candidate
                        const uint32_t e2s = P2 & 63u;
                        if (chain2 && lane == e2s - 1)
                            tab.put(th, pos);
                        asm volatile("" ::: "memory");
                        const uint32_t cn = tab.get(th);
                        const uint64_t st = ballot64(cn != tcand) & ~below63(e2s);
                        if (st) {
                            // (what has been written since the tile's look-ups was written by this tile's own inserts: the new
                            // candidates lie in the tile, their strings in the register buffer -- no round trip to memory)
                            patch_regs(st, cn);
                            jumps();
                        }
                        PROF_LAPV(23);
                        e = e2s;
                        c_in = chain2;
                        sc_in = sc2;
                        MARK("visit_end");
                    }
                }
                if (!chain)
                    scan_base = P - sc;
                want_prep(P - (chain ? 1u : 0u), chain);
                continue;
            }
            if (chain || sc <= 31) {
                // =================================================== dense window
                if (P >= ip_limit)
                    break;  // remainder
                hold_fast = false;
                if (!prepped) {
                    prep(prepW, prepC);
                    prepped = true;
                }
                const uint32_t first = chain ? 1u : 0u;
                const uint32_t W0 = P - first;
                const uint32_t pos = W0 + lane;
                const bool valid = pos < ip_limit;  // a prefix of the lanes; lane `first` is valid
                const bool wide = pos + 32 <= n;    // a2 really holds bytes 16..31 (a prefix too)
                const uint32_t h = hN;  // hash, insert of P-1 and lookup: done ahead by prep(W0, chain, .)
                PROF_LAPV(0);
                const bool spec = valid && lane >= first;
                const uint32_t cand = candN;
                // lanes that may be probed: 63 at most (lane 63 only lends its bytes; see the walk)
                const uint32_t nvalid = ip_limit - W0 < 63 ? ip_limit - W0 : 63;  // W0 < ip_limit
                PROF_LAPV(1);
                // candidates' bytes: unpredicated (cand < pos <= n-16, or 0 for lanes without one)
                u32x4 c = {~a.x, 0, 0, 0}, c2 = {0, 0, 0, 0};  // (no load: no lane hits)
                if (!zok || __ballot(cand != 0 || (spec && a.x == uni(z0v))) != 0) {
                    c = ld128(in + cand);
                    c2 = ld128(in + (cand + 16 < last16 ? cand + 16 : last16));
                }
                const bool hit = spec && c.x == a.x;
                // local match length: 4..32 (4..16 near the block end); `mcap` reached = may go on
                const uint32_t mcap = wide ? 32u : 16u;
                uint32_t ml = matchlen32(a, a2, c, c2);
                ml = ml < mcap ? ml : mcap;
                const uint32_t mlw = ml == mcap ? 64u : ml;  // for the walk: a long match ends the window
                const uint64_t Hm_all = __ballot(hit);
                const uint64_t E16 = __ballot(hit && ml == mcap);  // lanes whose match must be extended
                PROF_LAPV(2);

                // ---- scalar walk over the hit mask: which lanes does the serial parse visit?
                // The lookups above are exact as long as no two INSERTED lanes of the window share a
                // bucket (the later one should have seen the earlier one).  That is checked after the
                // fact: commit the inserts, read them back, and in the rare case of a lost write roll
                // back and redo the walk on the lanes before the second user of the bucket.
                uint32_t lim = nvalid;  // lanes [first, lim) can be probed
                uint64_t V, Hp, Ins;    // probed lanes / hits on the path / inserted-only lanes
                uint32_t j, s, long_lane, lit_lane, new_scan_base;
                bool ch, to_strided;
                for (;;) {
                    const uint64_t Hm = Hm_all & below63(lim);  // lim <= 63
                    V = 0;
                    Hp = 0;
                    Ins = 0;
                    j = first;
                    ch = chain;
                    s = sc;
                    long_lane = 64;  // hit whose match runs past the local bytes: finished below
                    to_strided = false;
                    lit_lane = 64;   // lane where the literal that is open at exit began
                    new_scan_base = scan_base;
                    // Can a scan of this window reach probe 32 (where the step becomes 2)?  Only through
                    // 32 misses in a row: a run of 32 hit-free lanes, or the misses carried in plus those
                    // before the first hit.  Almost never in compressible data -> lean walk.
                    const uint64_t range = below63(lim) & ~(uint64_t)first;
                    uint64_t z = ~Hm & range;
                    z &= z >> 1;
                    z &= z >> 2;
                    z &= z >> 4;
                    z &= z >> 8;
                    z &= z >> 16;
                    const uint32_t lead = (Hm ? ctz64(Hm) : lim) - first;  // misses before the first hit
                    if (z == 0 && (chain || sc + lead <= 31)) {
                        // Every lane works out where the parse goes from it: to the next hit at or after
                        // it and then past that hit's match (63 = out of the window; lane 63 is never
                        // probed, so it can stand for "out").  The serial part is then one v_readlane per
                        // element: S collects the lanes at which a literal run or a hit begins.
                        const uint64_t tl = Hm >> lane;
                        const uint32_t nh = (lane + (uint32_t)__builtin_ctzll(tl | (1ull << 63))) & 63u;
                        const uint32_t m_nh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nh << 2), (int)mlw);
                        const uint32_t tgt = nh + m_nh;
                        const uint32_t vJ = (tl != 0 && tgt < lim) ? tgt : 63u;
                        uint64_t S = 0;
                        uint32_t jj = first;
                        do {
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                S |= 1ull << jj;
                                jj = lane_get(vJ, jj);
                            }
                        } while (jj != 63u);
                        S &= ~(1ull << 63);
                        const uint32_t s_last = 63u - (uint32_t)__builtin_clzll(S);  // (lane `first` is in S)
                        // path hits = the first hit of every segment [start, next start): the borrow of
                        // Hm - S runs from a start up to the first hit above it and clears it
                        Hp = Hm & ~(Hm - S);
                        if ((Hm >> s_last) == 0) {  // no hit after the last start: trailing misses up to lim
                            V = ((Hp << 1) | (1ull << lim)) - S;  // [start, hit] of each element + [s_last, lim)
                            if (Hp || chain) {
                                lit_lane = s_last;
                                new_scan_base = W0 + s_last + 1;
                                s = lim - s_last - 1;
                            } else {
                                s = sc + lim - s_last;
                            }
                            ch = false;
                            j = lim;
                        } else {
                            V = (Hp << 1) - S;
                            const uint32_t jh = 63u - (uint32_t)__builtin_clzll(Hp);
                            j = jh + lane_get(mlw, jh);
                            ch = true;
                            s = 0;
                            if ((E16 >> jh) & 1)
                                long_lane = jh;
                        }
                        // the lane before each chain probe is inserted too: it is the last lane of a
                        // copy, i.e. not visited itself while its successor is
                        Ins = (V >> 1) & ~V & ~1ull;  // (lane 0 = P-1 in chain mode went in up front)
                    } else {
                        while (j < lim) {
                            const uint64_t Hrem = Hm >> j;
                            const uint32_t k = Hrem ? ctz64(Hrem) : 64u;  // misses before the next hit
                            const uint32_t budget = ch ? 32u : 31u - s;   // lanes left before probe 32 (step 2)
                            const uint32_t avail = lim - j;
                            if (k <= budget && k < avail) {
                                const uint32_t jh = j + k;
                                V |= lanes_below(k + 1) << j;
                                Hp |= 1ull << jh;
                                lit_lane = 64;
                                ch = true;
                                s = 0;
                                if ((E16 >> jh) & 1) {
                                    long_lane = jh;
                                    break;
                                }
                                const uint32_t jn = jh + lane_get(ml, jh);
                                if (jn < lim)
                                    Ins |= 1ull << (jn - 1);
                                j = jn;
                            } else if (k > budget && budget < avail) {
                                // the scan reaches probe 32 (step 2) inside the window: leave dense mode
                                V |= lanes_below(budget + 1) << j;
                                if (ch) {
                                    lit_lane = j;
                                    new_scan_base = W0 + j + 1;
                                }
                                ch = false;
                                s = 32;
                                to_strided = true;
                                break;
                            } else {  // misses up to the end of the usable lanes
                                V |= lanes_below(avail) << j;
                                if (ch) {
                                    lit_lane = j;
                                    new_scan_base = W0 + j + 1;
                                    s = avail - 1;
                                } else {
                                    s += avail;
                                }
                                ch = false;
                                j = lim;
                            }
                        }
                    }
                    PROF_LAPV(8);
                    // ---- commit the inserts of the visited path, then check that none was lost
                    const bool ins_lane = ((V | Ins) >> lane) & 1;
                    uint64_t L;
                    if constexpr (Tab::kInLds) {
                        uint32_t w = pos & 0xffffu;
                        if (ins_lane) {
                            tab.put(h, pos);
                            asm volatile("" ::: "memory");  // read back from LDS, not from `pos`
                            w = tab.get(h);
                        }
                        UNIFORM_JOIN();
                        L = __ballot(ins_lane && w != (pos & 0xffffu));
                        PROF_LAPV(9);
                        if (L == 0)
                            break;
                        // two inserted lanes share a bucket: roll back (both read the same old entry) and
                        // keep the lanes before the second user; the redo cannot conflict again
                        asm volatile("" ::: "memory");
                        if (ins_lane)
                            tab.put(h, cand);
                        UNIFORM_JOIN();
                    } else {
                        // Table in global memory: a read-back would be 64 more L2 misses on the critical
                        // path.  Shared buckets are found in a 16384-bit LDS bitmap instead (atomic OR:
                        // of the lanes with one bucket exactly one sees its bit clear); the table is
                        // written only once the path is final, so there is nothing to roll back.
                        const uint32_t bit = 1u << (h & 31u);
                        uint32_t old = 0;
                        if (ins_lane)
                            old = __hip_atomic_fetch_or(tab.bm_word(h), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        L = __ballot(ins_lane && (old & bit));
                        if (ins_lane)
                            __hip_atomic_store(tab.bm_word(h), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        PROF_LAPV(9);
                        if (L == 0) {
                            if (ins_lane)
                                tab.put(h, pos);
                            UNIFORM_JOIN();
                            break;
                        }
                    }
                    uint32_t cut = 64;
                    while (L) {
                        const uint32_t hv = lane_get(h, ctz64(L));
                        const uint64_t G = __ballot(ins_lane && h == hv);
                        L &= ~G;
                        const uint64_t G2 = G & (G - 1);
                        const uint32_t second = G2 ? ctz64(G2) : 64u;  // (global home: a lane alone in G shared its BIT with another bucket)
                        cut = second < cut ? second : cut;
                    }
                    if constexpr (!Tab::kInLds) {
                        if (cut >= 64) {  // only bits of the scratch bitmap were shared, no bucket: the path is final
                            if (ins_lane)
                                tab.put(h, pos);
                            UNIFORM_JOIN();
                            break;
                        }
                    }
                    // (a lost write that no shared bucket explains would repeat forever: with an LDS
                    // table it cannot happen; with a global-memory table it would mean this wave's
                    // store -> load order was not kept.  Poison the block's result, never hang.)
                    if (cut >= lim) {
                        poisoned = true;
                        break;
                    }
                    lim = cut;
                }
                scan_base = new_scan_base;
                PROF_LAP(3);

                PROF_ADD(10, lim);
                PROF_ADD(11, lim < nvalid ? 1 : 0);
                PROF_ADD(12, __builtin_popcountll(Hp));
                PROF_ADD(13, (long_lane < 64 ? W0 + long_lane : W0 + j) - P);
                // ---- next state, decided before the stores so that the next window's bytes are already
                // on their way while this window's elements are written
                const uint32_t abyte = a.x & 0xffu;  // a is about to be reloaded
                const bool entry_chain = chain;
                const uint32_t entry_emit = next_emit;
                if (long_lane == 64) {
                    P = to_strided ? scan_base + 32 : W0 + j;  // (strided recomputes its positions)
                    chain = ch;
                    sc = s;
                    if (ch)
                        next_emit = P;
                    else if (lit_lane < 64)
                        next_emit = W0 + lit_lane;
                    if (chain || sc <= 31)
                        want_prep(P - (chain ? 1u : 0u), chain);
                }
                first_scan = first_scan && !Hp && (scan_base == 1);

                // ---- emit the elements closed inside this window
                if (Hp) {
                    const uint32_t jfirst = ctz64(Hp);
                    const uint32_t jlast = 63u - (uint32_t)__builtin_clzll(Hp);
                    if (!entry_chain)  // a literal was open on entry (it began before W0): it ends at the first hit
                        emit_lit(entry_emit, W0 + jfirst - entry_emit);
                    uint64_t Lit = V & ~Hp & below63(jlast);
                    if (!entry_chain)
                        Lit &= ~below63(jfirst);
                    const bool is_lit = (Lit >> lane) & 1;
                    const bool is_hit = (Hp >> lane) & 1;
                    const bool run_start = is_lit && !((lane > 0) && ((Lit >> (lane - 1)) & 1));
                    // literal length of the run starting here = distance to the next hit on the path
                    const uint64_t hp_after = Hp >> lane;
                    const uint32_t runlen = (uint32_t)__builtin_ctzll(hp_after | (1ull << 63));
                    const uint32_t off = pos - cand;
                    const bool short_form = ml < 12 && off < 2048;
                    uint32_t cnt = 0, word = 0;
                    if (is_lit) {
                        cnt = run_start ? 2 : 1;
                        word = run_start ? (((runlen - 1) << 2) | (abyte << 8)) : abyte;
                    } else if (is_hit && lane != long_lane) {
                        cnt = short_form ? 2 : 3;
                        word = short_form ? (1u | ((ml - 4) << 2) | ((off >> 8) << 5) | ((off & 0xffu) << 8))
                                          : (2u | ((ml - 1) << 2) | (off << 8));
                    }
                    const uint32_t incl = wave_inclusive_sum(cnt);
                    uint8_t *o = out + op + (incl - cnt);
                    if (ABLATION_KEEPS(!(flags & 0x400u)))  // (-DSZ_ABLATION timing builds: no element stores)
                    {
                    if (cnt >= 1) o[0] = (uint8_t)word;
                    if (cnt >= 2) o[1] = (uint8_t)(word >> 8);
                    if (cnt >= 3) o[2] = (uint8_t)(word >> 16);
                    }
                    UNIFORM_JOIN();
                    op += lane_get(incl, 63);
                }
                PROF_LAP(4);

                if (long_lane < 64) {  // the last hit's match runs past the local window: finish it
                    const uint32_t ip = W0 + long_lane, cnd = lane_get(cand, long_lane);
                    const uint32_t m = match_length<kMatchPieces>(in, cnd, ip, n, lane, lane_get(mcap, long_lane));
                    op = emit_copy(out, op, ip - cnd, m, lane);
                    P = ip + m;
                    chain = true;
                    sc = 0;
                    next_emit = P;
                    want_prep(P - 1, true);
                }
                PROF_LAP(5);
                continue;
            }

            // ======================================================= strided batch (sc >= 32)
            const uint32_t n0 = sc;
            uint32_t p, pn;
            const bool aligned = !kLean && (((n0 & 63u) == 0 && n0 < 384) || n0 == 32);
            const uint32_t bsel = n0 >> 6;  // 0 only for the half batch n0 == 32
            uint32_t cap = 64;  // lanes allowed in this batch
            if (aligned) {
                p = scan_base + sel6(Pr0b, Pr1, Pr2, Pr3, Pr4, Pr5, bsel);
                pn = scan_base + sel6(Pn0b, Pn1, Pn2, Pn3, Pn4, Pn5, bsel);
                cap = n0 == 32 ? 32 : 64;
            } else {
                uint32_t idx = n0 + lane;
                idx = idx < kProbeEntries - 2 ? idx : kProbeEntries - 2;
                p = scan_base + g_probe.off[idx];
                pn = scan_base + g_probe.off[idx + 1];
                cap = 64 - (n0 & 63u);  // realign to the register-resident schedule
            }
            const bool valid = pn <= ip_limit && lane < cap;  // the reference checks next_ip before probing ip
            const bool limited = __ballot(pn > ip_limit && lane < cap) != 0;
            u32x4 a = {0, 0, 0, 0};
            bool have_win = true;  // a.y..a.w hold bytes p+4..p+15
            if (first_scan && aligned && bsel <= 4) {
                a.x = (bsel == 0) ? pv0b : (bsel == 1) ? pv1 : (bsel == 2) ? pv2 : (bsel == 3) ? pv3 : pv4;
                have_win = false;
            } else if (valid) {
                a = ld128(in + p);
            }
            const uint32_t v = a.x;
            uint32_t h = 0, cand = 0, w = p & 0xffffu;
            const uint32_t nvalid = (uint32_t)__builtin_popcountll(__ballot(valid));  // valid is a prefix
            uint32_t cut = 64;
            if constexpr (Tab::kInLds) {
                if (valid) {
                    h = (v * kHashMul) >> shift;
                    cand = tab.get(h);
                    tab.put(h, p);
                    asm volatile("" ::: "memory");  // the read-back below must come from LDS, not from p
                    w = tab.get(h);
                }
                const uint64_t losers = __ballot(valid && w != (p & 0xffffu));
                if (losers) {
                    // Two lanes of this batch share a bucket: serial order says the later one must see the
                    // earlier one's position.  Roll everything back, keep the longest collision-free prefix.
                    if (valid)
                        tab.put(h, cand);
                    uint64_t L = losers;
                    while (L) {
                        const uint32_t hv = lane_get(h, ctz64(L));
                        const uint64_t G = __ballot(valid && h == hv);
                        L &= ~G;
                        const uint64_t G2 = G & (G - 1);
                        const uint32_t second = ctz64(G2);
                        cut = second < cut ? second : cut;
                    }
                    if (valid && lane < cut)
                        tab.put(h, p);
                    UNIFORM_JOIN();
                }
            } else {
                // Table in global memory: a read-back would be one more round trip to the L2 in every batch (round 5:
                // a batch had four dependent ones -- probe bytes, look-up, read-back, candidate strings).  Lanes of
                // one bucket are found in the LDS scratch bitmap, as in the dense paths, BEFORE anything is written:
                // nothing to roll back, the lanes below the second user of a bucket are inserted.
                uint32_t old = 0;
                const uint32_t bit = 1u << (((v * kHashMul) >> shift) & 31u);
                if (valid) {
                    h = (v * kHashMul) >> shift;
                    cand = tab.get(h);
                    old = __hip_atomic_fetch_or(tab.bm_word(h), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                uint64_t L = __ballot(valid && (old & bit) != 0);
                if (valid)
                    __hip_atomic_store(tab.bm_word(h), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                while (L) {
                    const uint32_t hv = lane_get(h, ctz64(L));
                    const uint64_t G = __ballot(valid && h == hv);
                    L &= ~G;
                    const uint64_t G2 = G & (G - 1);
                    const uint32_t second = G2 ? ctz64(G2) : 64u;  // (a lane alone in G shared its BIT with another bucket)
                    cut = second < cut ? second : cut;
                }
                if (valid && lane < cut)
                    tab.put(h, p);
                UNIFORM_JOIN();
            }
            const uint32_t nact = nvalid < cut ? nvalid : cut;
            const bool act = lane < nact;
            u32x4 c = {~v, 0, 0, 0};  // (no load: no lane hits)
            if (!zok || __ballot(act && (cand != 0 || v == uni(z0v))) != 0) {
                c = u32x4{0, 0, 0, 0};
                if (act)
                    c = ld128(in + cand);  // cand < p <= n-16: in bounds
            }
            const uint64_t hits = __ballot(act && c.x == v);
            PROF_LAP(6);
            if (hits == 0) {
                if (nact == nvalid && limited)
                    break;  // ran into ip_limit: remainder
                sc = n0 + nact;
                continue;
            }
            const uint32_t f = ctz64(hits);
            if (act && lane > f)
                tab.put(h, cand);  // probes behind the first hit never happened
            const uint32_t ip = lane_get(p, f);
            const uint32_t cnd = lane_get(cand, f);
            uint32_t m;
            if (have_win) {
                m = lane_get(matchlen16(a.y, a.z, a.w, c.y, c.z, c.w), f);
                if (m == 16)
                    m = match_length<kMatchPieces>(in, cnd, ip, n, lane, 16);
            } else {
                m = match_length<kMatchPieces>(in, cnd, ip, n, lane, 4);
            }
            // (a7) pending literal, (a6) the copy; then the copy chain continues in a dense window
            emit_lit(next_emit, ip - next_emit);
            op = emit_copy(out, op, ip - cnd, m, lane);
            P = ip + m;
            chain = true;
            sc = 0;
            next_emit = P;
            first_scan = false;
            want_prep(P - 1, true);
            PROF_LAP(7);
        }
        PROF_FLUSH();
        PE_LAP(30);
        if constexpr (kTeamW > 1) {
            if (team_gone)  // another wave finishes (or finished) the block; a lost wave says so
                return team_lost ? kTeamLost : kTeamNotMine;
        }
        if (next_emit < n) {  // remainder literal
            emit_lit(next_emit, n - next_emit);
            next_emit = n;
        }
        if (big_len) {  // the block began with a long literal: tag + bytes from the prefetched registers
            const uint32_t word = (61u << 2) | ((big_len - 1) << 8);  // len >= 257: two length bytes
            if (lane < 3)
                out[blk_op + lane] = (uint8_t)(word >> (8 * lane));
            uint8_t *d = out + blk_op + 3;
            if (lane < bhead)
                d[lane] = in[lane];
            uint32_t nreg = (big_len - bhead) >> 10;  // whole chunks inside the literal
            nreg = nreg < nchunk ? nreg : nchunk;
            uint8_t *dl = d + bhead + 16u * lane;
#pragma unroll
            for (int u = 0; u < kBlkChunks; u++) {
                if ((uint32_t)u < nreg)
                    __builtin_nontemporal_store(blk[u], reinterpret_cast<u32x4 *>(dl + 1024 * u));
            }
            const uint32_t done = bhead + (nreg << 10);  // dst + done is 16-byte aligned
            if (done < big_len)
                wave_copy(d + done, in + done, big_len - done, lane);
        }
    }
    if (next_emit < n)  // blocks shorter than the 15-byte margin are one literal
        op = emit_literal<(kLean ? 0 : 16)>(out, op, in + next_emit, n - next_emit, lane);
    if constexpr (kTeamW > 1) {
        if (n >= 1024)  // the waves that wait for a token leave
            team_publish(team.ptok, (uint64_t)kTeamDoneTurn << 42);
    }
    PE_LAP(31);
    return poisoned ? 0xfffffffeu : op;
}

// A pointer made from an integer (the team kernel reads its arguments as integers from the kernel-argument segment)
// has no address space for the compiler: every access through it becomes a FLAT instruction, which counts in BOTH
// memory counters -- each wait for an LDS result then also waits for every load in flight.  Said here: global memory.
template <typename T>
__device__ __forceinline__ T *as_global(uint64_t addr)
{
    typedef __attribute__((address_space(1))) T GT;
    return (T *)(GT *)(uintptr_t)addr;
}

// block length: explicit per block, or the fragments of one contiguous input (ragged tail), or fixed
__device__ __forceinline__ uint32_t block_length(const uint32_t *__restrict__ in_len, uint64_t total_len,
                                                 uint64_t in_stride, uint32_t b)
{
    uint32_t n_raw;
    if (in_len)
        n_raw = in_len[b];
    else if (total_len) {
        const uint64_t left = total_len - (uint64_t)b * in_stride;
        n_raw = (uint32_t)(left < in_stride ? left : in_stride);
    } else
        n_raw = (uint32_t)in_stride;
    return uni(n_raw);
}

// ---- choosing between the kernels: per launch, by a sample of the data (auto, the default)
// The persistent kernels win on compressible data (several waves per SIMD fill each other's issue slots), the single
// kernel on incompressible data (whole block in registers, no slow global-table wave: +7 %).  A tiny sampling kernel
// decides for the whole launch (below); every candidate kernel is enqueued and the ones not chosen exit at once: no
// host synchronisation, graph-capturable, and a wrong guess only costs speed.  (A choice per block inside the launch --
// the single kernel handing blocks with a copy to the multi-wave kernel, SZ_KERNEL=route in rounds 2-3 -- measured
// slower on every workload but a compressible minority in an incompressible launch and was removed in round 4:
// tools/patches/route_mode.patch, EXPERIMENTS.md section 4.1.2.)
__global__ __launch_bounds__(64) void reset_launch_state_kernel(uint32_t *__restrict__ select)
{
    // All 16 words, the arrival counter of the sample's workgroups (select[15]) too: it returns to zero by itself after
    // every sampled launch that runs to its end, but one that did not -- an aborted launch -- would make every later
    // sample on this workspace elect its last workgroup early; a launch with a kernel named by the caller is the one
    // place where no sample is in flight on the stream.
    if (threadIdx.x < 16)
        select[threadIdx.x] = 0;
}

// Sampled choice (default auto mode): repeats among the 4-byte shingles of the first 1 KiB of up to 64
// evenly spaced blocks; both kernels are launched and the one not chosen exits at once.
__device__ __forceinline__ bool prefers_dual(const uint32_t *__restrict__ select)
{
    return (uint64_t)select[0] * 10 > select[1];  // repeats > 10 % of the sampled positions
}

// Cost class of one block, by one wave (see "longest blocks first" below): 0 cheap (no repeats among the shingles of
// its first 1 KiB: the scan accelerates), 1 medium (repeats, but few distinct strings among 128 samples spread over
// the block: it repeats itself at long range, long matches skip most of it; or a short block), 2 expensive.
// `bitmap`: 384 zeroed words of LDS of this wave's own.
__device__ __forceinline__ uint32_t block_cost_class(const uint8_t *__restrict__ in, uint32_t n, uint32_t lane, uint32_t *bitmap)
{
    const uint32_t n1 = n < 1024 ? n : 1024;
    const uint32_t npos = n1 >= 4 ? n1 - 3 : 0;
    // (every load of the block is requested before the first result is used: one round trip to memory, not two or
    // eighteen -- in a routed launch this runs in front of every block of the single kernel, one wave per SIMD)
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t p = lane + 64u * k;
        v[k] = p < npos ? ld32(in + p) : 0u;
    }
    uint64_t far[2] = {0, 0};
    const uint32_t nsamp = n >= 4096 ? 128u : 0u;
    // (a power-of-two stride for a full block: data that repeats with a period that divides a multiple of 512
    // -- pages, records, a pattern tiled over the block -- shows few distinct samples)
    const uint32_t stride = n >> 7;  // 128 samples over the block; the last one ends at 127 n / 128 + 8 <= n
    if (nsamp) {
        far[0] = ld64(in + lane * stride);
        far[1] = ld64(in + (lane + 64u) * stride);
    }
    uint32_t dups = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t p = lane + 64u * k;
        if (p < npos) {
            const uint32_t h = (v[k] * kHashMul) >> 19;  // 13 bits
            const uint32_t bit = 1u << (h & 31u);
            dups += (atomicOr(&bitmap[h >> 5], bit) & bit) ? 1u : 0u;
        }
    }
    uint32_t fresh = 0;
    if (nsamp) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t h = ((uint32_t)((far[k] * 0x9e3779b97f4a7c15ull) >> 40)) & 4095u;
            const uint32_t bit = 1u << (h & 31u);
            fresh += (atomicOr(&bitmap[256 + (h >> 5)], bit) & bit) ? 0u : 1u;
        }
    }
    dups = lane_get(wave_inclusive_sum(dups), 63);
    fresh = lane_get(wave_inclusive_sum(fresh), 63);
    uint32_t c = 0;
    if ((uint64_t)dups * 10 > npos)
        c = (nsamp && fresh * 2 < nsamp) ? 1u : (n >= 4096 ? 2u : 1u);
    return c;
}
// The launch sample saw blocks of more than one cost class (select[7] = classes seen): ordering the queue pays
__device__ __forceinline__ bool launch_is_mixed(const uint32_t *__restrict__ select)
{
    return __builtin_popcount(select[7] & 7u) >= 2;
}

// (round 5) No reset kernel in front of it: every workgroup leaves its three numbers in a slot of its own, the
// workgroup that arrives last (a counter that it sets back to zero: the only word that has to be zero between
// launches, and it is again when the kernel ends -- replays of a captured graph included) adds them up, writes the
// totals with plain stores and clears the words the kernels behind it count in.
__global__ __launch_bounds__(256) void sample_compressibility_kernel(
    const uint8_t *__restrict__ src, uint64_t in_stride, const uint32_t *__restrict__ in_len,
    uint64_t total_len, uint32_t n_blocks, uint32_t *__restrict__ select, uint32_t small_launch)
{
    __shared__ uint32_t bitmap[2048];  // 65536 bits
    __shared__ uint32_t class_bm[384];
    __shared__ uint32_t wg_dups, wg_class, wg_last;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < 2048; i += 256)
        bitmap[i] = 0;
    for (uint32_t i = t; i < 384; i += 256)
        class_bm[i] = 0;
    if (t == 0) {
        wg_dups = 0;
        wg_class = 0;
    }
    __syncthreads();
    // one block out of each of gridDim.x equal stretches of the launch, at a place inside its stretch that follows no
    // period (evenly spaced samples met only the text blocks of a launch whose every 16th block was text)
    const uint32_t b0 = (uint32_t)(((uint64_t)blockIdx.x * n_blocks) / gridDim.x);
    const uint32_t b1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_blocks) / gridDim.x);
    const uint32_t b = b0 + (b1 > b0 ? ((blockIdx.x * 0x9e3779b1u) >> 12) % (b1 - b0) : 0u);
    uint32_t n = block_length(in_len, total_len, in_stride, b);
    const uint8_t *in = src + (uint64_t)b * in_stride;
    n = n > kMaxBlock ? 0 : n;
    const uint32_t n1 = n < 1024 ? n : 1024;  // text repeats >= 23 % of its shingles within 1 KiB
    const uint32_t npos = n1 >= 4 ? n1 - 3 : 0;
    if (t < 64) {  // wave 0: the cost class of the block (does routing / ordering the launch's blocks pay?)
        const uint32_t c = block_cost_class(in, n, t, class_bm);
        if (t == 0)
            wg_class = 1u << c;
    } else {       // waves 1..3: repeats among the shingles of its first KiB
        uint32_t dups = 0;
        for (uint32_t p = t - 64; p < npos; p += 192) {
            const uint32_t h = (ld32(in + p) * kHashMul) >> 16;
            const uint32_t bit = 1u << (h & 31u);
            dups += (atomicOr(&bitmap[h >> 5], bit) & bit) ? 1u : 0u;
        }
        atomicAdd(&wg_dups, dups);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t *mine = select + 16 + 3 * blockIdx.x;
        __hip_atomic_store(mine + 0, wg_dups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, npos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, wg_class, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t arrived = __hip_atomic_fetch_add(select + 15, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        wg_last = arrived == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (wg_last && t < 64) {  // (gridDim.x <= 64)
        uint32_t d = 0, p = 0, c = 0;
        if (t < gridDim.x) {
            const uint32_t *its = select + 16 + 3 * t;
            d = __hip_atomic_load(its + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = __hip_atomic_load(its + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = __hip_atomic_load(its + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        d = lane_get(wave_inclusive_sum(d), 63);
        p = lane_get(wave_inclusive_sum(p), 63);
        c = (ballot64((c & 1u) != 0) ? 1u : 0u) | (ballot64((c & 2u) != 0) ? 2u : 0u) | (ballot64((c & 4u) != 0) ? 4u : 0u);
        // A small launch (fewer than two blocks per SIMD) that mixes classes: the single kernel takes all of it, as in
        // rounds 1-4 -- with a handful of expensive blocks, the persistent kernel's lone waves would take most of them and
        // each needs twice a team's time (mix of 1 536 blocks: 1.78 ms by the single kernel, 2.29 ms routed).
        if (small_launch && __builtin_popcount(c & 7u) >= 2) {
            d = 0;
            c = 1u | 8u;  // (bit 3 of select[7]: the single kernel takes every block of the launch, compressible ones too: its instance with the fast dense path)
        }
        if (t < 16)
            select[t] = t == 0 ? d : t == 1 ? p : t == 7 ? c : 0u;  // (select[15]: the arrival counter, zero again)
    }
}

// ---- blocks, not launches, are routed (round 5), and the expensive ones go first (round 3)
// A launch whose sample has met more than one cost class (launch_is_mixed) is compressed by BOTH kernels: the single
// kernel is started for every block, works out the block's class itself (block_cost_class: ~1 us of one wave) and
// keeps the blocks of class 0 -- no repeats among the shingles of their first KiB: the scan accelerates, the block
// streams through the register file, 26 us -- while a block of class 1 or 2 is put into the persistent kernel's queue
// (one atomic) and left; the team kernel, next on the stream, takes the queue: class 2 first (dense parse of the
// whole block: a 64 KiB block of text takes ~100 x as long as an incompressible one), then class 1 (few distinct
// strings among 128 samples spread over the block: long matches skip most of it).  Blocks cannot be split, so a
// mixed launch ends with its last expensive blocks running alone, and the lever is their order.  No host
// synchronisation, graph-capturable; a wrong class only costs speed, every kernel writes the same bytes.
// A launch of one class runs as before: one kernel for all its blocks by the sample's verdict, no queue.
//   select[4 + c] = blocks of class c in the queue (c = 1, 2); queue[0 .. half) class 2, queue[half ..) class 1.
// (Rounds 3-4 ordered the queue with two kernels of their own and left the incompressible third of a mixed launch
// to the persistent kernel's general code at 145 us a block; a compressible minority in an incompressible launch
// went through the single kernel's instance without the fast dense path at half the team kernel's rate.)
__device__ __forceinline__ bool launch_is_routed(const uint32_t *__restrict__ select, uint32_t queue_half)
{
    return queue_half != 0 && launch_is_mixed(select);
}

// Single kernel: one 64-lane workgroup per block, table in LDS, whole block in registers.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void compress_blocks_kernel(
    const uint8_t *__restrict__ src, uint64_t in_stride, const uint32_t *__restrict__ in_len,
    uint64_t total_len, uint32_t n_blocks, uint8_t *__restrict__ dst, uint64_t dst_stride,
    uint32_t *__restrict__ out_len, uint32_t flags, uint32_t *__restrict__ select, uint32_t *__restrict__ queue,
    uint32_t queue_half)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds_tab[kMaxTable];  // the whole LDS budget: 32 KiB
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks)
        return;
    const bool routed = select && launch_is_routed(select, queue_half);
    // (read here, with the verdict: the same word behind the routed launch's classifier cost a mixed launch 5 % -- mix 156 -> 149 GB/s)
    const bool lean_instance = select && (select[7] & 8u) == 0;
    if (select && !routed && prefers_dual(select)) {  // auto mode, one kind of block: the sampled data is compressible, the team kernel runs
        // (every block a persistent kernel is to take starts out POISONED: if every team and lone wave of that grid should
        // give up -- ADVICE r4 --, the blocks nobody took read 0xfffffffe, not what the caller's buffer held before)
        out_len[b] = 0xfffffffeu;
        return;
    }
    const uint32_t n = block_length(in_len, total_len, in_stride, b);
    if (n > kMaxBlock) {  // caller error: flagged, nothing written
        out_len[b] = 0xffffffffu;  // (every lane, same address, same value: see the note at the dual kernel)
        return;
    }
    if (routed) {
        // a mixed launch: this block's own class decides who compresses it (the table's LDS is the classifier's scratch)
        uint32_t *scratch = reinterpret_cast<uint32_t *>(lds_tab);
        for (uint32_t i = lane; i < 384; i += 64)
            scratch[i] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t c = uni(block_cost_class(src + (uint64_t)b * in_stride, n, lane, scratch));
        if (c != 0) {
            if (lane == 0)
                queue[(c == 2 ? 0u : queue_half) + atomicAdd(&select[4 + c], 1u)] = b;
            out_len[b] = 0xfffffffeu;  // (poisoned until the team kernel has compressed it: see above)
            return;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const HashTab<true> tab{lds_tab, nullptr, nullptr, nullptr};
    uint32_t op;
    if (lean_instance)  // auto mode, and the sample (or this block's class) says incompressible: the instance without the fast dense path
        op = compress_one_block<64, false, HashTab<true>, false>(tab, src + (uint64_t)b * in_stride, n,
                                                                         dst + (uint64_t)b * dst_stride, flags, lane);
    else
        op = compress_one_block<64, false>(tab, src + (uint64_t)b * in_stride, n, dst + (uint64_t)b * dst_stride,
                                           flags, lane);
    out_len[b] = EXP_SINGLE_MARK(op);  // (a9) every lane, same address, same value
}

// Multi-wave ("dual") kernel: workgroups of kGroupWaves independent wavefronts, each compressing its
// own blocks, four workgroups per CU, so that every SIMD hosts kGroupWaves waves and their
// instruction latencies overlap (a lone wave issues ~1 instruction per 10 cycles).  Wave 0 keeps
// its table in LDS; the others have theirs in global memory (one 32 KiB slot each in `scratch`).  Persistent grid: all waves pull block indices from one device-wide counter (zeroed
// by a tiny kernel ahead of the launch), so the faster (LDS-table) waves take more of them and
// mixed workloads balance.  Measured on 64 KiB text blocks: 2 waves 42 GB/s, 3 waves 53, 4 waves 45,
// 5 waves 42 (with three the SIMDs already have an instruction in flight 95 % of the time; more
// waves only add L2 misses on the global tables); one wave per SIMD (single kernel) 32.  The block
// is not kept in registers here (incompressible data goes to the single kernel).
#ifndef SZ_GROUP_WAVES
#define SZ_GROUP_WAVES 3  // waves per workgroup = waves per SIMD (one LDS-table wave + global-table waves)
#endif
static constexpr int kGroupWaves = SZ_GROUP_WAVES;
static constexpr int kDualChunks = kGroupWaves == 2 ? 24 : 0;
static constexpr bool kDualLean = kGroupWaves >= 3;
__global__ __launch_bounds__(64 * kGroupWaves) __attribute__((amdgpu_waves_per_eu(kGroupWaves, kGroupWaves))) void compress_blocks_dual_kernel(
    const uint8_t *__restrict__ src, uint64_t in_stride, const uint32_t *__restrict__ in_len,
    uint64_t total_len, uint32_t n_blocks, uint8_t *__restrict__ dst, uint64_t dst_stride,
    uint32_t *__restrict__ out_len, uint32_t flags, uint16_t *scratch_tables,
    uint32_t *__restrict__ select)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds_tab[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint32_t dup_bm[(kGroupWaves - 1) * kGlobLdsWords];  // per global-table wave: scratch bitmap + occupancy bitmap
    // (only ever launched by name -- SZ_KERNEL=dual --: every block of the launch, in index order)
    const uint32_t n_work = n_blocks;
    if (blockIdx.x * kGroupWaves >= n_work)
        return;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= 1)
        for (uint32_t i = lane; i < kGlobLdsWords; i += 64)
            dup_bm[(wave - 1) * kGlobLdsWords + i] = 0;
    const uint32_t my_slot = blockIdx.x * (kGroupWaves - 1) + (wave ? wave - 1 : 0);
    uint16_t *my_glob = scratch_tables + (size_t)my_slot * kMaxTable;
    // The waves whose table is in LDS go first at the issue port: they wait least (no table access
    // leaves the CU), so every slot they take is used; the others spend most of their time waiting
    // for their table in the L2 anyway (measured: text +2.7 %, low-entropy +2.4 %, mix unchanged).
    if (wave == 0)
        __builtin_amdgcn_s_setprio(kLdsWavePrio);
    for (;;) {
        // One queue for the whole grid: faster waves take more blocks.  The counter's address goes
        // through VGPRs on purpose: experimental builds of this kernel that hung had `select` spilled
        // to a VGPR lane and reloaded (v_readlane + s_nop 4) right before the atomic's scalar base;
        // that was not the whole story, but the pattern is kept out of the
        // kernels (tests/test_abi.py scans their ISA for it).
        uint32_t b = 0;
        if (lane == 0) {
            uint64_t qa = (uint64_t)(uintptr_t)(select + 2);
            uint32_t qlo = (uint32_t)qa, qhi = (uint32_t)(qa >> 32);
            asm volatile("" : "+v"(qlo), "+v"(qhi));
            b = atomicAdd(reinterpret_cast<uint32_t *>((uintptr_t)(((uint64_t)qhi << 32) | qlo)), 1u);
        }
        b = uni(b);
        if (b >= n_work)
            break;
        const uint32_t n = block_length(in_len, total_len, in_stride, b);
        if (n > kMaxBlock) {
            out_len[b] = 0xffffffffu;
            continue;
        }
        uint32_t op;
        if (wave >= 1) {
            const HashTab<false> tab{nullptr, my_glob, dup_bm + (wave - 1) * kGlobLdsWords + kGlobOccWords, dup_bm + (wave - 1) * kGlobLdsWords};
            op = compress_one_block<kDualChunks, kDualLean>(tab, src + (uint64_t)b * in_stride, n,
                                                            dst + (uint64_t)b * dst_stride, flags, lane);
        } else {
            const HashTab<true> tab{lds_tab, nullptr, nullptr, nullptr};
            op = compress_one_block<kDualChunks, kDualLean>(tab, src + (uint64_t)b * in_stride, n,
                                                            dst + (uint64_t)b * dst_stride, flags, lane);
        }
        // (a9) Written by every lane -- same address, same value, one transaction -- not by lane 0
        // alone: builds of this kernel in which the compiler merged the two lane-0 stores of this
        // loop into one tail with a per-lane address hung the whole grid on gfx950 (every wave stalled
        // at the end of its first block, its output written, its length not); with uniform stores the
        // same source ran.  DESIGN.md section 7.
        out_len[b] = op;
    }
}

// Team kernel (round 3): workgroups of kTeamWaves wavefronts that compress ONE block together, sharing one
// hash table in LDS (see "wave teams" above), plus -- HYBRID -- kLoneWaves wavefronts that compress blocks of
// their own with their table in global memory (the multi-wave kernel's extra waves); four workgroups per CU,
// persistent grid over one device-wide block counter.  A team is through a block ~2.5 x sooner than a lone wave
// and uses the CU's four LDS tables accordingly better; the lone waves fill the issue slots the teams leave
// (a team's waves wait for their token a third of the time).
#ifndef SZ_TEAM_WAVES
#define SZ_TEAM_WAVES 2  // waves of a team (2..4)
#endif
static constexpr int kLoneWavesDefault = 2;  // global-table waves beside the team in each workgroup (0..2)
static constexpr int kTeamWaves = SZ_TEAM_WAVES > 1 ? SZ_TEAM_WAVES : 2;
static constexpr int kLoneWaves = kLoneWavesDefault;
static constexpr int kHybridWaves = kTeamWaves + kLoneWaves;
static constexpr int kHybridPerSimd = kHybridWaves;  // four workgroups per CU, one wave of each per SIMD
// The team kernel's arguments as ONE struct passed by value: it lies at offset 0 of the kernel-argument segment, and
// the persistent loops read its fields from there (scalar loads through a laundered pointer) where they need them
// instead of holding ~22 scalar registers of arguments across the per-block code, which has none to spare (the
// parse state alone fills the scalar file: 116-124 scalar spills in this kernel).
struct TeamArgs {
    const uint8_t *src;
    uint64_t in_stride;
    const uint32_t *in_len;
    uint64_t total_len;
    uint8_t *dst;
    uint64_t dst_stride;
    uint32_t *out_len;
    uint16_t *scratch_tables;
    uint32_t *select;
    const uint32_t *queue;  // routed launches: block indices, class 2 from [0], class 1 from [queue_half]
    uint32_t n_blocks, flags, auto_mode;  // auto_mode bit 0: the launch was sampled (else: every block, by name); bit 1: no adoption at the launch's end
    uint32_t queue_half;    // 0: no routing
};
__global__ __launch_bounds__(64 * kHybridWaves) __attribute__((amdgpu_waves_per_eu(kHybridPerSimd, kHybridPerSimd))) void compress_blocks_team_kernel(TeamArgs args_in_kernarg)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds_tab[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint64_t tokens[2];
    __shared__ __attribute__((aligned(16))) uint8_t dup_scr[kTeamWaves * 256];
    __shared__ __attribute__((aligned(16))) uint32_t lone_bm[(kLoneWaves > 0 ? kLoneWaves : 1) * kLoneLdsWords];  // per global-table wave: occupancy bitmap + scratch bitmap + adoption words
    __shared__ uint32_t next_block, team_bar, team_dead, len_claim;
    // THE END OF A LAUNCH (round 6).  The queue runs empty while every lone wave is inside a block that it needs
    // ~3.8 ms for (a team: 1.75): round 5 ended a launch of text with ~2.9 ms of idle teams.  A team that finds
    // the queue empty ADOPTS the block of a lone wave of its own workgroup: it asks (the lone wave's request word,
    // behind its bitmaps in lone_bm), the lone wave hands the block over at the end of its tile -- parse state into its
    // record; its table is in memory, the bitmap of the buckets it has written in this very LDS -- and leaves; the team
    // copies the table into its own (32 KiB out of the L2) and goes on from that state at a team's pace: the same bytes by
    // construction (the state is the one the fast dense path hands to the general code; tests/test_gpu_adoption.py).
    // Nothing is compressed twice, a length has one writer.  The loop at the end of the kernel.
    __shared__ uint32_t adopt_sel;
    auto lone_word = [&](uint32_t w, uint32_t k) -> uint32_t * { return lone_bm + w * kLoneLdsWords + k; };
    (void)args_in_kernarg;
    using KArgs = const __attribute__((address_space(4))) TeamArgs;
    auto ka = [&]() -> KArgs * {  // a fresh look at the arguments: what is read through it is not kept alive from an earlier look
        KArgs *p = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
    // routed (a mixed launch): the queue the single kernel has filled, expensive class first; else every block of the
    // launch in index order -- if the sample says compressible
    const bool use_queue = (ka()->auto_mode & 1u) != 0 && launch_is_routed(ka()->select, ka()->queue_half);
    if ((ka()->auto_mode & 1u) != 0 && !use_queue && !prefers_dual(ka()->select))  // auto mode: incompressible sample, the single kernel ran
        return;
    const uint32_t n_q2 = use_queue ? uni(ka()->select[6]) : 0u;
    const uint32_t n_work = use_queue ? n_q2 + uni(ka()->select[5]) : uni(ka()->n_blocks);
    if (blockIdx.x * (1u + kLoneWaves) >= n_work)
        return;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    auto next_index = [&]() -> uint32_t {  // one queue for the whole grid (address through VGPRs: see the multi-wave kernel)
        uint32_t b0 = 0;
        if (lane == 0) {
            uint64_t qa = (uint64_t)(uintptr_t)(ka()->select + 2);
            uint32_t qlo = (uint32_t)qa, qhi = (uint32_t)(qa >> 32);
            asm volatile("" : "+v"(qlo), "+v"(qhi));
            b0 = atomicAdd(reinterpret_cast<uint32_t *>((uintptr_t)(((uint64_t)qhi << 32) | qlo)), 1u);
        }
        return uni(b0);
    };
    auto queued = [&](uint32_t i) -> uint32_t { return use_queue ? uni(ka()->queue[i < n_q2 ? i : ka()->queue_half + (i - n_q2)]) : i; };
    auto store_len = [&](uint32_t b, uint32_t v) { ka()->out_len[b] = v; };  // every lane, same address, same value
    if (threadIdx.x == 0) {
        __hip_atomic_store(&team_bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&team_dead, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&len_claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&adopt_sel, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int w = 0; w < kLoneWaves; w++) {
            __hip_atomic_store(lone_word(w, kLoneReqWord), kAdoptIdle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(lone_word(w, kLoneStateWord), kLoneRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();  // (the only workgroup-wide barrier: before the waves part ways)
    if (kLoneWaves > 0 && wave >= kTeamWaves) {
        // ---- a lone wave: blocks of its own, table in global memory (as the multi-wave kernel's extra waves)
        const uint32_t lw = wave - kTeamWaves;
        for (uint32_t i = lane; i < kGlobLdsWords; i += 64)
            lone_bm[lw * kLoneLdsWords + i] = 0;
        for (;;) {
            uint32_t b = next_index();
            if (b >= n_work)
                break;
            b = queued(b);
            // (what is read through the laundered pointer is uniform -- scalar loads from the argument segment -- but
            // not PROVABLY so for the compiler's uniformity analysis: said so here, tests/test_abi.py checks the proof)
            KArgs *a = ka();
            const uint32_t n = uni(block_length(a->in_len, a->total_len, a->in_stride, b));
            if (n > kMaxBlock) {
                store_len(b, 0xffffffffu);
                continue;
            }
            uint16_t *my_glob = as_global<uint16_t>(uni64((uint64_t)(uintptr_t)a->scratch_tables)) +
                                (size_t)(blockIdx.x * (kLoneWaves > 0 ? kLoneWaves : 1) + lw) * kMaxTable;
            const HashTab<false> gtab{nullptr, my_glob, lone_bm + lw * kLoneLdsWords + kGlobOccWords, lone_bm + lw * kLoneLdsWords};
            const uint8_t *in_b = as_global<const uint8_t>(uni64((uint64_t)(uintptr_t)a->src + (uint64_t)b * a->in_stride));
            uint8_t *out_b = as_global<uint8_t>(uni64((uint64_t)(uintptr_t)a->dst + (uint64_t)b * a->dst_stride));
            EXP_STARTED(b);
            const uint32_t op = compress_one_block<0, true, HashTab<false>, true, kMatchPiecesDefault, NoTeam, LoneAdopt>(
                gtab, in_b, n, out_b, uni(a->flags), lane);
            if (op == 0xfffffffeu && uni(__hip_atomic_load(lone_word(lw, kLoneReqWord), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == kAdoptTaken) {
                // handed over: the team of this workgroup goes on with the block (the queue is empty: it asked)
                if (lane == 0)
                    *lone_word(lw, kLoneRecWord + kAdoptRecBlock) = b;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(lone_word(lw, kLoneStateWord), kLoneHanded, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
            store_len(b, EXP_LONE_MARK(op));
            EXP_FINISHED(b);
        }
        __hip_atomic_store(lone_word(lw, kLoneStateWord), kLoneLeft, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const HashTab<true> tab{lds_tab, nullptr, nullptr, nullptr};
    WaveTeam<kTeamWaves> team{&tokens[0], &tokens[1], wave, nullptr, dup_scr + wave * 256u, &team_bar, nullptr};
    __builtin_amdgcn_s_setprio(kTeamBasePrio);  // (a team's waves go before the lone ones; the token's holder before both)
    // Giving up (a wave of the team stalled for seconds: SZ_TEAM_SPIN_CAP): the wave that gives up raises
    // `team_dead` and every wave of the team leaves the persistent loop at its next look at it -- the barrier's
    // arrival count is not trusted again.  The block in hand is poisoned (0xfffffffe).  out_len[b] has ONE writer
    // per block: the first wave to claim `len_claim` (the finisher with the length, or a wave that gave up with
    // the poison); the blocks this team has not taken stay in the grid's queue for the other workgroups.
    for (;;) {
        if (uni(__hip_atomic_load(&team_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0)
            break;
        if (wave == 0) {
            const uint32_t b0 = next_index();
            __hip_atomic_store(&next_block, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t lost = team_barrier<kTeamWaves, true>(&team_bar, &team_dead, lane);  // every wave of the team is past the block before (it left on the "done" token)
        uint32_t b = uni(__hip_atomic_load(&next_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (wave == 0 && lost == 0) {  // only now: a wave still polling for "done" must not find an empty token instead
            __hip_atomic_store(&tokens[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&tokens[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&len_claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lost == 0)
            lost = team_barrier<kTeamWaves, true>(&team_bar, &team_dead, lane);  // (everyone has the index and sees fresh tokens before any wave goes on)
        if (__builtin_expect(lost != 0, 0)) {
            __hip_atomic_store(&team_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == 0 && b < n_work)  // (wave 0 took this index out of the queue: nobody else will report on it)
                store_len(queued(b), 0xfffffffeu);
            break;
        }
        if (b >= n_work)
            break;
        b = queued(b);
        KArgs *a = ka();
        const uint32_t n = uni(block_length(a->in_len, a->total_len, a->in_stride, b));
        if (n > kMaxBlock) {
            if (wave == 0)
                store_len(b, 0xffffffffu);
            continue;
        }
        TRACE_ATTACH(team, b, wave, kTeamWaves);
        EXP_STARTED(b);
        const uint8_t *in_b = as_global<const uint8_t>(uni64((uint64_t)(uintptr_t)a->src + (uint64_t)b * a->in_stride));
        uint8_t *out_b = as_global<uint8_t>(uni64((uint64_t)(uintptr_t)a->dst + (uint64_t)b * a->dst_stride));
        const uint32_t op = compress_one_block<0, true, HashTab<true>, true, kMatchPiecesDefault, WaveTeam<kTeamWaves>>(
            tab, in_b, n, out_b, uni(a->flags), lane, team);
        if (op != kTeamNotMine) {
            // (a9) the wave that finished the block, or the first wave that gave it up: every lane, same address, same value
            // (every lane adds one; exactly one lane of one wave reads 0, and that wave has the claim -- whichever
            // lane it is: nothing is assumed about the order in which an instruction's lanes are served)
            const bool mine = __ballot(__hip_atomic_fetch_add(&len_claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) != 0;
            if (op == kTeamLost)
                __hip_atomic_store(&team_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (mine)
                store_len(b, op == kTeamLost ? 0xfffffffeu : op);
            if (op == kTeamLost)
                break;
        }
        if (op != kTeamNotMine && op != kTeamLost)
            EXP_FINISHED(b);
    }
    // ---- the end of the launch: adoption.  The queue is empty (a team that left its loop for another reason has raised
    // `team_dead`); a lone wave of this workgroup that is still inside a block is asked for it, hands it over where its
    // tile ends and leaves; the team copies its table and goes on from its state -- with an instance of the per-block
    // code of its own (TeamAdopt), in a loop of its own BEHIND the persistent one: the same steps inside that loop
    // (inline, or as functions the kernel calls) cost the blocks of the queue 2-8 % in spills and calling conventions
    // (EXPERIMENTS.md R6.6); what is not on their path cannot.
    if (kLoneWaves == 0 || (ka()->auto_mode & 2u) != 0)  // (SZ_NO_ADOPT=1: the end of a launch as in round 5, for A/B runs)
        return;
    for (;;) {
        if (uni(__hip_atomic_load(&team_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0)
            break;
        if (wave == 0) {
            uint32_t b0 = 0, sel = 0;  // sel: 1 + the lone wave whose block b0 is
            for (uint32_t w = 0; w < (uint32_t)kLoneWaves && sel == 0; w++) {
                uint32_t *const state = lone_word(w, kLoneStateWord), *const req = lone_word(w, kLoneReqWord);
                uint32_t st = uni(__hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (st == kLoneRunning) {
                    __hip_atomic_store(req, kAdoptAsked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // (it answers at the end of its tile, ~4 us, or not before the end of its block: a long match, the
                    // general code, too little left.  Bounded all the same -- well inside the cap of the team mate that
                    // waits at the barrier below --: the request is withdrawn, one compare-and-swap against the lone wave's
                    // own, and only a wave that had taken it just then is waited for, for the few stores it is about to make.)
                    for (uint32_t spin = 0; spin < (SZ_TEAM_SPIN_CAP >> 6) && st == kLoneRunning; spin++) {
                        __builtin_amdgcn_s_sleep(8);
                        st = uni(__hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                    }
                    if (st == kLoneRunning) {
                        uint32_t was = kAdoptIdle;
                        if (lane == 0) {
                            uint32_t expect = kAdoptAsked;
                            (void)__hip_atomic_compare_exchange_strong(req, &expect, kAdoptWithdrawn, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            was = expect;
                        }
                        if (uni(was) == kAdoptTaken)
                            for (uint32_t spin = 0; spin < (1u << 22) && st == kLoneRunning; spin++) {
                                __builtin_amdgcn_s_sleep(8);
                                st = uni(__hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                            }
                    }
                }
                if (st == kLoneHanded) {
                    sel = 1u + w;
                    b0 = uni(__hip_atomic_load(lone_word(w, kLoneRecWord + kAdoptRecBlock), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    __hip_atomic_store(state, kLoneAdopted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __hip_atomic_store(&next_block, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&adopt_sel, sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t lost = team_barrier<kTeamWaves, true>(&team_bar, &team_dead, lane);  // every wave of the team is past the block before
        const uint32_t b = uni(__hip_atomic_load(&next_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        const uint32_t sel = uni(__hip_atomic_load(&adopt_sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (wave == 0 && lost == 0) {
            __hip_atomic_store(&tokens[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&tokens[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&len_claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lost == 0)
            lost = team_barrier<kTeamWaves, true>(&team_bar, &team_dead, lane);
        if (__builtin_expect(lost != 0, 0)) {
            __hip_atomic_store(&team_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == 0 && sel != 0)  // (wave 0 took this block from its lone wave: nobody else will report on it)
                store_len(b, 0xfffffffeu);
            break;
        }
        if (sel == 0)
            break;  // no lone wave of this workgroup is inside a block any more
        KArgs *a = ka();
        const uint32_t n = uni(block_length(a->in_len, a->total_len, a->in_stride, b));
        EXP_STARTED(b);
        {
            // The table the lone wave has built so far becomes the team's: its entries from memory (written through to the
            // L2 this CU reads from; agent-scope loads, as that wave's own look-ups), a bucket it has not written in this
            // block -- its table is never cleared -- reads as empty (its bitmap, in this workgroup's LDS, says which).  Both
            // waves, a slice each; compress_one_block's barrier behind its (skipped) clearing pass stands behind this copy.
            const uint32_t *occ = lone_bm + (sel - 1u) * kLoneLdsWords;
            const uint64_t *g = reinterpret_cast<const uint64_t *>(
                as_global<uint16_t>(uni64((uint64_t)(uintptr_t)a->scratch_tables)) +
                (size_t)(blockIdx.x * (kLoneWaves > 0 ? kLoneWaves : 1) + (sel - 1u)) * kMaxTable);
            uint64_t *l = reinterpret_cast<uint64_t *>(lds_tab);
#pragma unroll 4
            for (uint32_t i = wave * 64u + lane; i < kMaxTable / 4u; i += 64u * kTeamWaves) {
                const uint64_t v = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t m4 = occ[i >> 3] >> ((i & 7u) << 2);
                const uint64_t keep = ((m4 & 1u) ? 0xffffull : 0ull) | ((m4 & 2u) ? 0xffffull << 16 : 0ull) |
                                      ((m4 & 4u) ? 0xffffull << 32 : 0ull) | ((m4 & 8u) ? 0xffffull << 48 : 0ull);
                l[i] = v & keep;
            }
        }
        team.arec = lone_word(sel - 1u, kLoneRecWord);
        const uint8_t *in_b = as_global<const uint8_t>(uni64((uint64_t)(uintptr_t)a->src + (uint64_t)b * a->in_stride));
        uint8_t *out_b = as_global<uint8_t>(uni64((uint64_t)(uintptr_t)a->dst + (uint64_t)b * a->dst_stride));
        const uint32_t op = compress_one_block<0, true, HashTab<true>, true, kMatchPiecesDefault, WaveTeam<kTeamWaves>, TeamAdopt>(
            tab, in_b, n, out_b, uni(a->flags), lane, team);
        if (op != kTeamNotMine) {  // (a9) one writer, as above
            const bool mine = __ballot(__hip_atomic_fetch_add(&len_claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) != 0;
            if (op == kTeamLost)
                __hip_atomic_store(&team_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (mine)
                store_len(b, op == kTeamLost ? 0xfffffffeu : EXP_ADOPT_MARK(op));
            if (op == kTeamLost)
                break;
            EXP_FINISHED(b);
        }
    }
}

static constexpr uint32_t kSmallBlockMax = 8192;

// Small fixed-size blocks (<= 8 KiB each, e.g. 4 KiB pages): the table is 2 * T bytes of dynamic LDS
// (T = 256 .. 8192 entries for the block size), so many more blocks fit a CU than with the 32 KiB
// of the kernels above; lean per-block code, at least four waves per SIMD, tables in LDS for all.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void compress_small_blocks_kernel(
    const uint8_t *__restrict__ src, uint32_t block_bytes, uint32_t n_blocks, uint8_t *__restrict__ dst,
    uint64_t dst_stride, uint32_t *__restrict__ out_len, uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t dyn_tab[];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks)
        return;
    const HashTab<true> tab{dyn_tab, nullptr, nullptr, nullptr};
    // (1 KiB steps of match extension: the blocks are small and the kernel runs four waves per SIMD)
    const uint32_t op = compress_one_block<0, true, HashTab<true>, true, 1>(tab, src + (uint64_t)b * block_bytes, block_bytes,
                                                    dst + (uint64_t)b * dst_stride, flags, lane);
    out_len[b] = op;  // (a9) every lane, same address, same value
}

// ------------------------------------------------------------------------------ launchers

// sz_profile_next_launch: events around the two kernels that can be a launch's dominant one (see snappy_kernels.h)
struct ProfileEvents {
    hipEvent_t single0 = nullptr, single1 = nullptr, multi0 = nullptr, multi1 = nullptr;
};
static thread_local ProfileEvents tl_profile;
void set_profile_events(hipEvent_t single0, hipEvent_t single1, hipEvent_t multi0, hipEvent_t multi1)
{
    tl_profile = ProfileEvents{single0, single1, multi0, multi1};
}
void drop_profile_events() { tl_profile = ProfileEvents{}; }

static hipError_t launch_any(const void *d_src, uint64_t in_stride, const uint32_t *d_in_len, uint64_t total_len,
                             uint32_t n_blocks, void *d_dst, uint64_t dst_stride, uint32_t *d_out_len,
                             uint32_t flags, const DualScratch *dual, hipStream_t stream)
{
    const uint8_t *src = static_cast<const uint8_t *>(d_src);
    uint8_t *dst = static_cast<uint8_t *>(d_dst);
    const ProfileEvents prof = tl_profile;  // (one shot)
    tl_profile = ProfileEvents{};
    // fixed small blocks: their own kernel (measured: 2.4x on incompressible 4 KiB pages, 1.6x on text
    // pages, 1.45x / 1.0x at 8 KiB; from 16 KiB on its 32 KiB table leaves one wave per SIMD and it loses)
    if (!d_in_len && total_len == 0 && in_stride <= kSmallBlockMax) {
        uint32_t T = 256;
        while (T < 16384 && T < in_stride)
            T <<= 1;
        if (prof.single0)  // (the one kernel of such a launch: the "one wavefront per block" pair)
            (void)hipEventRecord(prof.single0, stream);
        hipLaunchKernelGGL(compress_small_blocks_kernel, dim3(n_blocks), dim3(64), 2 * T, stream, src,
                           (uint32_t)in_stride, n_blocks, dst, dst_stride, d_out_len, flags);
        if (prof.single1)
            (void)hipEventRecord(prof.single1, stream);
        return hipGetLastError();
    }
    const bool dual_ok = dual && dual->tables && n_blocks >= dual->min_blocks;
    const bool sampled = dual_ok && dual->mode == kDualAuto;
    // blocks of a mixed launch are routed one by one (the single kernel queues what it leaves to the team kernel)
    const uint32_t queue_half = sampled && dual->kernel == kKernelTeam && dual->queue && n_blocks <= dual->queue_cap / 2 ? dual->queue_cap / 2 : 0u;
    if (dual_ok && !sampled)  // a kernel named by the caller: launch state := 0.  (A kernel, not hipMemsetAsync: captured
                              // into a hipGraph, the memset node made every replay after the first a no-op.)
        hipLaunchKernelGGL(reset_launch_state_kernel, dim3(1), dim3(64), 0, stream, dual->select);
    if (sampled) {
        // sample (it also resets the launch state) -> both kernels, each checks the verdict first (graph-capture
        // safe: no host sync)
        const uint32_t ns = n_blocks < 64 ? n_blocks : 64;
        hipLaunchKernelGGL(sample_compressibility_kernel, dim3(ns), dim3(256), 0, stream, src, in_stride,
                           d_in_len, total_len, n_blocks, dual->select, n_blocks < 2u * dual->n_slots ? 1u : 0u);
    }
    if (!dual_ok || sampled) {
        if (prof.single0)
            (void)hipEventRecord(prof.single0, stream);
        hipLaunchKernelGGL(compress_blocks_kernel, dim3(n_blocks), dim3(64), 0, stream, src, in_stride, d_in_len,
                               total_len, n_blocks, dst, dst_stride, d_out_len, flags,
                               sampled ? dual->select : (uint32_t *)nullptr, queue_half ? dual->queue : (uint32_t *)nullptr,
                               queue_half);
        if (prof.single1)
            (void)hipEventRecord(prof.single1, stream);
    }
    if (dual_ok) {
        // enough blocks to keep kGroupWaves waves per SIMD busy on every CU: persistent grid
        if (dual->kernel == kKernelTeam) {
            uint32_t tgrid = (n_blocks + kLoneWaves) / (1u + kLoneWaves);
            tgrid = tgrid < dual->n_slots ? tgrid : dual->n_slots;
            static_assert(kLoneWaves <= (int)kDualTablesPerGroup, "the workspace has kDualTablesPerGroup tables per workgroup");
            const TeamArgs targs{src, in_stride, d_in_len, total_len, dst, dst_stride, d_out_len, dual->tables, dual->select,
                                 dual->queue, n_blocks, flags, (sampled ? 1u : 0u) | (getenv("SZ_NO_ADOPT") ? 2u : 0u), queue_half};
            if (prof.multi0)
                (void)hipEventRecord(prof.multi0, stream);
            hipLaunchKernelGGL(compress_blocks_team_kernel, dim3(tgrid), dim3(64 * kHybridWaves), 0, stream, targs);
            if (prof.multi1)
                (void)hipEventRecord(prof.multi1, stream);
            return hipGetLastError();
        }
        uint32_t grid = (n_blocks + kGroupWaves - 1) / kGroupWaves;
        grid = grid < dual->n_slots ? grid : dual->n_slots;
        if (prof.multi0)
            (void)hipEventRecord(prof.multi0, stream);
        hipLaunchKernelGGL(compress_blocks_dual_kernel, dim3(grid), dim3(64 * kGroupWaves), 0, stream, src, in_stride,
                           d_in_len, total_len, n_blocks, dst, dst_stride, d_out_len, flags, dual->tables,
                           dual->select);
        if (prof.multi1)
            (void)hipEventRecord(prof.multi1, stream);
    }
    return hipGetLastError();
}

hipError_t launch_compress_blocks(const void *d_src, uint64_t in_stride, const uint32_t *d_in_len,
                                  uint32_t n_blocks, void *d_dst, uint64_t dst_stride,
                                  uint32_t *d_out_len, uint32_t flags, const DualScratch *dual,
                                  hipStream_t stream)
{
    if (n_blocks == 0)
        return hipSuccess;
    return launch_any(d_src, in_stride, d_in_len, 0, n_blocks, d_dst, dst_stride, d_out_len, flags, dual, stream);
}

hipError_t launch_compress_stream(const void *d_src, uint64_t n, void *d_slots, uint64_t slot_stride,
                                  uint32_t *d_len, uint32_t flags, const DualScratch *dual, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    const uint32_t n_blocks = (uint32_t)((n + kMaxBlock - 1) / kMaxBlock);
    return launch_any(d_src, kMaxBlock, nullptr, n, n_blocks, d_slots, slot_stride, d_len, flags, dual, stream);
}

}  // namespace szk
