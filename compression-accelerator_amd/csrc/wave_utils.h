// wave_utils.h -- device helpers shared by the gfx950 kernels of this library (compress, pack,
// decompress): unaligned loads, wave-uniform values, lane reads, a cooperative byte copy, the DPP
// prefix sum, lane masks.  64-lane wavefronts only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snappy_kernels.h"

namespace szk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr uint32_t kMaxBlock = 65536u;

// ------------------------------------------------------------------------------- helpers

typedef u32x4 u32x4_u __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t ld32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);  // gfx950: one (possibly unaligned) global_load_dword
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ u32x4 ld128(const uint8_t *p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ u32x4 ld128_nt(const uint8_t *p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t l)
{
    return __builtin_amdgcn_readlane(v, uni(l));
}
__device__ __forceinline__ uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
// ballot of a per-lane condition (the HIP __ballot takes an int: the condition would go through a
// 0/1 vector register and a second compare)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// dword at byte offset `off` of a block of n bytes; bytes past the end read as 0 (never touched)
__device__ __forceinline__ uint32_t ld32_clamped(const uint8_t *in, uint32_t off, uint32_t n)
{
    if (off + 4 <= n)
        return ld32(in + off);
    uint32_t v = 0;
    if (off < n) v |= in[off];
    if (off + 1 < n) v |= (uint32_t)in[off + 1] << 8;
    if (off + 2 < n) v |= (uint32_t)in[off + 2] << 16;
    return v;
}

// Cooperative byte copy by one wave, dst-aligned 16-byte stores in the bulk; long copies stream
// with non-temporal loads/stores, kDepth KiB in flight (0: none; the depth costs 4 VGPRs per KiB).
template <int kDepth = 16>
__device__ __forceinline__ void wave_copy(uint8_t *__restrict__ d, const uint8_t *__restrict__ s,
                                          uint32_t len, uint32_t lane)
{
    if (len <= 64) {
        if (lane < len)
            d[lane] = s[lane];
        return;
    }
    uint32_t head = (16u - (uint32_t)((uintptr_t)d & 15u)) & 15u;
    if (lane < head)
        d[lane] = s[lane];
    d += head;
    s += head;
    len -= head;
    const uint32_t nvec = len >> 4;
    uint32_t i = lane;
    if (kDepth > 0 && nvec >= 64u * kDepth) {
        for (; i + 64u * (kDepth - 1) < nvec; i += 64u * kDepth) {
            u32x4 q[kDepth > 0 ? kDepth : 1];
#pragma unroll
            for (int u = 0; u < kDepth; u++)
                q[u] = ld128_nt(s + 16 * (size_t)(i + 64 * u));
#pragma unroll
            for (int u = 0; u < kDepth; u++)
                __builtin_nontemporal_store(q[u], reinterpret_cast<u32x4 *>(d + 16 * (size_t)(i + 64 * u)));
        }
    }
    for (; i + 192 < nvec; i += 256) {
        u32x4 q0 = ld128(s + 16 * (size_t)i);
        u32x4 q1 = ld128(s + 16 * (size_t)(i + 64));
        u32x4 q2 = ld128(s + 16 * (size_t)(i + 128));
        u32x4 q3 = ld128(s + 16 * (size_t)(i + 192));
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)i) = q0;
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)(i + 64)) = q1;
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)(i + 128)) = q2;
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)(i + 192)) = q3;
    }
    for (; i < nvec; i += 64)
        *reinterpret_cast<u32x4 *>(d + 16 * (size_t)i) = ld128(s + 16 * (size_t)i);
    const uint32_t done = nvec << 4;
    if (lane < (len & 15u))
        d[done + lane] = s[done + lane];
}

// Inclusive prefix sum over the 64 lanes with DPP (no LDS): Hillis-Steele inside each row of 16,
// then row_bcast:15 / row_bcast:31 to carry the row totals.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
#define SZ_DPP(v, ctrl, rmask) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, true)
    x += SZ_DPP(x, 0x111, 0xf);  // row_shr:1
    x += SZ_DPP(x, 0x112, 0xf);  // row_shr:2
    x += SZ_DPP(x, 0x114, 0xf);  // row_shr:4
    x += SZ_DPP(x, 0x118, 0xf);  // row_shr:8
    x += SZ_DPP(x, 0x142, 0xa);  // row_bcast:15 -> rows 1,3
    x += SZ_DPP(x, 0x143, 0xc);  // row_bcast:31 -> rows 2,3
#undef SZ_DPP
    return x;
}

__device__ __forceinline__ uint64_t below63(uint32_t k) { return (1ull << k) - 1; }  // bits 0..k-1, k <= 63
__device__ __forceinline__ uint64_t lanes_below(uint32_t k)  // bits 0..k-1, k in 0..64
{
    return k >= 64 ? ~0ull : ((1ull << k) - 1);
}

}  // namespace szk
