// fg_dev.h -- wave64 device primitives shared by the gfx950 FLAC kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef long long i64;

namespace fgdev {

// DPP controls (gfx9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND);
}

// Inclusive prefix sum over the 64 lanes (6 DPP adds, no LDS traffic).
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x)
{
    x += dpp0<0x111>(x);
    x += dpp0<0x112>(x);
    x += dpp0<0x114>(x);
    x += dpp0<0x118>(x);
    x += dpp0<0x142, 0xA, 0xF, false>(x);   // lane 15 of rows 0,2 -> rows 1,3
    x += dpp0<0x143, 0xC, 0xF, false>(x);   // lane 31 -> rows 2,3
    return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(x), 63);
}
__device__ __forceinline__ uint32_t wave_or32(uint32_t x)
{
    x |= dpp0<0x111>(x);
    x |= dpp0<0x112>(x);
    x |= dpp0<0x114>(x);
    x |= dpp0<0x118>(x);
    x |= dpp0<0x142, 0xA, 0xF, false>(x);
    x |= dpp0<0x143, 0xC, 0xF, false>(x);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_xor32(uint32_t x)
{
    x ^= dpp0<0x111>(x);
    x ^= dpp0<0x112>(x);
    x ^= dpp0<0x114>(x);
    x ^= dpp0<0x118>(x);
    x ^= dpp0<0x142, 0xA, 0xF, false>(x);
    x ^= dpp0<0x143, 0xC, 0xF, false>(x);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// 64-bit sum via three limb sums (each limb sum stays below 2^32)
__device__ __forceinline__ u64 wave_sum64(u64 v)
{
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const u64 a = wave_sum(lo & 0xFFFF), b = wave_sum(lo >> 16), c = wave_sum(hi & 0xFFFFFF), d = wave_sum(hi >> 24);
    return a + (b << 16) + (c << 32) + (d << 56);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint32_t ilog2_32(uint32_t v) { return 31u - (uint32_t)__clz(v); }
__device__ __forceinline__ uint32_t ilog2_64(u64 v) { return 63u - (uint32_t)__clzll(v); }

// LDS operations of one wave execute in issue order, so a cross-lane read-after-write through LDS needs
// no wait -- only a barrier against compiler reordering.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Exclusive prefix sum of one u64 per thread over a workgroup of up to 1024 threads (wave scans through shuffles, then the
// wave totals through LDS).  `wtot` holds one entry per wave (16); returns the prefix, *total gets the grand total.
__device__ __forceinline__ u64 block_scan_excl_u64(u64 v, u64 *wtot, u64 *total)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    u64 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d);
        if ((int)lane >= d) inc += ((u64)hi << 32) | lo;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    u64 base = 0, all = 0;
    for (uint32_t w = 0; w < nw; w++) { const u64 t = wtot[w]; if (w < wave) base += t; all += t; }
    *total = all;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
    // a*b mod x^16+x^15+x^2+1 (CRC-16 polynomial 0x8005)
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000) ? (((r << 1) ^ 0x8005) & 0xFFFF) : ((r << 1) & 0xFFFF);
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}

// CRC-16 steps without tables: x^16 + x^15 + x^2 + 1 is sparse enough for a closed form.  For a 16-bit u,
//     u * x^16 mod P = (u << 1 ^ u << 2) & 0xFFFF ^ parity(u) * 0x8003 ^ u[15] * 0x000A ^ u[14] * 0x8005
// (checked over all 65536 values); a byte v = (crc >> 8) ^ b takes crc to crc << 8 ^ parity(v) * 0x8003 ^ v << 1 ^ v << 2.
__device__ __forceinline__ uint32_t crc16_s(uint32_t u)          // u * x^16 mod P, u < 65536
{
    const uint32_t p = (uint32_t)__popc(u) & 1u;
    return (((u << 1) ^ (u << 2)) & 0xFFFFu) ^ ((0u - p) & 0x8003u) ^ ((0u - ((u >> 15) & 1u)) & 0x000Au) ^ ((0u - ((u >> 14) & 1u)) & 0x8005u);
}
// s * x^8192 mod P for a 16-bit s: x^8192 = x^8 + x^4 + x (0x0112), so the product has 24 bits and its top byte v folds back the
// way a byte of data does, v * x^16 = parity(v) * 0x8003 ^ v << 1 ^ v << 2 (checked against gf16_mul over all 65536 values):
// twelve instructions where the general multiplication takes sixty.
__device__ __forceinline__ uint32_t crc16_mul_x8192(uint32_t s)
{
    const uint32_t t = (s << 8) ^ (s << 4) ^ (s << 1), v = t >> 16;
    return (t & 0xFFFFu) ^ ((0u - ((uint32_t)__popc(v) & 1u)) & 0x8003u) ^ (v << 1) ^ (v << 2);
}
__device__ __forceinline__ uint32_t crc16_word(uint32_t c, uint32_t w) { return crc16_s(crc16_s(c ^ (w >> 16)) ^ (w & 0xFFFFu)); }
__device__ __forceinline__ uint32_t crc16_byte(uint32_t c, uint32_t b)
{
    const uint32_t v = ((c >> 8) ^ b) & 0xFFu;
    return ((c << 8) & 0xFFFFu) ^ ((0u - ((uint32_t)__popc(v) & 1u)) & 0x8003u) ^ (v << 1) ^ (v << 2);
}

}  // namespace fgdev
