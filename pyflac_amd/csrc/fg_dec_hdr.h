// fg_dec_hdr.h -- the fields of a FLAC frame header from its bytes (format.h:418-462): ONE copy of the rules for the passes that
// apply them -- the header pass (lane = frame, bytes from memory) and the index pass's resolve kernel, which leaves a packed
// record of every frame it settles so that the wave parser can start without waiting for the header pass (FgDecSelf).
#pragma once
#include <stdint.h>

namespace fgdev {

// one byte through the header CRC-8 (x^8 + x^2 + x + 1): u = c ^ byte, u x^8 mod P = the low byte of t = u (x^2 + x + 1) plus
// the two bits that spill past it times (x^2 + x + 1) again (checked against the bit loop for all 256 values)
__device__ __forceinline__ uint32_t crc8_step(uint32_t c, uint32_t byte)
{
    const uint32_t u = (c ^ byte) & 0xFFu;
    const uint32_t t = (u << 2) ^ (u << 1) ^ u, h = t >> 8;
    return (t & 0xFFu) ^ (h << 2) ^ (h << 1) ^ h;
}

struct HdrFields { uint32_t n, hdr_bytes, cac, bpc, extra; };

// p(i): byte i of the frame, i < 16 (a header has at most 16 bytes); len: bytes of the frame (the length rules of the format:
// a caller that does not know the length yet passes a large one and applies fg_hdr_len_ok later).  CRC8: also check the CRC-8.
// Returns 0 and the fields, or 1.
template <bool CRC8, typename Bytes>
__device__ __forceinline__ uint32_t fg_dec_header_fields(const Bytes &p, uint32_t len, HdrFields &h)
{
    h.n = 0; h.hdr_bytes = 0; h.cac = 0; h.bpc = 0; h.extra = 0;
    if (len < 7 || p(0) != 0xFF || (p(1) & 0xFE) != 0xF8) return 1;
    const uint32_t b2 = p(2), b3 = p(3);
    const uint32_t bsc = b2 >> 4, src = b2 & 15, cac = b3 >> 4, bpc = (b3 >> 1) & 7;
    if (bsc == 0 || src == 15 || cac > 10 || bpc == 3 || (b3 & 1)) return 1;
    uint32_t pos = 4;
    // UTF-8 coded frame / sample number
    const uint32_t x = p(pos++);
    uint32_t extra;
    if (!(x & 0x80)) extra = 0;
    else if ((x & 0xE0) == 0xC0) extra = 1;
    else if ((x & 0xF0) == 0xE0) extra = 2;
    else if ((x & 0xF8) == 0xF0) extra = 3;
    else if ((x & 0xFC) == 0xF8) extra = 4;
    else if ((x & 0xFE) == 0xFC) extra = 5;
    else if (x == 0xFE) extra = 6;
    else return 1;
    if (pos + extra + 4 > len) return 1;
    uint32_t bad = 0, n = 0;
    for (uint32_t i = 0; i < extra; i++) if ((p(pos++) & 0xC0) != 0x80) bad = 1;
    if (bad) return 1;
    switch (bsc) {
    case 1: n = 192; break;
    case 2: case 3: case 4: case 5: n = 576u << (bsc - 2); break;
    case 6: n = p(pos) + 1; pos += 1; break;
    case 7: n = ((p(pos) << 8) | p(pos + 1)) + 1; pos += 2; break;
    default: n = 256u << (bsc - 8); break;
    }
    if (src == 12) pos += 1; else if (src == 13 || src == 14) pos += 2;
    if (pos + 1 > len) return 1;
    if (CRC8) {
        uint32_t c8 = 0;
        for (uint32_t i = 0; i < pos; i++) c8 = crc8_step(c8, p(i));
        if (c8 != p(pos)) return 1;
    }
    pos++;
    h.n = n; h.hdr_bytes = pos; h.cac = cac; h.bpc = bpc; h.extra = extra;
    return 0;
}

// the length rules above for a header whose fields are known (extra = bytes of the coded number behind its first)
__device__ __forceinline__ bool fg_hdr_len_ok(uint32_t len, uint32_t extra, uint32_t hdr_bytes) { return len >= 7 && 9 + extra <= len && hdr_bytes <= len; }
__device__ __forceinline__ uint32_t fg_hdr_bps(uint32_t bpc, uint32_t si_bps)
{
    const uint32_t bp = bpc == 1 ? 8u : bpc == 2 ? 12u : bpc == 4 ? 16u : bpc == 5 ? 20u : bpc == 6 ? 24u : bpc == 7 ? 32u : 0u;
    return bpc ? bp : si_bps;
}
// the packed record: bits 0-15 n - 1, 16-20 header bytes, 21-24 channel assignment code, 25-27 sample size code, 28-30 extra, 31 valid
__device__ __forceinline__ uint32_t fg_hdr_pack(const HdrFields &h) { return 0x80000000u | ((h.n - 1) & 0xFFFFu) | (h.hdr_bytes << 16) | (h.cac << 21) | (h.bpc << 25) | (h.extra << 28); }

}  // namespace fgdev
