// flac_enc_merge.hip -- frames of more than two channels from one-channel frames (gfx950).
//
// libFLAC codes the channels of a stream with three or more channels independently (stream_encoder.c process_subframes_:
// mid/side exists for stereo only), so a frame of C channels is C subframes that know nothing of each other, behind ONE frame
// header and in front of one CRC-16.  The encoder pipeline (flac_enc_pipe_impl.h) has a one-channel shape; fg_ctx.cpp runs it
// over C strided one-channel VIEWS of the interleaved PCM (FgBlockDesc.reserved = C), which yields C complete one-channel
// frames per block in a scratch stream, and the kernels here splice them:
//
//   fg_merge_sizes_kernel   thread = final frame: header length (from the one-channel frame's own header bytes), the exact bit
//                           length of every subframe (the pipeline's chunk bit counts, or what the generic kernel recorded),
//                           the final frame's size and its error flags
//   fg_merge_scan_kernel    one workgroup: exclusive scan of the sizes = the frames' byte offsets, total and error flags behind
//   fg_merge_frames_kernel  wave = final frame: the header of channel 0's frame with the channel-assignment field set to
//                           C - 1 and a fresh CRC-8, the C subframe bit strings moved up against each other (funnel shifts,
//                           lane = output word), zero padding, CRC-16.  No tables: for the polynomial x^16 + x^15 + x^2 + 1 a
//                           byte step has the closed form crc' = crc << 8 ^ parity(v) * 0x8003 ^ v << 1 ^ v << 2 with
//                           v = crc >> 8 ^ byte; lanes own interleaved words (state * x^2048 + crc(word)) and fold with
//                           x^(32 k) powers at the end.
//
// Reference path replaced: the frame assembly of FLAC__stream_encoder_process_interleaved for channels > 2 (frame header
// format.h:418-475, frame_header / subframe order stream_encoder.c process_frame_ -> process_subframes_ -> add_subframe_).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fg_dev.h"
#include "fg_types.h"

using namespace fgdev;

namespace {

__device__ __forceinline__ uint32_t mg_gf16_mul(uint32_t a, uint32_t b)        // a * b mod x^16 + x^15 + x^2 + 1
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000u) r ^= 0x18005u;
        if ((b >> i) & 1u) r ^= a;
    }
    return r & 0xFFFFu;
}
__device__ __forceinline__ uint32_t mg_crc16_byte(uint32_t crc, uint32_t byte)
{
    const uint32_t v = ((crc >> 8) ^ byte) & 0xFFu;
    return ((crc << 8) & 0xFFFFu) ^ ((__popc(v) & 1) ? 0x8003u : 0u) ^ (v << 1) ^ (v << 2);
}
__device__ __forceinline__ uint32_t mg_crc16_word(uint32_t w)                    // CRC of four bytes, MSB first, from state 0
{
    uint32_t c = 0;
    c = mg_crc16_byte(c, w >> 24); c = mg_crc16_byte(c, (w >> 16) & 0xFF); c = mg_crc16_byte(c, (w >> 8) & 0xFF); c = mg_crc16_byte(c, w & 0xFF);
    return c;
}
__device__ __forceinline__ uint32_t mg_xpow(uint32_t e)                          // x^e mod P
{
    uint32_t acc = 1, base = 2;
    while (e) {
        if (e & 1) acc = mg_gf16_mul(acc, base);
        base = mg_gf16_mul(base, base);
        e >>= 1;
    }
    return acc;
}
__device__ __forceinline__ uint32_t mg_crc8_byte(uint32_t crc, uint32_t byte)    // x^8 + x^2 + x + 1
{
    crc ^= byte;
#pragma unroll
    for (int b = 0; b < 8; b++) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xFFu : (crc << 1) & 0xFFu;
    return crc;
}

// length of the header of the one-channel frame at p (format.h:418-475): 4 bytes, the UTF-8 coded number, the optional block
// size and sample rate bytes, the CRC-8
__device__ __forceinline__ uint32_t mg_header_len(const uint8_t *p)
{
    const uint32_t b2 = p[2], u0 = p[4];
    uint32_t ul = 1;
    if (u0 & 0x80u) { ul = 2; if ((u0 & 0xE0u) == 0xE0u) ul = 3; if ((u0 & 0xF0u) == 0xF0u) ul = 4; if ((u0 & 0xF8u) == 0xF8u) ul = 5; if ((u0 & 0xFCu) == 0xFCu) ul = 6; if ((u0 & 0xFEu) == 0xFEu) ul = 7; }
    const uint32_t bc = b2 >> 4, rc = b2 & 15u;
    return 4 + ul + (bc == 6 ? 1u : bc == 7 ? 2u : 0u) + (rc == 12 ? 1u : (rc == 13 || rc == 14) ? 2u : 0u) + 1;
}

// bits of the one-channel frame in slot p in front of its padding
__device__ __forceinline__ uint32_t mg_payload_bits(const FgBlockResult *res, const uint32_t *chunk_bits, uint32_t p)
{
    if (res[p].reserved == 4 && chunk_bits) return chunk_bits[(size_t)p * 4] + chunk_bits[(size_t)p * 4 + 1] + chunk_bits[(size_t)p * 4 + 2] + chunk_bits[(size_t)p * 4 + 3];
    return res[p].best_bits[3];
}

__global__ void __launch_bounds__(256)
fg_merge_sizes_kernel(const FgBlockResult *res, const uint32_t *chunk_bits, const u64 *moffs, const uint8_t *mtmp, const uint32_t *fbase,
                      const uint32_t *fstr, uint32_t C, uint32_t nframes, uint32_t *sizes, uint32_t *errs)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nframes) return;
    const uint32_t p0 = fbase[f], str = fstr[f];
    const uint32_t hdr = mg_header_len(mtmp + moffs[p0]);
    uint32_t bits = 8 * hdr, e = 0;
    bool empty = false;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t p = p0 + c * str;
        const uint32_t t = mg_payload_bits(res, chunk_bits, p);
        e |= res[p].err;
        if (res[p].bytes == 0 || t < 8 * hdr) empty = true;
        bits += t - 8 * hdr;
    }
    sizes[f] = empty ? 0u : ((bits + 7) >> 3) + 2;
    errs[f] = e | (empty ? FG_ERR_REDO : 0u);          // (a one-channel frame that never came: reported, nothing written)
}

// offsets[f] = sum of the sizes in front of f; offsets[n] = total, offsets[n + 1] = OR of the error flags
__global__ void __launch_bounds__(1024)
fg_merge_scan_kernel(const uint32_t *sizes, const uint32_t *errs, uint32_t n, u64 *offsets)
{
    __shared__ u64 wsum[16];
    __shared__ uint32_t werr[16];
    __shared__ u64 carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    uint32_t eacc = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
        const uint32_t i = t0 + tid;
        const u64 v = i < n ? sizes[i] : 0;
        if (i < n) eacc |= errs[i];
        u64 x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u64 y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        u64 before = carry_s;
        for (uint32_t w = 0; w < wv; w++) before += wsum[w];
        if (i < n) offsets[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = before + x;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) eacc |= (uint32_t)__shfl_xor((int)eacc, o);
    if (lane == 0) werr[wv] = eacc;
    __syncthreads();
    if (tid == 0) {
        uint32_t e = 0;
        for (int w = 0; w < 16; w++) e |= werr[w];
        offsets[n] = carry_s;
        offsets[n + 1] = e;
    }
}

// 32 bits from absolute bit position `bit` of the big-endian byte string at the 4-byte aligned `w`
__device__ __forceinline__ uint32_t mg_read32(const uint32_t *w, u64 bit)
{
    const u64 wi = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t hi = __builtin_bswap32(w[wi]), lo = __builtin_bswap32(w[wi + 1]);
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

#define FG_MERGE_WPB 4
__global__ void __launch_bounds__(64 * FG_MERGE_WPB)
fg_merge_frames_kernel(const FgBlockResult *res, const uint32_t *chunk_bits, const u64 *moffs, const uint8_t *mtmp, const uint32_t *fbase,
                       const uint32_t *fstr, uint32_t C, uint32_t nframes, const uint32_t *sizes, const uint32_t *errs, const u64 *offsets,
                       uint8_t *dst, u64 dst_cap, FgBlockResult *fres, u64 *user_offsets)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t f = blockIdx.x * FG_MERGE_WPB + (threadIdx.x >> 6);
    if (f >= nframes) return;
    const uint32_t nb = sizes[f];
    const u64 off = offsets[f];
    if (lane == 0) {
        FgBlockResult r;
        r.bytes = nb; r.ca = 0; r.err = errs[f]; r.best_bits[0] = r.best_bits[1] = r.best_bits[2] = r.best_bits[3] = 0; r.reserved = 0;
        fres[f] = r;
        if (user_offsets) { user_offsets[f] = off; if (f == 0) user_offsets[nframes] = offsets[nframes]; }
    }
    if (nb == 0 || !dst || off + nb > dst_cap) return;        // (the host reports the short buffer once it has read the total)
    const uint32_t p0 = fbase[f], str = fstr[f];
    const uint8_t *h0 = mtmp + moffs[p0];
    const uint32_t hdr = mg_header_len(h0);
    // the subframes: where each starts in the scratch stream (absolute bit) and in the final frame, and how long it is
    u64 sb[8];
    uint32_t ds[8], ln[8];
    uint32_t at = 8 * hdr;
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        sb[c] = 0; ds[c] = at; ln[c] = 0;
        if (c < C) {
            const uint32_t p = p0 + c * str;
            sb[c] = moffs[p] * 8 + 8 * hdr;
            ln[c] = mg_payload_bits(res, chunk_bits, p) - 8 * hdr;
            at += ln[c];
        }
    }
    const uint32_t nbytes = (at + 7) >> 3;                      // the frame without its CRC-16 (= nb - 2)
    const uint32_t W = nbytes >> 2, tail = nbytes & 3;
    uint8_t *out = dst + off;
    // header bytes of the final frame (every lane keeps a copy: at most 16): channel assignment C - 1, fresh CRC-8
    uint32_t hb[16];
    {
        uint32_t c8 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            uint32_t b = k < hdr ? h0[k] : 0;
            if (k == 3) b = (b & 0x0Fu) | ((C - 1) << 4);
            if (k + 1 < hdr) c8 = mg_crc8_byte(c8, b);
            else if (k + 1 == hdr) b = c8;
            hb[k] = b;
        }
    }
    const uint32_t *mw = (const uint32_t *)mtmp;
    auto word = [&](uint32_t j) -> uint32_t {                   // bits [32 j, 32 j + 32) of the final frame
        uint32_t v = 0;
        const uint32_t b0 = j * 32;
        if (b0 < 8 * hdr) {
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t q = k >> 2;
                if (q == j && k < hdr) v |= hb[k] << (24 - 8 * (k & 3));
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            if (c < C && ln[c] != 0 && b0 + 32 > ds[c] && b0 < ds[c] + ln[c]) {
                uint32_t x;
                if (b0 >= ds[c]) {
                    const uint32_t rel = b0 - ds[c], valid = ln[c] - rel;
                    x = mg_read32(mw, sb[c] + rel);
                    if (valid < 32) x &= ~(0xFFFFFFFFu >> valid);
                }
                else {
                    const uint32_t lead = ds[c] - b0;              // 1 .. 31 bits of this word belong to what lies in front
                    const uint32_t valid = ln[c] < 32 - lead ? ln[c] : 32 - lead;
                    x = mg_read32(mw, sb[c]);
                    if (valid < 32) x &= ~(0xFFFFFFFFu >> valid);
                    x >>= lead;
                }
                v |= x;
            }
        }
        return v;
    };
    struct __attribute__((packed)) U32 { uint32_t v; };
    // lanes own interleaved words; in front of the first row as many zero words as make the count a multiple of 64 (no effect
    // on a CRC that starts at 0)
    const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
    const uint32_t x2048 = mg_xpow(2048);
    uint32_t s = 0;
    for (uint32_t t = 0; t < T; t++) {
        const int qi = (int)(t * 64 + lane) - (int)pad;
        uint32_t v = 0;
        if (qi >= 0) {
            v = word((uint32_t)qi);
            ((U32 *)out)[qi].v = __builtin_bswap32(v);
        }
        s = mg_gf16_mul(s, x2048) ^ mg_crc16_word(v);
    }
    s = mg_gf16_mul(s, mg_xpow(32 * (63 - lane)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s ^= (uint32_t)__shfl_xor((int)s, o);
    if (lane == 0) {
        uint32_t crc = s;
        if (tail) {
            const uint32_t v = word(W);
            for (uint32_t k = 0; k < tail; k++) { const uint32_t b = (v >> (24 - 8 * k)) & 0xFFu; out[W * 4 + k] = (uint8_t)b; crc = mg_crc16_byte(crc, b); }
        }
        out[nbytes] = (uint8_t)(crc >> 8);
        out[nbytes + 1] = (uint8_t)crc;
    }
}

}  // namespace

extern "C" int fg_launch_merge(const FgBlockResult *d_res, const uint32_t *d_chunk_bits, const unsigned long long *d_moffs, const uint8_t *d_mtmp,
                               const uint32_t *d_fbase, const uint32_t *d_fstr, uint32_t channels, uint32_t nframes, uint32_t *d_sizes,
                               uint32_t *d_errs, unsigned long long *d_offsets, uint8_t *d_dst, unsigned long long dst_cap, FgBlockResult *d_fres,
                               unsigned long long *d_user_offsets, hipStream_t stream)
{
    if (nframes == 0) return 0;
    hipLaunchKernelGGL(fg_merge_sizes_kernel, dim3((nframes + 255) / 256), dim3(256), 0, stream, d_res, d_chunk_bits, (const u64 *)d_moffs, d_mtmp,
                       d_fbase, d_fstr, channels, nframes, d_sizes, d_errs);
    hipLaunchKernelGGL(fg_merge_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t *)d_sizes, (const uint32_t *)d_errs, nframes, (u64 *)d_offsets);
    hipLaunchKernelGGL(fg_merge_frames_kernel, dim3((nframes + FG_MERGE_WPB - 1) / FG_MERGE_WPB), dim3(64 * FG_MERGE_WPB), 0, stream, d_res, d_chunk_bits,
                       (const u64 *)d_moffs, d_mtmp, d_fbase, d_fstr, channels, nframes, (const uint32_t *)d_sizes, (const uint32_t *)d_errs,
                       (const u64 *)d_offsets, d_dst, (u64)dst_cap, d_fres, (u64 *)d_user_offsets);
    return (int)hipGetLastError();
}
