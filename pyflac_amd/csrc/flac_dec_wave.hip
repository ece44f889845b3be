// flac_dec_wave.hip -- wave-parallel FLAC subframe parser for gfx950: one frame per wavefront, 64 lanes on ONE Rice partition.
//
// The lane-serial parser (flac_dec_fast.hip: lane = frame) needs as many frames as the chip has lanes; a single stream has a
// few thousand, so its parse ran on one wave per CU with under half the lanes alive and took what 8192 serial codes take.
// Here the serial dependency of a Rice-coded partition -- a code starts where the previous one ends -- is broken the way
// self-synchronising variable-length codes allow:
//
//   * the bits of the partition are cut into chunks of B = 16 .. 128 bits, one per lane (a batch = 64 B bits; B follows the
//     Rice parameter and what is left of the partition); a code belongs to the chunk its first bit lies in.  Every lane
//     keeps its chunk and the 95 bits behind it as a private row of words in LDS (odd row stride: no bank conflicts);
//   * every lane walks its chunk from a GUESSED entry (offset 0): per code one two-word LDS read, v_alignbit, v_ffbh and
//     v_add3 (count leading zeros = quotient, + k + 1 = length).  It keeps the number of codes and the exit offset, i.e.
//     where the first code of the next chunk starts;
//   * sync rounds: a lane whose entry differs from its left neighbour's exit walks again from there.  Lane 0 is right
//     from the start, so after round r lanes 0..r are final; because a wrong walk falls into step with the right one as
//     soon as both hit the same stop bit (about one code in five), a handful of rounds settle all 64 lanes.  The loop
//     ends when no entry changed -- or when every lane in front of the first changed one already covers what is left of
//     the partition;
//   * one DPP prefix sum over the counts gives every lane the sample index of its first code; a last walk extracts
//     quotient and remainder, undoes the zig-zag and puts the residuals into an LDS buffer by sample index, which the
//     wave then writes to the plane with coalesced stores;
//   * the lane that owns the partition's last code knows where the next partition starts.
//
// A code whose unary part exceeds the 32-bit window (a residual above 32 * 2^k) stops the batch in front of it, is read bit
// by bit, and the batches resume behind it.  Escape-coded partitions, VERBATIM and CONSTANT subframes and the warm-up /
// coefficient fields are fixed-width fields at computable positions: lane = field.
//
// Output: the residual plane (frame-planar int32, warm-up samples in place), the FgDecSub records and the parse status,
// exactly as fg_dec_rice_kernel leaves them for fg_dec_restore_kernel.  Frames outside the restore kernel's envelope
// (predictor order > 12, 33-bit subframes, ...) get status 3 and go to the generic decoder as before.
//
// Reference path replaced: read_subframe_*, read_residual_partitioned_rice_ inside libFLAC (SURVEY.md section 8a row D2,
// Appendix B; format: /root/reference/pyflac/include/FLAC/format.h:191-396).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "fg_dev.h"
#include "fg_types.h"
#include "fg_dec_hdr.h"

using namespace fgdev;

#define FG_DMAXO 12
#define FG_DEC_RPARAMS 256

extern "C" int fg_func_set_lds(const void *fn, size_t bytes);   // fg_ctx.cpp: per device, thread-safe

namespace {

// Where the bits of one frame live: big-endian 32-bit words from the aligned word that holds the frame's first byte.
struct WRd {
    const uint32_t *fw;
    uint32_t lw;          // last loadable word, relative to fw (the word that holds the last stream byte)
    uint32_t bit0;        // bit offset of the frame start inside fw[0]: 0, 8, 16 or 24
};

__device__ __forceinline__ uint32_t wr_word(const WRd &r, uint32_t w) { return __builtin_bswap32(r.fw[w < r.lw ? w : r.lw]); }

// 32 bits from frame-relative bit position `pos` (per lane, or uniform)
__device__ __forceinline__ uint32_t wr_peek(const WRd &r, uint32_t pos)
{
    const uint32_t gp = r.bit0 + pos, w = gp >> 5, o = gp & 31;
    const uint32_t hi = wr_word(r, w), lo = wr_word(r, w + 1);
    return o ? __builtin_amdgcn_alignbit(hi, lo, 32 - o) : hi;
}
__device__ __forceinline__ uint32_t wr_bits(const WRd &r, uint32_t pos, uint32_t n)        // n <= 32
{
    return n ? wr_peek(r, pos) >> (32 - n) : 0;
}
__device__ __forceinline__ int32_t wr_sbits(const WRd &r, uint32_t pos, uint32_t n)
{
    return n ? (int32_t)wr_peek(r, pos) >> (32 - n) : 0;
}

__device__ __forceinline__ int32_t unzig(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }

__device__ __forceinline__ uint32_t wave_min32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o); v = t < v ? t : v; }
    return v;
}

// A lane's row in LDS: the stream from the word that holds its chunk's first bit, every word with its bits reversed (stream
// bit b of a word = bit b counted from the LSB).  The 32 bits that start q bits into the row are then
// alignbit(row[(q >> 5) + 1], row[q >> 5], q) -- v_alignbit shifts right by the low five bits of q, no special case at word
// boundaries --, a code's leading zeros are the window's TRAILING zeros (v_ffbl), and the remainder comes out bit-reversed.
__device__ __forceinline__ uint32_t row_word(uint32_t raw) { return __builtin_bitreverse32(__builtin_bswap32(raw)); }
__device__ __forceinline__ uint32_t row_win(const uint32_t *row, uint32_t q)
{
    const uint32_t *w = row + (q >> 5);
    return __builtin_amdgcn_alignbit(w[1], w[0], q);
}

// Walk the chunk [q, hi) from q: number of codes that start inside it, and how far behind its end the next code starts.
// Seven VALU instructions and one LDS read a code; lanes leave the loop through EXEC as they pass `hi`.  A window without a
// stop bit makes v_ffbl return -1: the walk then moves on by k bits -- wrong, but it moves on (k >= 1), and the output walk
// reports the code.  With k = 0 that would stand still, so that case keeps a v_min_u32 (SAFE).
#define WP_WALK_ASM(GUARD)                                                  \
    asm volatile("s_mov_b64 %[sv], exec\n"                                  \
                 "v_cmp_lt_u32 vcc, %[q], %[hi]\n"                          \
                 "s_and_b64 exec, exec, vcc\n"                              \
                 "s_cbranch_execz 2f\n"                                     \
                 "1:\n"                                                     \
                 "v_lshrrev_b32 %[t], 5, %[q]\n"                            \
                 "v_lshl_add_u32 %[t], %[t], 2, %[ra]\n"                    \
                 "ds_read2_b32 v[62:63], %[t] offset1:1\n"                  \
                 "v_add_u32 %[n], 1, %[n]\n"                                \
                 "s_waitcnt lgkmcnt(0)\n"                                   \
                 "v_alignbit_b32 %[t], v63, v62, %[q]\n"                    \
                 "v_ffbl_b32 %[t], %[t]\n" GUARD                            \
                 "v_add3_u32 %[q], %[q], %[kp1], %[t]\n"                    \
                 "v_cmp_lt_u32 vcc, %[q], %[hi]\n"                          \
                 "s_and_b64 exec, exec, vcc\n"                              \
                 "s_cbranch_execnz 1b\n"                                    \
                 "2:\n"                                                     \
                 "s_mov_b64 exec, %[sv]\n"                                  \
                 : [q] "+v"(q), [n] "+v"(n), [t] "=&v"(t), [sv] "=&s"(sv)   \
                 : [hi] "v"(hi), [kp1] "v"(kp1), [ra] "v"(ra)               \
                 : "vcc", "v62", "v63", "memory")
template <bool SAFE>
__device__ __forceinline__ void row_walk(const uint32_t *row, uint32_t q, uint32_t hi, uint32_t kp1, uint32_t &cnt, uint32_t &exitq)
{
    uint32_t n = 0, t;
    unsigned long long sv;
    const uint32_t ra = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)row;
    if (SAFE) WP_WALK_ASM("v_min_u32 %[t], 32, %[t]\n");
    else WP_WALK_ASM("");
    cnt = n;
    exitq = q - hi;
}

// lane i <- lane i - 1 (lane 0 <- 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }

#define WP_ROWS_W (64 * 9)      // 64 rows of at most 9 words (B = 192: 6 + 3)
#define WP_OUT_W 704            // words: the residuals of one batch, 704 of 32 bits (11 a lane) or 1408 of 16 bits (22 a lane)
// (5120 bytes of LDS a wave: eight workgroups of four waves fill a CU's 160 KB, so that all frames of a ten-minute stream --
// 7032 -- are resident at once; with fewer slots than frames the kernel takes two wave lifetimes instead of one)
#define WP_WAVE_W (WP_ROWS_W + WP_OUT_W)
typedef uint32_t wp_u32x4 __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte accesses at 4-byte alignment
typedef int32_t wp_i32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t wp_u32x4h __attribute__((ext_vector_type(4), aligned(2)));      // ... at 2-byte alignment (global memory only)

// P16 (streams of up to 16 bits, round 4): the residual plane is 16 bits wide.  The plane of a subframe keeps its place -- byte
// offset 4 (out_off C + ch n) of the scratch area -- and uses the first half of its space: the parser writes and the restore
// kernel reads half the bytes (0.23 GB less traffic per launch of the headline stream).  A frame with a value that does not fit
// -- a side channel's warm-up sample beyond 16 bits, a residual of a predictor gone wild, wide escape codes, Rice parameters of
// 13 and more with large quotients -- ends with status 6, and the host repeats the call with 32-bit planes (decode_frames_impl;
// it remembers the outcome for the stream).  No second attempt inside the kernel: the state it needs to start a subframe again
// cost this kernel, which lives on exactly 64 registers, forty spilled ones.
template <bool WIDE, bool P16>
__global__ void __launch_bounds__(256, 8)
fg_dec_wparse_kernel(const uint8_t *stream, u64 stream_len, const FgDecFrame *frames, uint32_t nframes, int32_t *scratch,
                     FgDecSub *subs, FgDecResult *results, uint16_t *rparams, unsigned long long *counters, FgDecSelf SF)
{
    __shared__ uint32_t lds[4 * WP_WAVE_W];
    const int lane = threadIdx.x & 63;
    uint32_t *const rows = lds + (threadIdx.x >> 6) * WP_WAVE_W;
    int32_t *const outb = (int32_t *)(rows + WP_ROWS_W);
    const uint32_t f = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (f >= nframes) return;
    FgDecFrame fr;
    u64 plane_el;                                       // first element (int32) of the frame's part of the plane
    if (SF.offsets) {
        // ---- on its own (FgDecSelf): the frame from its offsets and its header record, its part of the plane by its number
        const u64 o0 = SF.offsets[f], o1 = SF.offsets[f + 1];
        const bool inside = o0 < stream_len && o1 <= stream_len && o1 > o0 && o1 - o0 < 0x7FFFFFFFull;
        const uint32_t len = inside ? (uint32_t)(o1 - o0) : 0;
        const uint32_t rec = rfl(SF.hdrrec[f]);
        // stride: the largest block size among the first 64 records
        uint32_t nmax = 0;
        { const uint32_t r = (uint32_t)lane < nframes ? SF.hdrrec[lane] : 0u; nmax = (r >> 31) ? (r & 0xFFFFu) + 1u : 0u; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o); nmax = t > nmax ? t : nmax; }
        nmax = rfl(nmax);
        const uint32_t hn = (rec & 0xFFFFu) + 1u, hb = (rec >> 16) & 31u, cac = (rec >> 21) & 15u, bpc = (rec >> 25) & 7u, extra = (rec >> 28) & 7u;
        uint32_t st0 = 0;                               // 1: not a frame (what the header pass says too), 3: the generic decoder's
        if (!inside) st0 = 1;
        else if (!(rec >> 31)) st0 = 3;                 // (a contested slot: the header pass decides)
        else if (!fg_hdr_len_ok(len, extra, hb)) st0 = 1;
        fr.byte_off = inside ? o0 : 0; fr.out_off = 0; fr.bytes = len; fr.n = hn; fr.hdr_bytes = hb;
        fr.bps = fg_hdr_bps(bpc, SF.si_bps);
        if (cac < 8) { fr.channels = cac + 1; fr.ca = 0; } else { fr.channels = 2; fr.ca = cac - 7; }
        const u64 stride = ((u64)nmax * fr.channels * 4u + 15ull) & ~15ull;
        const u64 slot = (u64)f * stride;
        if (!st0 && (hn > nmax || slot + stride > SF.plane_cap_bytes)) st0 = 3;
        if (lane == 0) SF.planeoff[f] = st0 ? 0ull : slot;
        if (st0) { if (lane == 0) results[f].err = st0; return; }
        plane_el = slot >> 2;
    }
    else {
        fr = frames[f];
        if (fr.bytes == 0) return;                      // rejected by the header pass
        plane_el = fr.out_off * fr.channels;
    }
    uint32_t err = 0;
    if (fr.bytes < fr.hdr_bytes + 2) err = 1;
    const uint32_t n = fr.n, C = fr.channels;
    const uint32_t end_bits = err ? 0 : (fr.bytes - 2) * 8;

    WRd rd;
    {
        const uintptr_t sa = (uintptr_t)stream;
        const u64 mis = (u64)(sa & 3);
        const u64 fb = mis + fr.byte_off;
        const u64 lastw = (mis + stream_len - 1) >> 2, w0 = fb >> 2;
        rd.fw = (const uint32_t *)(sa & ~(uintptr_t)3) + w0;
        const u64 room = lastw > w0 ? lastw - w0 : 0;
        rd.lw = room > 0x07FFFFF0ull ? 0x07FFFFF0u : (uint32_t)room;
        rd.bit0 = (uint32_t)(fb & 3) * 8;
    }
    uint32_t pos = fr.hdr_bytes * 8;                    // frame-relative bit position (uniform)
    uint32_t n_batches = 0, n_rounds = 0, n_hard = 0;   // (counters: tuning aid)

    for (uint32_t ch = 0; ch < C && !err; ch++) {
      {
        FgDecSub *sd = &subs[(size_t)f * C + ch];
        int32_t *pl = scratch + plane_el + (u64)ch * n;
        int16_t *const pl16 = (int16_t *)pl;
        constexpr bool nar = P16;
        uint32_t ovf = 0;                    // a value of this subframe does not fit 16 bits
        auto fits = [](int32_t v) -> uint32_t { return v != (int32_t)(int16_t)v ? 1u : 0u; };
        uint32_t sb = fr.bps;
        if ((fr.ca == 1 && ch == 1) || (fr.ca == 2 && ch == 0) || (fr.ca == 3 && ch == 1)) sb++;
        const uint32_t hdr = wr_bits(rd, pos, 8);
        pos += 8;
        uint32_t wasted = 0;
        if (hdr & 0x80) err = 1;
        if (!err && (hdr & 1)) {
            // unary-coded wasted bits - 1
            uint32_t z = 0;
            for (;;) {
                const uint32_t p = wr_peek(rd, pos);
                if (p) { const uint32_t l = (uint32_t)__builtin_clz(p); z += l; pos += l + 1; break; }
                z += 32; pos += 32;
                if (pos > end_bits) break;
            }
            wasted = z + 1;
            if (wasted >= sb) err = 1; else sb -= wasted;
        }
        if (!err && sb > (WIDE ? 32u : 24u)) err = 3;
        const uint32_t t = (hdr >> 1) & 0x3F;
        uint32_t mode = 0, order = 0;
        if (t == 0) mode = 0;
        else if (t == 1) mode = 1;
        else if (t >= 8 && t <= 12) { mode = 2; order = t & 7; }
        else if (t >= 32) { mode = 2; order = (t & 31) + 1; }
        else if (!err) err = 1;
        if (!err && order > n) err = 1;
        if (!err && order > FG_DMAXO) err = 3;
        if (err) break;

        int shift = 0;
        uint32_t sprec = 0, po = 0, plen = 4;
        int32_t cval = 0;
        if (mode == 0) {
            cval = wr_sbits(rd, pos, sb);
            pos += sb;
            if (nar) {
                ovf |= fits(cval);
                const uint32_t two = ((uint32_t)cval & 0xFFFFu) * 0x10001u;
                for (uint32_t i = lane; i < n / 2; i += 64) ((uint32_t *)pl16)[i] = two;
                if ((n & 1) && lane == 0) pl16[n - 1] = (int16_t)cval;
            }
            else for (uint32_t i = lane; i < n; i += 64) pl[i] = cval;
        }
        else if (mode == 1) {
            if (nar) {
                for (uint32_t i = lane; i < (n + 1) / 2; i += 64) {
                    const int32_t a = wr_sbits(rd, pos + 2 * i * sb, sb), b = 2 * i + 1 < n ? wr_sbits(rd, pos + (2 * i + 1) * sb, sb) : 0;
                    ovf |= fits(a) | fits(b);
                    if (2 * i + 1 < n) ((uint32_t *)pl16)[i] = ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
                    else pl16[2 * i] = (int16_t)a;
                }
            }
            else for (uint32_t i = lane; i < n; i += 64) pl[i] = wr_sbits(rd, pos + i * sb, sb);
            pos += n * sb;
        }
        else {
            // warm-up samples: lane = sample
            if ((uint32_t)lane < order) {
                const int32_t wv_ = wr_sbits(rd, pos + (uint32_t)lane * sb, sb);
                if (nar) { pl16[lane] = (int16_t)wv_; ovf |= fits(wv_); }
                else pl[lane] = wv_;
            }
            pos += order * sb;
            if (t >= 32) {
                const uint32_t ps = wr_bits(rd, pos, 9);
                pos += 9;
                const uint32_t prec = (ps >> 5) + 1;
                sprec = prec;
                if (prec == 16) err = 1;
                shift = (int32_t)(ps << 27) >> 27;
                if (!err && shift < 0) err = 1;
                // the 32-bit restore is exact only under libFLAC's own width rule (lpc.c: bps + precision + ilog2(order) <= 32)
                if (!err && !WIDE && sb + prec + ilog2_32(order) > 32) err = 3;
                if (err) break;
                if ((uint32_t)lane < order) sd->q[lane] = wr_sbits(rd, pos + (uint32_t)lane * prec, prec);
                pos += order * prec;
            }
            else if (lane == 0) {
                // fixed predictor of order k as FIR with binomial coefficients
                const int32_t c0 = (int32_t)order, c1 = order < 2 ? 0 : (order == 2 ? -1 : order == 3 ? -3 : -6);
                const int32_t c2 = order < 3 ? 0 : (order == 3 ? 1 : 4), c3 = order < 4 ? 0 : -1;
                sd->q[0] = c0; sd->q[1] = c1; sd->q[2] = c2; sd->q[3] = c3;
            }
            const uint32_t mp = wr_bits(rd, pos, 6);
            pos += 6;
            const uint32_t method = mp >> 4;
            if (method > 1) { err = 1; break; }
            po = mp & 15;
            plen = method ? 5 : 4;
            const uint32_t escv = method ? 31 : 15;
            const uint32_t psz = n >> po;
            if ((po > 0 && ((n & ((1u << po) - 1)) != 0 || psz < order)) || (po == 0 && n < order)) { err = 1; break; }

            uint32_t si = order;                         // next sample of the subframe
            for (uint32_t part = 0; part < (1u << po) && !err; part++) {
                const uint32_t pend = po == 0 ? n : (part + 1) * psz;
                uint32_t R = pend - si;                  // codes left in this partition
                const uint32_t k = wr_bits(rd, pos, plen);
                pos += plen;
                if (k == escv) {
                    const uint32_t raw = wr_bits(rd, pos, 5);
                    pos += 5;
                    if (rparams && part < FG_DEC_RPARAMS && lane == 0) rparams[((size_t)f * C + ch) * FG_DEC_RPARAMS + part] = (uint16_t)(0x8000u | (raw << 8));
                    for (uint32_t i = lane; i < R; i += 64) {
                        const int32_t ev = wr_sbits(rd, pos + i * raw, raw);
                        if (nar) { pl16[si + i] = (int16_t)ev; ovf |= fits(ev); }
                        else pl[si + i] = ev;
                    }
                    pos += R * raw;
                    si = pend;
                    if (pos > end_bits) err = 4;
                    continue;
                }
                if (rparams && part < FG_DEC_RPARAMS && lane == 0) rparams[((size_t)f * C + ch) * FG_DEC_RPARAMS + part] = (uint16_t)k;
                const uint32_t kp1 = k + 1;
                // Residuals that fit 16 bits go through the output buffer as such (twice the codes per batch, so chunks of up to 256
                // bits and fewer sync rounds): with k <= 12 a value outgrows 16 bits only through a quotient of 2^(16 - k) or more,
                // and those codes take the bit-by-bit path that codes of 32 and more leading zeros take anyway.
                // (16-bit plane: every batch goes through the buffer as 16-bit values)
                // (with k > 12 a value fits 16 bits while its quotient stays below 2^(16 - k): the test that sends long codes to the bit-by-bit
                // path finds the others, and such a subframe takes the 32-bit form)
                const bool o16 = nar || k <= 12;
                const uint32_t hardlz = k < 12 ? 32u : ((nar || k == 12) ? (k >= 16 ? 0u : 1u << (16 - k)) : 32u);
                const uint32_t outcap = o16 ? 2u * WP_OUT_W : WP_OUT_W;
                while (R > 0) {
                    n_batches++;
                    // ---- chunk size: at most outcap / 64 codes per lane, and no wider than the rest of the partition needs
                    const uint32_t Rb = R < outcap ? R : outcap;               // codes this batch may deliver
                    uint32_t B = (outcap >> 6) * kp1;                           // (a multiple of 32, or 16)
                    B = B >= 192 ? 192u : (B >= 32 ? (B & ~31u) : 16u);
                    { const uint32_t est = Rb * (k + 2) + 32; while (B > 16 && 32 * B >= est) B = B > 32 ? ((B >> 1) + 31) & ~31u : 16u; }
                    const uint32_t nw = ((B + 62) >> 5) + 2, rs = nw | 1;      // 4 .. 9 words; odd row stride
                    const bool maybe_end = Rb * kp1 < 64 * B;                  // (else the batch cannot hold what is left)
                    // ---- this lane's row: the words from the one that holds its chunk's first bit
                    const uint32_t a0 = rd.bit0 + pos + B * (uint32_t)lane, wr0 = a0 >> 5, q0 = a0 & 31, hi = q0 + B;
                    uint32_t *const row = rows + rs * (uint32_t)lane;
                    {
                        const uint32_t wlast = ((rd.bit0 + pos + B * 63u) >> 5) + 8;
                        if (wlast <= rd.lw) {
                            const wp_u32x4 v0 = *(const wp_u32x4 *)(rd.fw + wr0);
                            row[0] = row_word(v0.x); row[1] = row_word(v0.y); row[2] = row_word(v0.z); row[3] = row_word(v0.w);
                            if (nw > 4) {
                                const wp_u32x4 v1 = *(const wp_u32x4 *)(rd.fw + wr0 + 4);
                                row[4] = row_word(v1.x);
                                if (nw > 5) row[5] = row_word(v1.y);
                                if (nw > 6) row[6] = row_word(v1.z);
                                if (nw > 7) row[7] = row_word(v1.w);
                                if (nw > 8) row[8] = row_word(rd.fw[wr0 + 8]);
                            }
                        }
                        else {
#pragma unroll
                            for (uint32_t t = 0; t < 9; t++) { const uint32_t w = wr0 + t; if (t < nw) row[t] = row_word(rd.fw[w < rd.lw ? w : rd.lw]); }
                        }
                    }
                    // ---- guess, then sync rounds
                    uint32_t e = 0, cnt, xq;
                    if (k == 0) row_walk<true>(row, q0, hi, kp1, cnt, xq); else row_walk<false>(row, q0, hi, kp1, cnt, xq);
                    for (uint32_t round = 0; round < 64; round++) {
                        const uint32_t px = wave_shr1(xq);
                        const bool changed = px != e;
                        const u64 chm = __ballot(changed);
                        if (chm == 0) break;
                        if (maybe_end) {
                            // lanes in front of the first changed one are final: if they cover the rest of the partition, stop.  (A lane
                            // holds at most ceil(B / (k + 1)) codes: the prefix sum is only worth its fourteen instructions when that
                            // many codes a lane in front of the first changed one could reach what is left)
                            const uint32_t first = (uint32_t)__builtin_ctzll(chm);
                            if (first * (B + k) >= Rb * kp1) {
                                const uint32_t pf = wave_scan_add(cnt) - cnt;
                                if (rl(pf, (int)first) >= Rb) break;
                            }
                        }
                        n_rounds++;
                        if (changed) {
                            e = px;
                            if (k == 0) row_walk<true>(row, q0 + e, hi, kp1, cnt, xq); else row_walk<false>(row, q0 + e, hi, kp1, cnt, xq);
                        }
                    }
                    const uint32_t incl = wave_scan_add(cnt);
                    const uint32_t pfx = incl - cnt;
                    const uint32_t total = rl(incl, 63);
                    // ---- output walk: residuals into the LDS buffer by sample index.  Twenty instructions a code: the window from the stop
                    // bit on (bit 0 = stop bit, bits 1..k = the remainder, first bit lowest) reversed puts the remainder at bits
                    // 31-k..30 in order -- one v_bfe_u32 --; codes past what the batch delivers (lanes behind the partition's end,
                    // counts inflated by zero runs) are not walked at all -- a lane stops behind code Rb - 1, which also tells where
                    // the next partition starts --; of the long codes only the longest run of zeros is tracked here (one v_max_u32), the lane that
                    // met one walks again below to say which code it was.
                    uint32_t hardidx = 0xFFFFFFFFu, hardq = 0, endq_out = 0;
                    {
                        uint32_t q = q0 + e, lzmax = 0;
                        const uint32_t esz = o16 ? 2u : 4u;
                        const uint32_t ob = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int32_t *)outb;
                        uint32_t at = ob + pfx * esz;
                        const uint32_t atend = ob + Rb * esz;       // (a lane stops behind code Rb - 1: its q is then where the next partition starts)
                        const uint32_t kw = 31u - k;
                        while (q < hi && at < atend) {
                            const uint32_t win = row_win(row, q);
                            const uint32_t lz = win ? (uint32_t)__builtin_ctz(win) : 32u;
                            const uint32_t adv = lz + kp1;
                            uint32_t t = win >> (lz & 31);
                            if (adv > 32) t = row_win(row, q + (lz < 32 ? lz : 31u));          // ... when the remainder does not lie in the window
                            const uint32_t rem = __builtin_amdgcn_ubfe(__builtin_bitreverse32(t), kw, k);
                            const int32_t val = unzig((lz << k) | rem);
                            lzmax = lz > lzmax ? lz : lzmax;
                            if (o16) *(__attribute__((address_space(3))) int16_t *)(uintptr_t)at = (int16_t)val;
                            else *(__attribute__((address_space(3))) int32_t *)(uintptr_t)at = val;
                            q += adv;
                            at += esz;
                        }
                        endq_out = q - q0;
                        if (__any(lzmax >= hardlz)) {
                            // (rare: a code of 32 and more zeros, or one whose value outgrows the 16-bit buffer) which code, and where
                            uint32_t q2 = q0 + e, idx = pfx;
                            while (q2 < hi) {
                                const uint32_t win = row_win(row, q2);
                                const uint32_t lz = win ? (uint32_t)__builtin_ctz(win) : 32u;
                                if (lz >= hardlz && hardidx == 0xFFFFFFFFu) { hardidx = idx; hardq = q2 - q0; }
                                q2 += lz + kp1;
                                idx++;
                            }
                        }
                    }
                    const uint32_t lim = total < Rb ? total : Rb;
                    uint32_t hmin = 0xFFFFFFFFu;
                    if (__any(hardidx < lim)) hmin = wave_min32(hardidx);
                    // ---- the buffer goes to the plane: four samples a lane and store
                    {
                        const uint32_t nout = hmin < lim ? hmin : lim;
                        int32_t *const dst = pl + si;
                        wave_lds_fence();
                        if (nar) {
                            // the buffer's halfwords are the plane's: eight samples a lane and store.  A batch may start at an odd sample
                            // (an odd predictor order, a code read bit by bit in front of it): the stores are then 2-byte aligned 16-byte
                            // stores, which global memory takes (the compiler emits them as such: unaligned access mode)
                            int16_t *const d16 = pl16 + si;
                            const int16_t *const oh = (const int16_t *)outb;
                            for (uint32_t j = 8u * (uint32_t)lane; j < nout; j += 512) {
                                if (j + 8 <= nout) *(wp_u32x4h *)(d16 + j) = *(const wp_u32x4 *)(oh + j);
                                else for (uint32_t e = j; e < nout; e++) d16[e] = oh[e];
                            }
                        }
                        else if (o16) {
                            const uint2 *const o2 = (const uint2 *)outb;
                            for (uint32_t j = 4u * (uint32_t)lane; j < nout; j += 256) {
                                const uint2 w = o2[j >> 2];
                                const int32_t v0 = (int32_t)(w.x << 16) >> 16, v1 = (int32_t)w.x >> 16, v2 = (int32_t)(w.y << 16) >> 16, v3 = (int32_t)w.y >> 16;
                                if (j + 4 <= nout) { wp_i32x4 v; v.x = v0; v.y = v1; v.z = v2; v.w = v3; *(wp_i32x4 *)(dst + j) = v; }
                                else { dst[j] = v0; if (j + 1 < nout) dst[j + 1] = v1; if (j + 2 < nout) dst[j + 2] = v2; }
                            }
                        }
                        else {
                            const int4 *const o4 = (const int4 *)outb;
                            for (uint32_t j = 4u * (uint32_t)lane; j < nout; j += 256) {
                                const int4 w = o4[j >> 2];
                                if (j + 4 <= nout) { wp_i32x4 v; v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w; *(wp_i32x4 *)(dst + j) = v; }
                                else { dst[j] = w.x; if (j + 1 < nout) dst[j + 1] = w.y; if (j + 2 < nout) dst[j + 2] = w.z; }
                            }
                        }
                    }
                    if (hmin < lim) {
                        if (nar && k > 12) { err = 6; break; }       // (a value beyond 16 bits: the call is repeated with 32-bit planes)
                        // a code with too many leading zeros for the window (or the 16-bit buffer): everything in front of it stands;
                        // read it bit by bit, resume behind it
                        n_hard++;
                        const u64 own = __ballot(hardidx == hmin);
                        const int L = (int)__builtin_ctzll(own);
                        uint32_t hp = pos + B * (uint32_t)L + rl(hardq, L);
                        uint32_t z = 0;
                        bool found = false;
                        while (hp <= end_bits) {
                            const uint32_t p = wr_peek(rd, hp);
                            if (p) { const uint32_t l = (uint32_t)__builtin_clz(p); z += l; hp += l + 1; found = true; break; }
                            z += 32; hp += 32;
                        }
                        if (!found) { err = 4; break; }
                        const uint32_t rem = wr_bits(rd, hp, k);
                        hp += k;
                        {
                            const int32_t hv = unzig((z << k) | rem);
                            if (nar) { if (lane == 0) pl16[si + hmin] = (int16_t)hv; ovf |= fits(hv); }
                            else if (lane == 0) pl[si + hmin] = hv;
                        }
                        si += hmin + 1; R -= hmin + 1; pos = hp;
                    }
                    else if (total >= Rb) {
                        // the batch reaches the end of the partition (or fills the buffer): the lane that owns code Rb - 1 stopped its
                        // output walk behind it and says where the next code starts
                        const bool own = pfx < Rb && Rb <= pfx + cnt;
                        const int L = (int)__builtin_ctzll(__ballot(own));
                        pos = pos + B * (uint32_t)L + rl(endq_out, L);
                        si += Rb; R -= Rb;
                    }
                    else {
                        pos += B * 64u + rl(xq, 63);
                        si += total; R -= total;
                    }
                    if (pos > end_bits) { err = 4; break; }
                }
            }
            if (err) break;
        }
        // what FLAC__Frame.subframes[] reports (fg_types.h FgDecSub): type, coefficient precision, partition order, method
        if (nar && !err && __any(ovf != 0)) { err = 6; break; }      // (the call is repeated with 32-bit planes)
        if (lane == 0) {
            const uint32_t stype = mode == 0 ? 0u : mode == 1 ? 1u : (t >= 32 ? 3u : 2u);
            sd->order = order; sd->shift = shift; sd->wasted = wasted;
            sd->flags = stype | (sprec << 2) | (po << 7) | ((plen == 5 ? 1u : 0u) << 11) | (1u << 12);
            if (mode == 0) sd->q[0] = cval;
        }
        if (pos > end_bits) err = 4;
      }
    }
    if (!err) {
        const uint32_t endb = (pos + 7) & ~7u;
        const uint32_t padb = endb - pos;
        if (endb != end_bits) err = 4;
        else if (padb && (wr_peek(rd, pos) >> (32 - padb)) != 0) err = 5;      // libFLAC read_zero_padding_: lost sync
    }
    if (lane == 0) {
        results[f].err = err;
        if (counters) {
            atomicAdd(&counters[0], (unsigned long long)n_batches);
            atomicAdd(&counters[1], (unsigned long long)n_rounds);
            atomicAdd(&counters[2], (unsigned long long)n_hard);
        }
    }
}


// ------------------------------------------------------------------------------------------------ restore
// fg_dec_wrestore_kernel: the prediction recurrence and the output stage behind the wave-parallel parser.
//
//   workgroup = 64 chains (chain = one subframe: frame f, channel ch; chains of a frame are neighbours), four waves:
//   wave 0     recurrence  lane = chain: s[i] = r[i] + ((sum_j q[j] s[i-1-j]) >> shift) over a 64-sample tile in LDS, the history
//                          in registers (slot of a sample = i mod 4 / 8 / 16: static register indices); one v_mad_i32_i24 per
//                          tap, the older taps in two interleaved chains.  This is the one serial chain of the decoder that
//                          nothing parallelises (the floor in the shift makes the recurrence non-linear): 4096 steps of about
//                          (order + 3) instructions for a block of 4096.
//   wave 1     loader      tiles t + 1 and t + 2 of all 64 residual rows from the plane straight into LDS
//                          (global_load_lds_dwordx4: no registers, no landing pass; sixteen instructions a tile).  Two tiles are
//                          in flight because a step of the recurrence (1.4 us) is shorter than a round trip to the loaded
//                          memory system (2.5 us): with one tile in flight the kernel ran at the pace of that latency.
//   waves 2-3  writers     tile t - 1 out: wasted-bits shift, stereo decorrelation undone, channels interleaved, 16-byte
//                          stores; silence for frames that failed.  They never wait for memory (a wave that loads AND stores
//                          waits for its stores' round trip whenever it needs a load: memory operations retire in order).
//   Four tile buffers (two landing, one computing, one being written out), one workgroup barrier per tile.
//
//   Tile layout: row r = 64 words, no padding (the DMA writes 64 lanes x 16 bytes contiguously); the four-word group g of
//   row r sits at slot g ^ (r & 15), so that the sixteen lanes of one LDS access cycle -- sixteen different rows, same g -- hit
//   sixteen different slots.  The loader permutes the global addresses accordingly (still whole 256-byte rows per 16 lanes).
//   What lies behind the end of a row, or in rows without a frame, is garbage by design: the recurrence runs over it, the
//   writers never look at it.
//
// Replaces fg_dec_restore_kernel (flac_dec_fast.hip: 32 chains per two-wave workgroup, tiles of 192) behind the wave parser;
// same inputs (residual plane, FgDecSub, parse status + CRC verdict) and the same output contract.
// Reference path replaced: FLAC__fixed_restore_signal, FLAC__lpc_restore_signal, undo_channel_coding.
#ifndef FGX_DEC_DOT2
#define FGX_DEC_DOT2 1             // the 16-bit restore chain through v_dot2_i32_i16 (0: one v_mad_i32_i24 a tap)
#endif
#define WR_TS 64
#define WR_NB 4
#define WR_FA 12            // words of facts per chain
#define WR_TILE_W (64 * WR_TS)

// byte offset of group g (four samples) inside a lane's row; xr = (row & 15) << 4
__device__ __forceinline__ uint32_t wr_goff(uint32_t g, uint32_t xr) { return (g << 4) ^ xr; }

// Eight steps of the 8-tap recurrence as one block of assembly: per sample v_mul_i32_i24 + 7 v_mad_i32_i24 (one chain -- a lone
// wave issues an instruction every ~4.5 cycles whether or not it depends on the last one), the shift, the add that turns the
// residual into the sample IN PLACE.  Ten instructions a sample.  n[0..7] come in as the eight residuals (the registers of two
// 16-byte LDS reads) and go out as the eight samples (the registers of two 16-byte LDS writes, and the history of the next
// group): no register moves around the block.  Written out because the compiler does not: left alone it builds the sum from
// v_mul_i32_i24 and v_add3_u32 (three instructions for two taps), and a mad given to it one asm statement at a time is
// followed by a pad s_nop each.
__device__ __forceinline__ void wr_group8_asm(const int32_t (&h)[16], const int32_t (&q)[16], int shift, int32_t (&n)[8])
{
    int32_t t;
    asm volatile("v_mul_i32_i24 %[t], %[q7], %[h0]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[h7], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n0], %[n0], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h1]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n0], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n1], %[n1], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h2]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n1], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n2], %[n2], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h3]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n2], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n3], %[n3], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h4]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n3], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n4], %[n4], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h5]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n4], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n5], %[n5], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h6]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n5], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n6], %[n6], %[t]\n"
                 "v_mul_i32_i24 %[t], %[q7], %[h7]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n6], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 "v_add_u32 %[n7], %[n7], %[t]\n"
                 : [t] "=&v"(t), [n0] "+v"(n[0]), [n1] "+v"(n[1]), [n2] "+v"(n[2]), [n3] "+v"(n[3]), [n4] "+v"(n[4]), [n5] "+v"(n[5]),
                   [n6] "+v"(n[6]), [n7] "+v"(n[7])
                 : [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [q4] "v"(q[4]), [q5] "v"(q[5]), [q6] "v"(q[6]), [q7] "v"(q[7]),
                   [sh] "v"(shift), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]),
                   [h6] "v"(h[6]), [h7] "v"(h[7]));
}

// MAXO samples from sample index i0 of the tile (a multiple of MAXO)
template <int MAXO, bool WIDE, bool GATE>
__device__ __forceinline__ void wr_group(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, uint32_t i0, char *rowb, uint32_t xr)
{
    constexpr int TAPS = MAXO == 16 ? 12 : MAXO;
    int32_t r[MAXO];
#pragma unroll
    for (int u = 0; u < MAXO; u += 4) {
        const uint4 t = *(const uint4 *)(rowb + wr_goff((i0 + u) >> 2, xr));
        r[u] = (int32_t)t.x; r[u + 1] = (int32_t)t.y; r[u + 2] = (int32_t)t.z; r[u + 3] = (int32_t)t.w;
    }
    if constexpr (MAXO == 8 && !WIDE && !GATE) {
        int32_t n8[8] = {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]};
        wr_group8_asm(h, q, shift, n8);
#pragma unroll
        for (int u = 0; u < 8; u++) h[u] = n8[u];
        *(uint4 *)(rowb + wr_goff(i0 >> 2, xr)) = make_uint4((uint32_t)n8[0], (uint32_t)n8[1], (uint32_t)n8[2], (uint32_t)n8[3]);
        *(uint4 *)(rowb + wr_goff((i0 + 4) >> 2, xr)) = make_uint4((uint32_t)n8[4], (uint32_t)n8[5], (uint32_t)n8[6], (uint32_t)n8[7]);
        return;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u++) {
        int32_t pred;
        if (!WIDE) {
            // (the 8-tap form of whole tiles is wr_group8_asm; this one serves the first 16 samples and orders above 8)
            int32_t sum = 0;
#pragma unroll
            for (int j = TAPS - 1; j >= 0; j--) sum += __mul24(q[j], h[(u - 1 - j + 2 * MAXO) % MAXO]);
            pred = sum >> shift;
        }
        else {
            i64 sum = 0;
#pragma unroll
            for (int j = TAPS - 1; j >= 0; j--) sum += (i64)q[j] * (i64)h[(u - 1 - j + 2 * MAXO) % MAXO];
            pred = (int32_t)(sum >> shift);
        }
        int32_t v = r[u] + pred;
        if (GATE) v = (i0 + (uint32_t)u >= order) ? v : r[u];
        h[u] = v;
        r[u] = v;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u += 4)
        *(uint4 *)(rowb + wr_goff((i0 + u) >> 2, xr)) = make_uint4((uint32_t)r[u], (uint32_t)r[u + 1], (uint32_t)r[u + 2], (uint32_t)r[u + 3]);
}

template <int MAXO, bool WIDE>
__device__ __forceinline__ void wr_tile(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, bool first, char *rowb, uint32_t xr)
{
    if (first) {
        // (the warm-up samples, at most 12, lie in the first 16)
#pragma unroll
        for (int g = 0; g < 16 / MAXO; g++) wr_group<MAXO, WIDE, true>(h, q, shift, order, g * MAXO, rowb, xr);
#pragma unroll
        for (int g = 16 / MAXO; g < WR_TS / MAXO; g++) wr_group<MAXO, WIDE, false>(h, q, shift, order, g * MAXO, rowb, xr);
    }
    else if constexpr (MAXO == 8 && !WIDE) {
        // the residuals of group g + 1 are requested before group g is computed: an LDS round trip is 100+ cycles, a fifth of a group
        uint4 ra = *(const uint4 *)(rowb + wr_goff(0, xr)), rb = *(const uint4 *)(rowb + wr_goff(1, xr));
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint4 na = ra, nb = rb;
            if (g < 7) { na = *(const uint4 *)(rowb + wr_goff(2 * g + 2, xr)); nb = *(const uint4 *)(rowb + wr_goff(2 * g + 3, xr)); }
            int32_t n8[8] = {(int32_t)ra.x, (int32_t)ra.y, (int32_t)ra.z, (int32_t)ra.w, (int32_t)rb.x, (int32_t)rb.y, (int32_t)rb.z, (int32_t)rb.w};
            wr_group8_asm(h, q, shift, n8);
#pragma unroll
            for (int u = 0; u < 8; u++) h[u] = n8[u];
            *(uint4 *)(rowb + wr_goff(2 * g, xr)) = make_uint4((uint32_t)n8[0], (uint32_t)n8[1], (uint32_t)n8[2], (uint32_t)n8[3]);
            *(uint4 *)(rowb + wr_goff(2 * g + 1, xr)) = make_uint4((uint32_t)n8[4], (uint32_t)n8[5], (uint32_t)n8[6], (uint32_t)n8[7]);
            ra = na; rb = nb;
        }
    }
    else {
#pragma unroll
        for (int g = 0; g < WR_TS / MAXO; g++) wr_group<MAXO, WIDE, false>(h, q, shift, order, g * MAXO, rowb, xr);
    }
}

// ---- 16-bit residual tiles (P16).  Row r of an input tile = 64 halfwords = 32 words; the eight-sample group g (16 bytes) of row r
// sits at slot g ^ (r & 7): the eight lanes of one LDS access cycle -- eight different rows, same g -- then cover all 64 banks
// (rows 128 bytes apart alternate between the two halves of the banks, the slot picks the place inside a half).  The samples leave
// the recurrence as 32-bit values in a tile of the usual form (wr_goff), which the writers read.
#define WR_IN16_W (64 * 32)
// ring of 16-bit residual tiles: one being computed, WR_R16 - 1 landing (three, four or five: the same decode launch -- the loader is
// not what the kernel waits for; four, because wr16_chain8 wants an even ring).
// (timing experiments of the restore kernel -- FLACGPU_DEC_SKIP -- are compiled into tuning builds only)
#ifdef FG_TUNING
#define WR_DBG 1
#else
#define WR_DBG 0
#endif
#ifndef WR_R16
#define WR_R16 4
#endif
#define WR_WORDS ((WR_NB * WR_TILE_W) > (WR_R16 * WR_IN16_W + 2 * WR_TILE_W) ? (WR_NB * WR_TILE_W) : (WR_R16 * WR_IN16_W + 2 * WR_TILE_W))
__device__ __forceinline__ uint32_t wr16_goff(uint32_t g8, uint32_t xr8) { return (g8 << 4) ^ xr8; }      // xr8 = (row & 7) << 4
__device__ __forceinline__ void wr16_unpack(const uint4 t, int32_t (&r)[8])
{
    r[0] = (int32_t)(t.x << 16) >> 16; r[1] = (int32_t)t.x >> 16; r[2] = (int32_t)(t.y << 16) >> 16; r[3] = (int32_t)t.y >> 16;
    r[4] = (int32_t)(t.z << 16) >> 16; r[5] = (int32_t)t.z >> 16; r[6] = (int32_t)(t.w << 16) >> 16; r[7] = (int32_t)t.w >> 16;
}
// wr_group8_asm with the residuals packed two to a register: the add that turns a residual into its sample takes its halfword
// through SDWA (sign-extended), so the narrower plane costs the recurrence -- the one serial chain of the decoder -- nothing.
__device__ __forceinline__ void wr16_group8_asm(const int32_t (&h)[16], const int32_t (&q)[16], int shift, const uint4 p, int32_t (&n)[8])
{
    int32_t t;
#define WR16_ADD(ni, pj, half) "v_add_u32_sdwa %[" #ni "], sext(%[" #pj "]), %[t] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_" #half " src1_sel:DWORD\n"
    asm volatile("v_mul_i32_i24 %[t], %[q7], %[h0]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[h7], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n0, p0, 0)
                 "v_mul_i32_i24 %[t], %[q7], %[h1]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n0], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n1, p0, 1)
                 "v_mul_i32_i24 %[t], %[q7], %[h2]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n1], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n2, p1, 0)
                 "v_mul_i32_i24 %[t], %[q7], %[h3]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n2], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n3, p1, 1)
                 "v_mul_i32_i24 %[t], %[q7], %[h4]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n3], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n4, p2, 0)
                 "v_mul_i32_i24 %[t], %[q7], %[h5]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h6], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n4], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n5, p2, 1)
                 "v_mul_i32_i24 %[t], %[q7], %[h6]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[h7], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n5], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n6, p3, 0)
                 "v_mul_i32_i24 %[t], %[q7], %[h7]\n"
                 "v_mad_i32_i24 %[t], %[q6], %[n0], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q5], %[n1], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q4], %[n2], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q3], %[n3], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q2], %[n4], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q1], %[n5], %[t]\n"
                 "v_mad_i32_i24 %[t], %[q0], %[n6], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n7, p3, 1)
                 : [t] "=&v"(t), [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [n4] "=&v"(n[4]), [n5] "=&v"(n[5]),
                   [n6] "=&v"(n[6]), [n7] "=&v"(n[7])
                 : [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [q4] "v"(q[4]), [q5] "v"(q[5]), [q6] "v"(q[6]), [q7] "v"(q[7]),
                   [sh] "v"(shift), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]),
                   [h6] "v"(h[6]), [h7] "v"(h[7]), [p0] "v"(p.x), [p1] "v"(p.y), [p2] "v"(p.z), [p3] "v"(p.w));
#undef WR16_ADD
}

// The same eight steps with the history two samples to a register (round 5): X(k) = (s[k] in the low half, s[k - 1] in the high half),
// one register per sample, so that step i takes its eight taps as FOUR v_dot2_i32_i16 -- X(i - 1), X(i - 3), X(i - 5), X(i - 7)
// against the coefficient pairs (q0, q1) .. (q6, q7) -- and makes X(i) with one v_perm_b32: seven instructions a sample where the
// multiply-add chain takes ten, and this chain is the one thing in the decoder nothing parallelises (4096 steps a block at a lone
// wave's 4.5 cycles an instruction).  It holds while the SAMPLES fit 16 bits -- left, right and mid of a 16-bit stream always do, a
// side channel nearly always; the writers look at every sample on its way out, and a frame with a sample beyond 16 bits ends
// with status 6: the call is repeated with 32-bit planes and the multiply-add chain (flacgpu_dec_api.cpp, dec_p16_hold).
// xp[1..7] = X(i0 - 7) .. X(i0 - 1) come in, x[0..7] = X(i0) .. X(i0 + 7) and the eight samples n[0..7] go out.
__device__ __forceinline__ void wr16_group8_dot2_asm(const uint32_t (&xp)[8], const uint32_t (&qq)[4], int shift, uint32_t sel, const uint4 p,
                                                     int32_t (&n)[8], uint32_t (&x)[8])
{
    int32_t t;
#define WR16_ADD(ni, pj, half) "v_add_u32_sdwa %[" #ni "], sext(%[" #pj "]), %[t] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_" #half " src1_sel:DWORD\n"
    asm volatile("v_dot2_i32_i16 %[t], %[xp7], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[xp5], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp3], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp1], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n0, p0, 0)
                 "v_perm_b32 %[x0], %[xp7], %[n0], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x0], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[xp6], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp4], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp2], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n1, p0, 1)
                 "v_perm_b32 %[x1], %[x0], %[n1], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x1], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[xp7], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp5], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp3], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n2, p1, 0)
                 "v_perm_b32 %[x2], %[x1], %[n2], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x2], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[x0], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp6], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp4], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n3, p1, 1)
                 "v_perm_b32 %[x3], %[x2], %[n3], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x3], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[x1], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp7], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp5], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n4, p2, 0)
                 "v_perm_b32 %[x4], %[x3], %[n4], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x4], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[x2], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[x0], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp6], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n5, p2, 1)
                 "v_perm_b32 %[x5], %[x4], %[n5], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x5], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[x3], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[x1], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[xp7], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n6, p3, 0)
                 "v_perm_b32 %[x6], %[x5], %[n6], %[sel]\n"
                 "v_dot2_i32_i16 %[t], %[x6], %[q01], 0\n"
                 "v_dot2_i32_i16 %[t], %[x4], %[q23], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[x2], %[q45], %[t]\n"
                 "v_dot2_i32_i16 %[t], %[x0], %[q67], %[t]\n"
                 "v_ashrrev_i32 %[t], %[sh], %[t]\n"
                 WR16_ADD(n7, p3, 1)
                 "v_perm_b32 %[x7], %[x6], %[n7], %[sel]\n"
                 : [t] "=&v"(t), [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [n4] "=&v"(n[4]), [n5] "=&v"(n[5]),
                   [n6] "=&v"(n[6]), [n7] "=&v"(n[7]), [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]), [x4] "=&v"(x[4]),
                   [x5] "=&v"(x[5]), [x6] "=&v"(x[6]), [x7] "=&v"(x[7])
                 : [q01] "v"(qq[0]), [q23] "v"(qq[1]), [q45] "v"(qq[2]), [q67] "v"(qq[3]), [sh] "v"(shift), [sel] "v"(sel),
                   [xp1] "v"(xp[1]), [xp2] "v"(xp[2]), [xp3] "v"(xp[3]), [xp4] "v"(xp[4]), [xp5] "v"(xp[5]), [xp6] "v"(xp[6]), [xp7] "v"(xp[7]),
                   [p0] "v"(p.x), [p1] "v"(p.y), [p2] "v"(p.z), [p3] "v"(p.w));
#undef WR16_ADD
}

// MAXO samples from sample index i0 of the tile (a multiple of MAXO): residuals from the 16-bit row, samples to the 32-bit row
template <int MAXO, bool GATE>
__device__ __forceinline__ void wr16_group(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, uint32_t i0, const char *rin,
                                           char *rout, uint32_t xr8, uint32_t xr)
{
    constexpr int TAPS = MAXO == 16 ? 12 : MAXO;
    int32_t r[MAXO];
    if constexpr (MAXO == 4) {
        const uint2 t = *(const uint2 *)(rin + wr16_goff(i0 >> 3, xr8) + ((i0 & 4) ? 8u : 0u));
        r[0] = (int32_t)(t.x << 16) >> 16; r[1] = (int32_t)t.x >> 16; r[2] = (int32_t)(t.y << 16) >> 16; r[3] = (int32_t)t.y >> 16;
    }
    else {
#pragma unroll
        for (int u = 0; u < MAXO; u += 8) {
            int32_t r8[8];
            wr16_unpack(*(const uint4 *)(rin + wr16_goff((i0 + u) >> 3, xr8)), r8);
#pragma unroll
            for (int e = 0; e < 8; e++) r[u + e] = r8[e];
        }
    }
#pragma unroll
    for (int u = 0; u < MAXO; u++) {
        int32_t sum = 0;
#pragma unroll
        for (int j = TAPS - 1; j >= 0; j--) sum += __mul24(q[j], h[(u - 1 - j + 2 * MAXO) % MAXO]);
        int32_t v = r[u] + (sum >> shift);
        if (GATE) v = (i0 + (uint32_t)u >= order) ? v : r[u];
        h[u] = v;
        r[u] = v;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u += 4)
        *(uint4 *)(rout + wr_goff((i0 + u) >> 2, xr)) = make_uint4((uint32_t)r[u], (uint32_t)r[u + 1], (uint32_t)r[u + 2], (uint32_t)r[u + 3]);
}

template <int MAXO>
__device__ __forceinline__ void wr16_tile(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, bool first, const char *rin, char *rout,
                                          uint32_t xr8, uint32_t xr)
{
    if (first) {
        // (the warm-up samples, at most 12, lie in the first 16)
#pragma unroll
        for (int g = 0; g < 16 / MAXO; g++) wr16_group<MAXO, true>(h, q, shift, order, g * MAXO, rin, rout, xr8, xr);
#pragma unroll
        for (int g = 16 / MAXO; g < WR_TS / MAXO; g++) wr16_group<MAXO, false>(h, q, shift, order, g * MAXO, rin, rout, xr8, xr);
    }
    else if constexpr (MAXO == 8) {
        // the residuals of group g + 1 are requested before group g is computed (one 16-byte read a group)
        uint4 ra = *(const uint4 *)(rin + wr16_goff(0, xr8));
#if FGX_DEC_DOT2
        // (the history two samples to a register, wr16_group8_dot2_asm; h[] -- the last eight samples as 32-bit values -- is what the
        // tiles hand each other)
        const uint32_t sel = 0x05040100u;
        auto pk = [](int32_t lo, int32_t hi) -> uint32_t { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); };
        uint32_t xp[8], qq[4];
        xp[0] = 0;
#pragma unroll
        for (int k = 1; k < 8; k++) xp[k] = pk(h[k], h[k - 1]);
#pragma unroll
        for (int k = 0; k < 4; k++) qq[k] = pk(q[2 * k], q[2 * k + 1]);
        int32_t n8[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint4 na = ra;
            if (g < 7) na = *(const uint4 *)(rin + wr16_goff(g + 1, xr8));
            uint32_t x8[8];
            wr16_group8_dot2_asm(xp, qq, shift, sel, ra, n8, x8);
#pragma unroll
            for (int u = 0; u < 8; u++) xp[u] = x8[u];
            *(uint4 *)(rout + wr_goff(2 * g, xr)) = make_uint4((uint32_t)n8[0], (uint32_t)n8[1], (uint32_t)n8[2], (uint32_t)n8[3]);
            *(uint4 *)(rout + wr_goff(2 * g + 1, xr)) = make_uint4((uint32_t)n8[4], (uint32_t)n8[5], (uint32_t)n8[6], (uint32_t)n8[7]);
            ra = na;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) h[u] = n8[u];
#else
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint4 na = ra;
            if (g < 7) na = *(const uint4 *)(rin + wr16_goff(g + 1, xr8));
            int32_t n8[8];
            wr16_group8_asm(h, q, shift, ra, n8);
#pragma unroll
            for (int u = 0; u < 8; u++) h[u] = n8[u];
            *(uint4 *)(rout + wr_goff(2 * g, xr)) = make_uint4((uint32_t)n8[0], (uint32_t)n8[1], (uint32_t)n8[2], (uint32_t)n8[3]);
            *(uint4 *)(rout + wr_goff(2 * g + 1, xr)) = make_uint4((uint32_t)n8[4], (uint32_t)n8[5], (uint32_t)n8[6], (uint32_t)n8[7]);
            ra = na;
        }
#endif
    }
    else {
#pragma unroll
        for (int g = 0; g < WR_TS / MAXO; g++) wr16_group<MAXO, false>(h, q, shift, order, g * MAXO, rin, rout, xr8, xr);
    }
}

// One tile of the packed-history chain with the history left packed between tiles: xp[1..7] = X(i0 - 7) .. X(i0 - 1) in and out.
// ain[g] / aout[j]: this lane's LDS addresses of residual group g and sample quad j in ring slot 0 / output tile 0; in_off and
// out_off pick the tile.  With constant offsets (wr16_chain8's trips of four tiles) they are the immediates of the LDS instructions:
// the tile is its 448 chain instructions and 24 LDS instructions, nothing else -- a lone wave pays 2 ns for every instruction, and
// what the general form (wr16_tile) spends per tile on addresses and on packing and unpacking the history was 0.2 of its 1.3 us.
__device__ __forceinline__ void wr16_tile8p(uint32_t (&xp)[8], const uint32_t (&qq)[4], int shift, const char *const (&ain)[8], char *const (&aout)[16],
                                            uint32_t in_off, uint32_t out_off)
{
    const uint32_t sel = 0x05040100u;
    // (requesting the residuals two groups ahead instead of one changes nothing: 0.305 ms either way)
    uint4 ra = *(const uint4 *)(ain[0] + in_off);
#pragma unroll
    for (int g = 0; g < 8; g++) {
        uint4 na = ra;
        if (g < 7) na = *(const uint4 *)(ain[g + 1] + in_off);
        int32_t n8[8];
        uint32_t x8[8];
        wr16_group8_dot2_asm(xp, qq, shift, sel, ra, n8, x8);
#pragma unroll
        for (int u = 0; u < 8; u++) xp[u] = x8[u];
        *(uint4 *)(aout[2 * g] + out_off) = make_uint4((uint32_t)n8[0], (uint32_t)n8[1], (uint32_t)n8[2], (uint32_t)n8[3]);
        *(uint4 *)(aout[2 * g + 1] + out_off) = make_uint4((uint32_t)n8[4], (uint32_t)n8[5], (uint32_t)n8[6], (uint32_t)n8[7]);
        ra = na;
    }
}

// All steps of the recurrence wave for chains of order 5 .. 8 on 16-bit planes: tile 0 by the general form (its warm-up samples are
// gated), then trips of WR_R16 tiles with every LDS offset a constant, then what is left.  (WR_R16 is even: the output tile of
// step s is s & 1.)
__device__ __forceinline__ void wr16_chain8(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, uint32_t T, uint32_t S, uint32_t *in16,
                                            uint32_t *otile, int lane, bool report)
{
    unsigned long long tb0 = 0, tbar = 0;
    const unsigned long long tall = WR_DBG ? wall_clock64() : 0;
#define WR_BARRIER() do { if (WR_DBG) tb0 = wall_clock64(); __syncthreads(); if (WR_DBG) tbar += wall_clock64() - tb0; } while (0)
    static_assert((WR_R16 & 1) == 0 && WR_R16 >= 4, "wr16_chain8: an even ring of at least four tiles");
    const uint32_t xr = ((uint32_t)lane & 15) << 4, xr8 = ((uint32_t)lane & 7) << 4;
    WR_BARRIER();
    if (T == 0) return;
    wr16_tile<8>(h, q, shift, order, true, (const char *)(in16 + (uint32_t)lane * 32), (char *)(otile + (uint32_t)lane * WR_TS), xr8, xr);
    auto pk = [](int32_t lo, int32_t hi) -> uint32_t { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); };
    uint32_t xp[8], qq[4];
    xp[0] = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) xp[k] = pk(h[k], h[k - 1]);
#pragma unroll
    for (int k = 0; k < 4; k++) qq[k] = pk(q[2 * k], q[2 * k + 1]);
    const char *ain[8];
    char *aout[16];
#pragma unroll
    for (int g = 0; g < 8; g++) ain[g] = (const char *)(in16 + (uint32_t)lane * 32) + wr16_goff((uint32_t)g, xr8);
#pragma unroll
    for (int j = 0; j < 16; j++) aout[j] = (char *)(otile + (uint32_t)lane * WR_TS) + wr_goff((uint32_t)j, xr);
    uint32_t s = 2;
    // (a trip: steps s .. s + WR_R16 - 1 with s = 2 mod WR_R16, i.e. tiles 1, 2, .. of the ring's slots 1, 2, .., 0)
    for (; s + WR_R16 - 1 <= T; s += WR_R16) {
#pragma unroll
        for (int k = 0; k < WR_R16; k++) {
            WR_BARRIER();
            wr16_tile8p(xp, qq, shift, ain, aout, (uint32_t)((1 + k) % WR_R16) * (WR_IN16_W * 4), (uint32_t)((1 + k) & 1) * (WR_TILE_W * 4));
        }
    }
    for (; s <= S; s++) {
        WR_BARRIER();
        if (s > T) continue;
        const uint32_t t = s - 1;
        wr16_tile8p(xp, qq, shift, ain, aout, (t % WR_R16) * (WR_IN16_W * 4), (t & 1) * (WR_TILE_W * 4));
    }
    if (WR_DBG && report && blockIdx.x == 100 && lane == 0)
        printf("restore wg 100 chain wave: %u steps, %llu ticks of 10 ns, %llu of them at barriers\n", S, wall_clock64() - tall, tbar);
#undef WR_BARRIER
}

template <bool WIDE, bool P16>
__global__ void __launch_bounds__(256)
fg_dec_wrestore_kernel(const FgDecFrame *frames, uint32_t nframes, uint32_t C, const FgDecSub *subs, const int32_t *scratch,
                       int32_t *out, FgDecResult *results, uint32_t interleave, FgDecResult *host_rows, const unsigned long long *planeoff,
                       unsigned long long *join, unsigned long long epoch)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t wsm[];
    uint32_t *const tiles = wsm;                                   // WR_NB x 64 rows x 64 words
    // (P16: three tiles of 16-bit residuals -- two landing, one being computed -- and two tiles of samples -- one being computed,
    // one being written out: 56 KB)
    uint32_t *const in16 = wsm;                                    // WR_R16 x WR_IN16_W
    uint32_t *const otile = wsm + WR_R16 * WR_IN16_W;              // 2 x WR_TILE_W
    uint32_t *const fa = wsm + WR_WORDS;                           // 64 x WR_FA: n_in, n_out, plane lo/hi, out_off lo/hi, ca, wasted, n
    uint32_t *const ctl = fa + 64 * WR_FA;                         // [0] nmax, [1] all planes 16-byte aligned, [2] stereo fast output, [3] 16-bit check, [4] regular, [5] wasted bits somewhere
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    // whole frames per workgroup: G = 64 / C frames, lanes G C .. 63 idle (a frame's status is merged by the workgroup that owns it)
    const uint32_t G = 64 / C;
    const uint32_t f = blockIdx.x * G + (uint32_t)lane / C, ch = (uint32_t)lane % C;
    const bool mine = (uint32_t)lane < G * C && f < nframes;

    int32_t q[16], h[16];
    uint32_t order = 0;
    int shift = 0;
    bool dead = false;          // (wave 0: the wait for the join word ran out)
    if (wave == 0) {
        // Join (round 6, by the rules of the HSA memory model instead of an argument about timing): header pass, scan and CRC pass ran
        // beside the parser on streams of their own, and the main stream did not wait for them.  The kernel queued behind them
        // stores this call's epoch into join[0] with RELEASE order at agent scope -- every kernel in front of it on its stream has
        // ended by then, and the CRC stream's event has been waited for, so everything they wrote happens-before that store.  This
        // wave reads the word relaxed (one word, every lane the same address) until it holds the epoch, and then -- always, also
        // when the first look found it raised: the side streams' kernels may have ended between this kernel's start, which
        // emptied the caches, and that look, and a workgroup of theirs on this CU may have left lines of the tables in its L1 --
        // takes ONE agent-scope ACQUIRE fence (buffer_inv sc1).  The fence synchronises with the release; the plain loads of
        // frames[], results[], planeoff[] and subs[] behind it, in this wave and -- behind the __syncthreads() below -- in the
        // others, see what those kernels left.  (Round 5 had no fence here and argued the loop never turns; 1.7 us of a launch.)
        // A wait that runs out (a tool that serialises kernels across streams has queued this kernel in front of the one that
        // raises the word) sets bit 1 of join[-2] and this workgroup does NOTHING: the tables are not final, out_off may be a
        // previous call's -- no sample is written, no status word touched.  The host repeats the call with events.
        // join[-2]'s upper half counts the workgroups whose first look found the word not yet raised (flacgpu_decode_stats).
        if (join) {
            unsigned long long seen = __hip_atomic_load(join, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = ((unsigned long long)rfl((uint32_t)(seen >> 32)) << 32) | rfl((uint32_t)seen);
            if (seen < epoch) {
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    if (wall_clock64() - t0 > FG_GATE_TICKS) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(16);
                    seen = __hip_atomic_load(join, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = ((unsigned long long)rfl((uint32_t)(seen >> 32)) << 32) | rfl((uint32_t)seen);
                    if (seen >= epoch) break;
                }
                // (one add a workgroup: + 1 in the upper half, and + 2 in the lower half for a wait that ran out -- the lower half
                // stays far below 2^32 and is non-zero exactly when some wait ran out, which is what the host tests)
                if (lane == 0) atomicAdd(join - 2, (1ull << 32) | (dead ? 2ull : 0ull));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        // ---- facts of this chain (and the CRC verdict merged into the frame status, as fg_dec_restore_kernel does)
        uint32_t n = 0, status = 1, ca = 0, wasted = 0;
        u64 out_off = 0;
        if (mine && !dead) {
            const FgDecFrame fr = frames[f];
            if (fr.bytes != 0 && fr.channels == C) {
                n = fr.n; status = results[f].err; ca = fr.ca; out_off = fr.out_off;
                // (interleave bit 11, unused since round 6: the CRC verdict is not to be looked at)
                const uint32_t cw = (interleave & 0x800) ? 0u : results[f].crc;
                if (status == 0 && (cw & 0x80000000u)) status = 2;          // CRC-16 mismatch (fg_dec_crc_kernel)
            }
        }
        const bool ok = mine && n != 0 && status == 0;
#pragma unroll
        for (int j = 0; j < 16; j++) { q[j] = 0; h[j] = 0; }
        if (ok) {
            const FgDecSub *sd = &subs[(size_t)f * C + ch];
            order = sd->order; shift = sd->shift; wasted = sd->wasted;
#pragma unroll
            for (int j = 0; j < FG_DMAXO; j++) if ((uint32_t)j < order) q[j] = sd->q[j];
        }
        const uint32_t n_in = ok ? n : 0;
        const uint32_t n_out = (mine && !dead && status != 3) ? n : 0;      // status 3: the generic kernel writes the frame
        // (planeoff: the parser placed the frames' parts of the plane itself, FgDecSelf)
        const u64 plane = ok ? (planeoff ? (u64)(planeoff[f] >> 2) : out_off * C) + (u64)ch * n : 0;
        uint32_t *fm = fa + lane * WR_FA;
        fm[0] = n_in; fm[1] = n_out; fm[2] = (uint32_t)plane; fm[3] = (uint32_t)(plane >> 32);
        fm[4] = (uint32_t)out_off; fm[5] = (uint32_t)(out_off >> 32); fm[6] = ca; fm[7] = wasted; fm[8] = n;
        uint32_t nmax = n_out;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o); nmax = t > nmax ? t : nmax; }
        // stereo fast output (decided per frame by the writers): C == 2 and a 16-byte aligned output base
        const bool so = C == 2 && (((uintptr_t)out) & 15) == 0;
        // every frame of the workgroup there, good, of the same length, at an even place: the writers' tiles in front of the last one
        // need no test per sample (ctl[4])
        const bool reg = n_in == nmax && n_out == nmax && (out_off & 1) == 0 && ((interleave & 1) || (nmax & 3) == 0);
        const bool allreg = so && G * C == 64 && __all(mine && reg) && !WIDE;
        const bool anyw = __any(wasted != 0);
        // (the packed-history chain holds while the samples fit 16 bits: the writers check, ctl[3] tells them to)
        const bool dot2 = P16 && FGX_DEC_DOT2 && !__any(order > 8) && __any(order > 4);
        if (lane == 0) { ctl[0] = nmax; ctl[1] = 1u; ctl[2] = so ? 1u : 0u; ctl[3] = dot2 ? 1u : 0u; ctl[4] = allreg ? 1u : 0u; ctl[5] = anyw ? 1u : 0u; }
    }
    __syncthreads();
    // (the status merge is written after every wave has read what it needs: only wave 0 reads results[])
    // (host_rows: the status words go to the host's pinned copy from here -- the call then ends with the few words of
    // fg_signal_kernel instead of a pass over all frames' words -- round 4's fg_export_kernel: 10 us of a decode launch)
    if (wave == 0 && mine && !dead && ch == 0 && !(interleave & 0x800)) {
        const FgDecFrame fr = frames[f];
        uint32_t status = results[f].err, cw = results[f].crc;
        if (fr.bytes != 0 && fr.channels == C && fr.n != 0) {
            if (status == 0 && (cw & 0x80000000u)) status = 2;
            cw &= 0xFFFFu;
            results[f].err = status; results[f].crc = cw;
        }
        if (host_rows) { FgDecResult r; r.err = status; r.crc = cw; host_rows[f] = r; }
    }
    const uint32_t nmax = ctl[0];
    const bool stereo_fast = ctl[2] != 0;
    constexpr bool narrow = P16;
    const uint32_t T = (nmax + WR_TS - 1) / WR_TS;
    const uint32_t S = T + 1;                  // steps (barriers) of every wave

    if (wave == 0) {
        const bool big = __any(order > 8), small = !__any(order > 4);
        const uint32_t xr = ((uint32_t)lane & 15) << 4;
        if (P16 && narrow && FGX_DEC_DOT2 && !big && !small && !(WR_DBG && (interleave & 0x200))) {
            wr16_chain8(h, q, shift, order, T, S, in16, otile, lane, (interleave & 0x2000) != 0);
            return;
        }
        if (P16 && narrow) {
            const uint32_t xr8 = ((uint32_t)lane & 7) << 4;
            unsigned long long tb0 = 0, tbar = 0, tall = WR_DBG ? wall_clock64() : 0;
            for (uint32_t s = 1; s <= S; s++) {
                if (WR_DBG) tb0 = wall_clock64();
                __syncthreads();
                if (WR_DBG) tbar += wall_clock64() - tb0;
                if (WR_DBG && s == S && (interleave & 0x2000) && blockIdx.x == 100 && lane == 0)
                    printf("restore wg 100 chain wave: %u steps, %llu ticks of 10 ns, %llu of them at barriers\n", S, wall_clock64() - tall, tbar);
                if (s > T || (WR_DBG && (interleave & 0x200))) continue;
                const uint32_t t = s - 1;
                const char *rin = (const char *)(in16 + (t % WR_R16) * WR_IN16_W + (uint32_t)lane * 32);
                char *rout = (char *)(otile + (t & 1) * WR_TILE_W + (uint32_t)lane * WR_TS);
                if (big) wr16_tile<16>(h, q, shift, order, t == 0, rin, rout, xr8, xr);
                else if (small) wr16_tile<4>(h, q, shift, order, t == 0, rin, rout, xr8, xr);
                else wr16_tile<8>(h, q, shift, order, t == 0, rin, rout, xr8, xr);
            }
            return;
        }
        for (uint32_t s = 1; s <= S; s++) {
            __syncthreads();
            if (s > T || (WR_DBG && (interleave & 0x200))) continue;
            const uint32_t t = s - 1;
            char *rowb = (char *)(tiles + (t % WR_NB) * WR_TILE_W + (uint32_t)lane * WR_TS);
            if (big) wr_tile<16, WIDE>(h, q, shift, order, t == 0, rowb, xr);
            else if (small) wr_tile<4, WIDE>(h, q, shift, order, t == 0, rowb, xr);
            else wr_tile<8, WIDE>(h, q, shift, order, t == 0, rowb, xr);
        }
        return;
    }

    if (P16 && wave == 1) {
        {
            // ---- loader of 16-bit tiles: instruction k of a tile fills rows 8k .. 8k + 7 (128 bytes each): lane L writes slot L & 7 of
            // row 8k + (L >> 3), i.e. it fetches the eight-sample group (L & 7) ^ (row & 7).  Eight instructions a tile, two tiles in flight.
            u64 rbase[8];
            uint32_t rlen[8], rcol[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t row = 8u * (uint32_t)k + ((uint32_t)lane >> 3);
                const uint32_t *fm = fa + row * WR_FA;
                rbase[k] = ((u64)fm[3] << 32) | fm[2];
                rlen[k] = fm[0];
                rcol[k] = ((((uint32_t)lane & 7) ^ (row & 7)) << 3);
            }
            auto issue = [&](uint32_t t) __attribute__((always_inline)) {
                uint32_t *tb = in16 + (t % WR_R16) * WR_IN16_W;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t c0 = t * WR_TS + rcol[k];
                    const int16_t *src = (t < T && c0 < rlen[k]) ? (const int16_t *)(scratch + rbase[k]) + c0 : (const int16_t *)scratch;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(tb + 8 * k * 32), 16, 0, 0);
                }
            };
            // (tile s must have landed when barrier s + 1 publishes it: everything but the WR_R16 - 2 tiles requested behind it)
#pragma unroll
            for (int t0 = 0; t0 < WR_R16 - 1; t0++) issue((uint32_t)t0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (WR_R16 - 2)) : "memory");
            unsigned long long tb0 = 0, tbar = 0, tw0 = 0, twait = 0, tall = WR_DBG ? wall_clock64() : 0;
            for (uint32_t s = 1; s <= S; s++) {
                if (WR_DBG) tb0 = wall_clock64();
                asm volatile("s_barrier" ::: "memory");
                if (WR_DBG) tbar += wall_clock64() - tb0;
                issue(s + WR_R16 - 2);
                if (WR_DBG) tw0 = wall_clock64();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (WR_R16 - 2)) : "memory");
                if (WR_DBG) twait += wall_clock64() - tw0;
            }
            if (WR_DBG && (interleave & 0x2000) && blockIdx.x == 100 && lane == 0)
                printf("restore wg 100 loader: %llu ticks, %llu at barriers, %llu waiting for tiles\n", wall_clock64() - tall, tbar, twait);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    if (wave == 1) {
        // ---- loader.  Instruction k of a tile fills rows 4k .. 4k + 3: lane L writes slot L & 15 of row 4k + (L >> 4), i.e. it
        // fetches group (L & 15) ^ (row & 15).  Lanes behind the end of their row (and all lanes past the last tile) fetch from
        // the start of the plane area instead: every path issues the same sixteen loads, so the counts in the waits are exact.
        // Planes that are not 16-byte aligned (odd block sizes) go word by word: 64 loads a tile, lane = column.
        u64 rbase[16];
        uint32_t rlen[16], rcol[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t row = 4u * (uint32_t)k + ((uint32_t)lane >> 4);
            const uint32_t *fm = fa + row * WR_FA;
            rbase[k] = ((u64)fm[3] << 32) | fm[2];
            rlen[k] = fm[0];
            rcol[k] = ((((uint32_t)lane & 15) ^ (row & 15)) << 2);
        }
        // (the 16-byte reads need no more than the 4-byte alignment every plane has: odd block sizes -- the tail of nearly every
        // real stream -- take the same path; what a read fetches behind the end of its row is never looked at)
        auto issue = [&](uint32_t t) __attribute__((always_inline)) {
            uint32_t *tb = tiles + (t % WR_NB) * WR_TILE_W;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t c0 = t * WR_TS + rcol[k];
                const int32_t *src = (t < T && c0 < rlen[k]) ? scratch + rbase[k] + c0 : scratch;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(tb + 4 * k * WR_TS), 16, 0, 0);
            }
        };
        issue(0); issue(1);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        for (uint32_t s = 1; s <= S; s++) {
            // barrier s publishes tile s - 1 (and whatever older)
            asm volatile("s_barrier" ::: "memory");
            issue(s + 1);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ---- writers: 128 lanes
    const uint32_t ml = threadIdx.x - 128;
    uint32_t w_rn[4], w_cc[4], w_wa[4], w_wb[4];
    bool w_ok[4];
    u64 w_oo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t *fm = fa + ((((uint32_t)k * 128 + ml) >> 4) * 2) * WR_FA;
        w_rn[k] = fm[1]; w_ok[k] = fm[0] != 0; w_cc[k] = fm[6]; w_wa[k] = fm[7]; w_wb[k] = fm[WR_FA + 7];
        w_oo[k] = ((u64)fm[5] << 32) | fm[4];
    }
    // a sample of a chain (before wasted bits and channel undo) beyond 16 bits while wave 0 runs the packed-history chain: the frame
    // ends with status 6 and the call is repeated with 32-bit planes and the multiply-add chain (as for a residual beyond 16 bits)
    auto beyond16 = [&](uint32_t row) __attribute__((always_inline)) {
        const uint32_t f = blockIdx.x * (64 / C) + row / C;
        if (f < nframes) {
            results[f].err = 6;
            if (host_rows) host_rows[f].err = 6;
        }
    };
    const bool whole = ctl[4] != 0, anywaste = ctl[5] != 0, chk16w = P16 && ctl[3] != 0;
    auto writeout_whole = [&](uint32_t t, auto ILV, auto WASTE) __attribute__((always_inline)) {
        {
            const uint32_t *tb = (P16 && narrow) ? otile + (t & 1) * WR_TILE_W : tiles + (t % WR_NB) * WR_TILE_W;
            const uint32_t i0 = t * WR_TS;
            // a tile of whole groups of regular stereo frames (all but the last tile of nearly every workgroup): nothing to test
            // per sample, all eight LDS reads of the lane's four tasks requested before the first is looked at (nothing else runs
            // on this SIMD to hide them), one range verdict for the four, and the channel undo with the assignment's facts as lane
            // constants -- mid/side: left = a + ((b + 1) >> 1), right = a - (b >> 1)  [= ((2a | (b & 1)) +- b) >> 1: 2a is even];
            // in all four assignments left = a + k23 * ((b + k3) >> k3), right = k13 * a + sg * (b >> k3)
            uint4 ta[4], tb4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t id = (uint32_t)k * 128 + ml;
                const uint32_t r0 = (id >> 4) * 2, g = id & 15;
                ta[k] = make_uint4(g, g, g, g); tb4[k] = ta[k];
                if (!WR_DBG || !(interleave & 0x1000)) {
                    ta[k] = *(const uint4 *)&tb[r0 * WR_TS + ((g ^ (r0 & 15)) << 2)];
                    tb4[k] = *(const uint4 *)&tb[(r0 + 1) * WR_TS + ((g ^ ((r0 + 1) & 15)) << 2)];
                }
            }
            uint32_t ov = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + (ml & 15) * 4;
                uint32_t xa[4] = {ta[k].x, ta[k].y, ta[k].z, ta[k].w}, xb[4] = {tb4[k].x, tb4[k].y, tb4[k].z, tb4[k].w};
                if (P16) {
#pragma unroll
                    for (int e = 0; e < 4; e++) ov |= (xa[e] + 0x8000u) | (xb[e] + 0x8000u);
                }
                if (decltype(WASTE)::value) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { xa[e] <<= w_wa[k]; xb[e] <<= w_wb[k]; }
                }
                const uint32_t cc = w_cc[k];
                const int32_t k3 = cc == 3 ? 1 : 0, k23 = cc >= 2 ? 1 : 0, k13 = (int32_t)(cc & 1), sg = k13 ? -1 : 1;
                int32_t a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int32_t av = (int32_t)xa[e], bv = (int32_t)xb[e];
                    const int32_t up = (bv + k3) >> k3, dn = bv >> k3;
                    if (P16) {
                        // (streams of up to 16 bits: every value here has at most 17)
                        a[e] = av + __mul24(up, k23);
                        b[e] = __mul24(av, k13) + __mul24(dn, sg);
                    }
                    else {
                        a[e] = av + (k23 ? up : 0);
                        b[e] = k13 ? av - dn : bv;
                    }
                }
                int32_t *o = out + w_oo[k] * 2;
                if (WR_DBG && (interleave & 0x400)) {
                    // (experiment: everything but the stores)
                    asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(o));
                    continue;
                }
                if (decltype(ILV)::value) {
                    int4 *d = (int4 *)(o + (size_t)i * 2);
                    d[0] = make_int4(a[0], b[0], a[1], b[1]);
                    d[1] = make_int4(a[2], b[2], a[3], b[3]);
                }
                else {
                    *(int4 *)(o + i) = make_int4(a[0], a[1], a[2], a[3]);
                    *(int4 *)(o + nmax + i) = make_int4(b[0], b[1], b[2], b[3]);
                }
            }
            // (a value beyond 16 bits anywhere sends the whole call to the 32-bit planes: which of the lane's four frames says so
            // does not matter)
            if (P16 && chk16w && (ov >> 16)) beyond16((ml >> 4) * 2);
        }
    };
    auto writeout = [&](uint32_t t) __attribute__((always_inline)) {
        const uint32_t *tb = (P16 && narrow) ? otile + (t & 1) * WR_TILE_W : tiles + (t % WR_NB) * WR_TILE_W;
        const uint32_t i0 = t * WR_TS;
        const bool chk16 = P16 && ctl[3] != 0;
        if (!WIDE && whole && i0 + WR_TS <= nmax) {
            // (the output form decided outside: a lone wave pays for every branch, taken or not, with an instruction fetch nothing hides)
            if (anywaste) { if (interleave & 1) writeout_whole(t, std::true_type(), std::true_type()); else writeout_whole(t, std::false_type(), std::true_type()); }
            else if (interleave & 1) writeout_whole(t, std::true_type(), std::false_type());
            else writeout_whole(t, std::false_type(), std::false_type());
            return;
        }
        if (stereo_fast) {
            // task = (frame pair of rows, group): 32 x 16, four per lane -- always the same four frames, whose facts sit in registers
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t id = (uint32_t)k * 128 + ml;
                const uint32_t r0 = (id >> 4) * 2, g = id & 15, i = i0 + g * 4;
                const uint32_t rn = w_rn[k];
                if (i >= rn) continue;
                const uint32_t cc = w_cc[k], wa = w_wa[k], wb = w_wb[k];
                int32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
                if (w_ok[k]) {
                    const uint4 ta = *(const uint4 *)&tb[r0 * WR_TS + ((g ^ (r0 & 15)) << 2)];
                    const uint4 tb4 = *(const uint4 *)&tb[(r0 + 1) * WR_TS + ((g ^ ((r0 + 1) & 15)) << 2)];
                    const uint32_t xa[4] = {ta.x, ta.y, ta.z, ta.w}, xb[4] = {tb4.x, tb4.y, tb4.z, tb4.w};
                    if (P16) {
                        uint32_t ov = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++) if (i + e < rn) ov |= (xa[e] + 0x8000u) | (xb[e] + 0x8000u);
                        if (chk16 && (ov >> 16)) beyond16(r0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int32_t av = (int32_t)(xa[e] << wa), bv = (int32_t)(xb[e] << wb);
                        int32_t ma, mb;
                        if (WIDE) {
                            // (32-bit streams: a side channel with wasted bits is a 33-bit value once shifted back)
                            const i64 side = (i64)((u64)(i64)(int32_t)xb[e] << wb);
                            const i64 mid = (i64)(((u64)(i64)av) << 1) | (side & 1);
                            ma = (int32_t)((mid + side) >> 1); mb = (int32_t)((mid - side) >> 1);
                        }
                        else {
                            const int32_t mid = (int32_t)(((uint32_t)av << 1) | ((uint32_t)bv & 1));
                            ma = (mid + bv) >> 1; mb = (mid - bv) >> 1;
                        }
                        a[e] = cc == 2 ? av + bv : cc == 3 ? ma : av;
                        b[e] = cc == 1 ? av - bv : cc == 3 ? mb : bv;
                    }
                }
                int32_t *o = out + w_oo[k] * 2;
                // whole groups of four at a 16-byte aligned place: two 16-byte stores; the last group of an odd block (the tail of
                // nearly every real stream) and frames at odd offsets: sample by sample
                const bool vec = i + 4 <= rn && (w_oo[k] & 1) == 0 && ((interleave & 1) || (rn & 3) == 0);
                if (interleave & 1) {
                    if (vec) {
                        int4 *d = (int4 *)(o + (size_t)i * 2);
                        d[0] = make_int4(a[0], b[0], a[1], b[1]);
                        d[1] = make_int4(a[2], b[2], a[3], b[3]);
                    }
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) if (i + e < rn) *(int2 *)(o + (size_t)(i + e) * 2) = make_int2(a[e], b[e]);
                    }
                }
                else {
                    if (vec) {
                        *(int4 *)(o + i) = make_int4(a[0], a[1], a[2], a[3]);
                        *(int4 *)(o + rn + i) = make_int4(b[0], b[1], b[2], b[3]);
                    }
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) if (i + e < rn) { o[i + e] = a[e]; o[rn + i + e] = b[e]; }
                    }
                }
            }
            return;
        }
        // general form: task = (row, column)
        for (uint32_t id = ml; id < 64 * WR_TS; id += 128) {
            const uint32_t row = id >> 6, col = id & 63, i = i0 + col;
            const uint32_t *fm = fa + row * WR_FA;
            const uint32_t rn = fm[1];
            if (i >= rn) continue;
            const uint32_t cch = row % C;
            const u64 oo = ((u64)fm[5] << 32) | fm[4];
            int32_t v = 0;
            if (C == 2) {
                const uint32_t r0 = row & ~1u;
                const uint32_t *f0 = fa + r0 * WR_FA;
                if (f0[0] != 0) {
                    const uint32_t xa = tb[r0 * WR_TS + ((((col >> 2) ^ (r0 & 15)) << 2) | (col & 3))];
                    const uint32_t xb = tb[(r0 + 1) * WR_TS + ((((col >> 2) ^ ((r0 + 1) & 15)) << 2) | (col & 3))];
                    if (P16 && chk16 && (((xa + 0x8000u) | (xb + 0x8000u)) >> 16)) beyond16(r0);
                    const uint32_t cc = f0[6], wa = f0[7], wb = f0[WR_FA + 7];
                    const int32_t av = (int32_t)(xa << wa), bv = (int32_t)(xb << wb);
                    const i64 side = WIDE ? (i64)((u64)(i64)(int32_t)xb << wb) : (i64)bv;
                    const i64 mid = (i64)(((u64)(i64)av) << 1) | (side & 1);
                    const int32_t ma = (int32_t)((mid + side) >> 1), mb = (int32_t)((mid - side) >> 1);
                    const int32_t lo = cc == 2 ? av + bv : cc == 3 ? ma : av;
                    const int32_t ro = cc == 1 ? av - bv : cc == 3 ? mb : bv;
                    v = cch == 0 ? lo : ro;
                }
            }
            else if (fm[0] != 0) {
                const uint32_t xv = tb[row * WR_TS + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3))];
                if (P16 && chk16 && ((xv + 0x8000u) >> 16)) beyond16(row);
                v = (int32_t)(xv << fm[7]);
            }
            int32_t *o = out + oo * C;
            if (interleave & 1) o[(size_t)i * C + cch] = v;
            else o[(size_t)cch * rn + i] = v;
        }
    };
    // (tile T - 1 is computed in step T and leaves in step S = T + 1)
    unsigned long long tb0 = 0, tbar = 0, tall = WR_DBG ? wall_clock64() : 0;
    for (uint32_t s = 1; s <= S; s++) {
        if (WR_DBG) tb0 = wall_clock64();
        __syncthreads();
        if (WR_DBG) tbar += wall_clock64() - tb0;
        if (s >= 2 && !(WR_DBG && (interleave & 0x100))) writeout(s - 2);
    }
    if (WR_DBG && (interleave & 0x2000) && blockIdx.x == 100 && ml == 0)
        printf("restore wg 100 writer: %llu ticks, %llu at barriers\n", wall_clock64() - tall, tbar);
}

}  // namespace

// Same contract as fg_launch_decode_fast (flac_dec_fast.hip): residual plane, subframe records, parse status.
extern "C" int fg_launch_decode_wparse(const uint8_t *d_stream, uint64_t stream_len, const FgDecFrame *d_frames, uint32_t nframes,
                                       int32_t *d_scratch, FgDecSub *d_subs, FgDecResult *d_results, int wide, uint16_t *d_rparams,
                                       unsigned long long *d_counters, hipStream_t stream, int plane16, const FgDecSelf *self)
{
    if (nframes == 0) return 0;
    const dim3 grid((nframes + 3) / 4);
    FgDecSelf SF;
    if (self) SF = *self; else { SF.offsets = nullptr; SF.hdrrec = nullptr; SF.planeoff = nullptr; SF.plane_cap_bytes = 0; SF.si_bps = 0; SF.reserved = 0; }
    if (wide) hipLaunchKernelGGL((fg_dec_wparse_kernel<true, false>), grid, dim3(256), 0, stream, d_stream, (u64)stream_len, d_frames, nframes, d_scratch, d_subs, d_results, d_rparams, d_counters, SF);
    else if (plane16) hipLaunchKernelGGL((fg_dec_wparse_kernel<false, true>), grid, dim3(256), 0, stream, d_stream, (u64)stream_len, d_frames, nframes, d_scratch, d_subs, d_results, d_rparams, d_counters, SF);
    else hipLaunchKernelGGL((fg_dec_wparse_kernel<false, false>), grid, dim3(256), 0, stream, d_stream, (u64)stream_len, d_frames, nframes, d_scratch, d_subs, d_results, d_rparams, d_counters, SF);
    return (int)hipGetLastError();
}

// Same contract as fg_launch_decode_finish (flac_dec_fast.hip), without the profile words.
extern "C" int fg_launch_decode_wrestore(const FgDecFrame *d_frames, uint32_t nframes, uint32_t channels, const int32_t *d_scratch,
                                         const FgDecSub *d_subs, int32_t *d_pcm, FgDecResult *d_results, uint32_t interleave, int wide,
                                         hipStream_t stream, int plane16, FgDecResult *h_rows, const unsigned long long *d_planeoff,
                                         unsigned long long *d_join, unsigned long long epoch)
{
    if (nframes == 0) return 0;
    const uint32_t C = channels ? channels : 1;
    if (C > 64) return -1;
    if (fg_tune("FLACGPU_DEC_SKIP")) {
        // experiments: 1 no output, 2 no recurrence, 4 the writers without their stores, 8 the writers without their LDS reads
        const uint32_t v = (uint32_t)atoi(fg_tune("FLACGPU_DEC_SKIP"));
        interleave |= ((v & 7u) << 8) | ((v & 8u) ? 0x1000u : 0u) | ((v & 16u) ? 0x2000u : 0u);       // 16: where the waves of one workgroup wait
    }
    const uint32_t G = 64 / C;
    const dim3 grid((nframes + G - 1) / G);
    const size_t lds = ((size_t)WR_WORDS + 64 * WR_FA + 8) * 4;
    const void *fn = wide ? (const void *)fg_dec_wrestore_kernel<true, false>
                          : (plane16 ? (const void *)fg_dec_wrestore_kernel<false, true> : (const void *)fg_dec_wrestore_kernel<false, false>);
    if (fg_func_set_lds(fn, lds) != 0) return -1;
    if (wide) hipLaunchKernelGGL((fg_dec_wrestore_kernel<true, false>), grid, dim3(256), lds, stream, d_frames, nframes, C, d_subs, d_scratch, d_pcm, d_results, interleave, h_rows, d_planeoff, d_join, (u64)epoch);
    else if (plane16) hipLaunchKernelGGL((fg_dec_wrestore_kernel<false, true>), grid, dim3(256), lds, stream, d_frames, nframes, C, d_subs, d_scratch, d_pcm, d_results, interleave, h_rows, d_planeoff, d_join, (u64)epoch);
    else hipLaunchKernelGGL((fg_dec_wrestore_kernel<false, false>), grid, dim3(256), lds, stream, d_frames, nframes, C, d_subs, d_scratch, d_pcm, d_results, interleave, h_rows, d_planeoff, d_join, (u64)epoch);
    return (int)hipGetLastError();
}
