// flac_dec_kernels.hip -- MI355X (gfx950) FLAC frame decoder.
//
// Replaces the decode hot loop of libFLAC (read_subframe_*, read_residual_partitioned_rice_,
// FLAC__fixed_restore_signal, FLAC__lpc_restore_signal, undo_channel_coding) that pyFLAC reaches
// through FLAC__stream_decoder_process_until_end_of_stream / process_single
// (reference: pyflac/decoder.py:196,294,388; algorithm: SURVEY.md Appendix B, rows D2-D5 of 8a).
//
// Rice decoding is serial in the bit position and the LPC recurrence has an arithmetic shift inside the
// feedback (hard part 2), so the honest parallelism is one lane per frame:
//   phase 1  lane = frame: bit reader over the frame's bytes, Rice decode + predictor restore of every
//            subframe into a frame-planar scratch area (history ring in LDS, [tap][lane] layout);
//   phase 2  the wave walks its 64 frames together: CRC-16 over the frame bytes (64 lanes over interleaved
//            words), then wasted-bit/stereo undo and the coalesced store of the output samples.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fg_dev.h"
#include "fg_types.h"
#include "fg_dec_hdr.h"

typedef unsigned long long u64;
typedef long long i64;

namespace {

__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000) ? (((r << 1) ^ 0x8005) & 0xFFFF) : ((r << 1) & 0xFFFF);
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}

// Per-lane MSB-first bit reader over 32-bit big-endian words of the stream buffer.
struct BitReader {
    const uint8_t *base;    // frame start
    uint32_t pos;           // bit position from base
    uint32_t end;           // frame length in bits
    uint32_t w_idx;         // cached word index
    uint32_t w0, w1;        // cached big-endian words w_idx, w_idx+1
    bool over;

    __device__ __forceinline__ uint32_t load_be(uint32_t wi) const
    {
        // bytes beyond the frame are never needed for valid streams; clamp reads to the frame
        const uint32_t b = wi * 4;
        const uint32_t nbytes = (end + 7) >> 3;
        uint32_t v = 0;
        if (b + 4 <= nbytes) {
            v = ((uint32_t)base[b] << 24) | ((uint32_t)base[b + 1] << 16) | ((uint32_t)base[b + 2] << 8) | base[b + 3];
        }
        else {
            for (uint32_t k = 0; k < 4; k++) v = (v << 8) | ((b + k < nbytes) ? base[b + k] : 0);
        }
        return v;
    }
    __device__ __forceinline__ void init(const uint8_t *p, uint32_t start_bit, uint32_t end_bit)
    {
        base = p; pos = start_bit; end = end_bit; over = false;
        w_idx = pos >> 5;
        w0 = load_be(w_idx); w1 = load_be(w_idx + 1);
    }
    __device__ __forceinline__ uint32_t peek32()
    {
        const uint32_t wi = pos >> 5;
        if (wi != w_idx) {
            if (wi == w_idx + 1) { w0 = w1; w1 = load_be(wi + 1); }
            else { w0 = load_be(wi); w1 = load_be(wi + 1); }
            w_idx = wi;
        }
        const u64 v = ((u64)w0 << 32) | w1;
        return (uint32_t)((v << (pos & 31)) >> 32);
    }
    __device__ __forceinline__ void skip(uint32_t n)
    {
        pos += n;
        if (pos > end) over = true;
    }
    __device__ __forceinline__ uint32_t bits(uint32_t n)   // n <= 32
    {
        if (n == 0) return 0;
        const uint32_t v = peek32() >> (32 - n);
        skip(n);
        return v;
    }
    __device__ __forceinline__ int32_t sbits(uint32_t n)   // n <= 32
    {
        if (n == 0) return 0;
        const int32_t v = (int32_t)peek32() >> (32 - n);
        skip(n);
        return v;
    }
    __device__ __forceinline__ uint32_t unary()
    {
        uint32_t z = 0;
        for (;;) {
            const uint32_t p = peek32();
            if (p) { const uint32_t l = (uint32_t)__clz(p); z += l; skip(l + 1); return z; }
            z += 32; skip(32);
            if (over) return z;
        }
    }
};

// Generic (slow) decoder: any predictor order up to 32.  Used only for the frames the register-resident
// decoder below flags with status 3; frame_list maps the launch's lanes to frame indices.
__global__ void __launch_bounds__(64)
fg_decode_slow_kernel(const uint8_t *stream, const FgDecFrame *frames_all, const uint32_t *frame_list, uint32_t nframes,
                      int32_t *scratch, int32_t *out, FgDecResult *results_all, const uint16_t *crctab, uint32_t interleave)
{
    __shared__ int32_t ring[32 * 64];
    __shared__ int32_t ring_hi[32 * 64];      // bit 32 and up of the history of a 33-bit side subframe
    __shared__ uint16_t crct[768];
    __shared__ uint32_t mult[64];
    __shared__ uint8_t wasted_s[64 * 8];
    const int lane = threadIdx.x;
    for (int j = lane; j < 768; j += 64) crct[j] = crctab[j];
    mult[lane] = crctab[768 + lane];
    __syncthreads();

    const uint32_t fl = blockIdx.x * 64 + lane;
    const bool valid = fl < nframes;
    const uint32_t f = valid ? frame_list[fl] : 0;
    FgDecFrame fr;
    if (valid) fr = frames_all[f];
    else { fr.byte_off = 0; fr.out_off = 0; fr.bytes = 0; fr.n = 0; fr.hdr_bytes = 0; fr.channels = 0; fr.ca = 0; fr.bps = 0; }
    uint32_t err = 0;
    const uint32_t n = fr.n, C = fr.channels;
    int32_t *planar = scratch + fr.out_off * C;     // [C][n]

    // ---------------------------------------------------------------- phase 1: lane = frame
    if (valid && fr.bytes >= fr.hdr_bytes + 2) {
        BitReader br;
        br.init(stream + fr.byte_off, fr.hdr_bytes * 8, (fr.bytes - 2) * 8);
        for (uint32_t ch = 0; ch < C && !err; ch++) {
            uint32_t sb = fr.bps;
            if ((fr.ca == 1 && ch == 1) || (fr.ca == 2 && ch == 0) || (fr.ca == 3 && ch == 1)) sb++;
            const uint32_t hdr = br.bits(8);
            uint32_t wasted = 0;
            if (hdr & 0x80) { err = 1; break; }
            if (hdr & 1) { wasted = br.unary() + 1; if (wasted >= sb) { err = 1; break; } sb -= wasted; }
            wasted_s[lane * 8 + ch] = (uint8_t)wasted;
            if (sb > 33) { err = 1; break; }
            // 33-bit side subframe (the side channel of a 32-bit stream): values are carried in 64 bits.  Left/side and
            // right/side only need the low word (the other channel is recovered in wrapping 32-bit arithmetic and fits);
            // mid/side needs bit 32, so the stereo pair is resolved right here, sample by sample, and the frame is handed to
            // phase 2 as two independent channels.
            const bool s33 = sb == 33;
            const bool ms33 = s33 && fr.ca == 3 && ch == 1;
            const uint32_t wmid = ms33 ? wasted_s[lane * 8 + 0] : 0;
            const uint32_t t = (hdr >> 1) & 0x3F;
            int32_t *dst = planar + (size_t)ch * n;
            auto sample = [&]() -> i64 {
                if (!s33) return (i64)br.sbits(sb);
                const uint32_t sgn = br.bits(1);
                const uint32_t lo = br.bits(32);
                return (i64)(((u64)(sgn ? 0xFFFFFFFFu : 0u) << 32) | lo);
            };
            auto store = [&](uint32_t i, i64 v) {
                if (ms33) {
                    const i64 m = (i64)planar[i] << wmid;
                    const i64 mid2 = (i64)((u64)m << 1) | (v & 1);
                    planar[i] = (int32_t)((mid2 + v) >> 1);
                    dst[i] = (int32_t)((mid2 - v) >> 1);
                }
                else dst[i] = (int32_t)v;
            };
            if (t == 0) {
                const i64 v = sample();
                for (uint32_t i = 0; i < n; i++) store(i, v);
            }
            else if (t == 1) {
                for (uint32_t i = 0; i < n; i++) store(i, sample());
            }
            else if ((t >= 8 && t <= 12) || t >= 32) {
                const bool lpc = t >= 32;
                const uint32_t order = lpc ? (t & 31) + 1 : (t & 7);
                if (order > n) { err = 1; break; }
                int32_t q[32];
                i64 p1 = 0, p2 = 0, p3 = 0, p4 = 0;
                for (uint32_t i = 0; i < order; i++) {
                    const i64 v = sample();
                    store(i, v);
                    ring[(i & 31) * 64 + lane] = (int32_t)v;
                    if (s33) ring_hi[(i & 31) * 64 + lane] = (int32_t)(v >> 32);
                    p4 = p3; p3 = p2; p2 = p1; p1 = v;
                }
                uint32_t prec = 0;
                int shift = 0;
                if (lpc) {
                    prec = br.bits(4) + 1;
                    if (prec == 16) { err = 1; break; }
                    shift = br.sbits(5);
                    if (shift < 0) { err = 1; break; }
                    for (uint32_t j = 0; j < 32; j++) q[j] = (j < order) ? br.sbits(prec) : 0;
                }
                const uint32_t method = br.bits(2);
                if (method > 1) { err = 1; break; }
                const uint32_t po = br.bits(4);
                const uint32_t plen = method ? 5 : 4, esc = method ? 31 : 15;
                const uint32_t psz = n >> po;
                if ((po > 0 && ((n & ((1u << po) - 1)) != 0 || psz < order)) || (po == 0 && n < order)) { err = 1; break; }
                uint32_t left = 0, part = 0, k = 0, raw = 0;
                bool is_esc = false;
                // 64-bit accumulation is a safe universal choice
                const bool narrow = lpc && (sb + prec + (32 - __clz(order)) <= 32);
                const bool fixwide = !lpc && sb + order > 32;
                for (uint32_t i = order; i < n; i++) {
                    if (left == 0) {
                        left = (part == 0) ? (psz - order) : psz;
                        if (po == 0) left = n - order;
                        part++;
                        k = br.bits(plen);
                        is_esc = (k == esc);
                        if (is_esc) raw = br.bits(5);
                        if (left == 0) { i--; continue; }
                    }
                    left--;
                    int32_t r;
                    if (is_esc) r = br.sbits(raw);
                    else {
                        const uint32_t p = br.peek32();
                        uint32_t u;
                        const uint32_t lz = p ? (uint32_t)__clz(p) : 32;
                        if (lz + 1 + k <= 32) {
                            const uint32_t rest = (lz + 1 < 32) ? (p << (lz + 1)) : 0;
                            u = (lz << k) | (k ? (rest >> (32 - k)) : 0);
                            br.skip(lz + 1 + k);
                        }
                        else {
                            const uint32_t msb = br.unary();
                            u = (msb << k) | br.bits(k);
                        }
                        r = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
                    }
                    i64 v;
                    if (s33) {
                        if (!lpc) {
                            switch (order) {
                            case 0: v = r; break;
                            case 1: v = (i64)r + p1; break;
                            case 2: v = (i64)r + 2 * p1 - p2; break;
                            case 3: v = (i64)r + 3 * p1 - 3 * p2 + p3; break;
                            default: v = (i64)r + 4 * p1 - 6 * p2 + 4 * p3 - p4; break;
                            }
                            p4 = p3; p3 = p2; p2 = p1; p1 = v;
                        }
                        else {
                            i64 sum = 0;
                            for (uint32_t j = 0; j < order; j++) {
                                const uint32_t sl = ((i - 1 - j) & 31) * 64 + lane;
                                sum += (i64)q[j] * (i64)(((u64)(uint32_t)ring_hi[sl] << 32) | (uint32_t)ring[sl]);
                            }
                            v = (i64)r + (sum >> shift);
                            ring[(i & 31) * 64 + lane] = (int32_t)v;
                            ring_hi[(i & 31) * 64 + lane] = (int32_t)(v >> 32);
                        }
                    }
                    else if (!lpc) {
                        // two's-complement wrap-around is exact whenever the true value fits int32
                        (void)fixwide;
                        switch (order) {
                        case 0: v = r; break;
                        case 1: v = (int32_t)((uint32_t)r + (uint32_t)p1); break;
                        case 2: v = (int32_t)((uint32_t)r + 2u * (uint32_t)p1 - (uint32_t)p2); break;
                        case 3: v = (int32_t)((uint32_t)r + 3u * (uint32_t)p1 - 3u * (uint32_t)p2 + (uint32_t)p3); break;
                        default: v = (int32_t)((uint32_t)r + 4u * (uint32_t)p1 - 6u * (uint32_t)p2 + 4u * (uint32_t)p3 - (uint32_t)p4); break;
                        }
                        p4 = p3; p3 = p2; p2 = p1; p1 = (int32_t)v;
                    }
                    else if (narrow) {
                        int32_t sum = 0;
                        for (uint32_t j = 0; j < order; j++) sum += q[j] * ring[((i - 1 - j) & 31) * 64 + lane];
                        v = r + (sum >> shift);
                        ring[(i & 31) * 64 + lane] = (int32_t)v;
                    }
                    else {
                        i64 sum = 0;
                        for (uint32_t j = 0; j < order; j++) sum += (i64)q[j] * (i64)ring[((i - 1 - j) & 31) * 64 + lane];
                        v = (int32_t)((i64)r + (sum >> shift));
                        ring[(i & 31) * 64 + lane] = (int32_t)v;
                    }
                    store(i, v);
                    if (br.over) { err = 4; break; }
                }
            }
            else { err = 1; break; }
            if (br.over) err = 4;
            if (ms33) { wasted_s[lane * 8 + 0] = 0; fr.ca = 0; }      // already left / right
        }
        if (!err) {
            // zero padding to the byte boundary must end exactly at the CRC-16
            const uint32_t endbits = (br.pos + 7) & ~7u;
            const uint32_t padb = endbits - br.pos;
            if (endbits != (fr.bytes - 2) * 8) err = 4;
            if (padb && (br.peek32() >> (32 - padb)) != 0) err = 5;     // libFLAC read_zero_padding_: lost sync
        }
    }
    else if (valid) err = 1;
    __syncthreads();

    // ---------------------------------------------------------------- phase 2: the wave walks its frames together
    const uint16_t *t0 = crct, *thi = crct + 256, *tlo = crct + 512;
    for (int L = 0; L < 64; L++) {
        const uint32_t fb = __shfl(fr.bytes, L), fn = __shfl(fr.n, L), fC = __shfl(fr.channels, L), fca = __shfl(fr.ca, L);
        const uint32_t ferr = __shfl(err, L);
        if (fb == 0) continue;
        const u64 boff = ((u64)__shfl((uint32_t)(fr.byte_off >> 32), L) << 32) | __shfl((uint32_t)fr.byte_off, L);
        const u64 ooff = ((u64)__shfl((uint32_t)(fr.out_off >> 32), L) << 32) | __shfl((uint32_t)fr.out_off, L);
        const uint8_t *fp = stream + boff;
        // CRC-16 of bytes [0, fb-2): lanes take interleaved 32-bit groups (front-padded), fold with x^(32k)
        const uint32_t nbytes = fb - 2;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
        uint32_t s = 0;
        for (uint32_t t = 0; t < T; t++) {
            const int qi = (int)(t * 64 + lane) - (int)pad;
            uint32_t wv = 0;
            if (qi >= 0) {
                const uint8_t *p = fp + (size_t)qi * 4;
                wv = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
            }
            s = thi[s >> 8] ^ tlo[s & 0xFF];
            uint32_t cw = 0;
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 24)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 16)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 8)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ wv) & 0xFF];
            s ^= cw;
        }
        s = gf16_mul(s, mult[63 - lane]);
        uint32_t crc = wave_xor(s);
        for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ fp[W * 4 + b]) & 0xFF];
        const uint32_t stored = ((uint32_t)fp[nbytes] << 8) | fp[nbytes + 1];
        const bool crc_ok = (crc == stored);
        uint32_t status = ferr ? ferr : (crc_ok ? 0u : 2u);
        const uint32_t fidx = __shfl(f, L);
        if (lane == 0) { results_all[fidx].err = status; results_all[fidx].crc = crc; }
        // undo wasted bits and channel coding; bad frames are delivered as silence
        const int32_t *pl = scratch + ooff * fC;
        int32_t *o = out + ooff * fC;
        uint32_t ws[8];
        for (uint32_t c = 0; c < 8; c++) ws[c] = c < fC ? wasted_s[L * 8 + c] : 0;
        for (uint32_t i = lane; i < fn; i += 64) {
            int32_t v[8];
            for (uint32_t c = 0; c < 8; c++) v[c] = (c < fC && status == 0) ? (int32_t)((uint32_t)pl[(size_t)c * fn + i] << ws[c]) : 0;
            if (status == 0) {
                if (fca == 1) v[1] = v[0] - v[1];
                else if (fca == 2) v[0] = v[0] + v[1];
                else if (fca == 3) {
                    const i64 side = (i64)((u64)(i64)pl[(size_t)fn + i] << ws[1]);     // 33 bits in a 32-bit stream
                    const i64 mid = (i64)(((u64)(i64)v[0]) << 1) | (side & 1);
                    v[0] = (int32_t)((mid + side) >> 1);
                    v[1] = (int32_t)((mid - side) >> 1);
                }
            }
            for (uint32_t c = 0; c < 8; c++) {
                if (c < fC) {
                    if (interleave) o[(size_t)i * fC + c] = v[c];
                    else o[(size_t)c * fn + i] = v[c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- frame header parse (format.h:418-462)
// lane = frame.  offsets[f] .. offsets[f+1] delimit the frame.  Fills FgDecFrame except out_off.  (The rules are in fg_dec_hdr.h.)
// write_err: this pass also leaves the verdict in results[f].err -- for parsers that start from the frame table; the wave parser on
// its own (FgDecSelf) writes the status of every frame itself, and this pass, running beside it, must not.
__device__ __forceinline__ void fg_dec_parse_header(const uint8_t *stream, u64 stream_len, const u64 *offsets, uint32_t f, uint32_t si_channels,
                                                    uint32_t si_bps, FgDecFrame *frames, FgDecResult *results, bool write_err)
{
    // the index may come from the device (the index kernel leaves ~0 in the slot of a frame it did not find; a caller's table
    // is not looked at by the host): nothing is read through an offset that does not lie inside the stream
    // (agent-scope loads: this kernel may have been let go by a word in memory, not by an event the runtime knows of: the fork of the decode launch, flacgpu_dec_api.cpp)
    const u64 o0 = __hip_atomic_load(&offsets[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), o1 = __hip_atomic_load(&offsets[f + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool inside = o0 < stream_len && o1 <= stream_len && o1 > o0 && o1 - o0 < 0x7FFFFFFFull;
    const uint8_t *p = stream + (inside ? o0 : 0);
    const uint32_t len = inside ? (uint32_t)(o1 - o0) : 0;
    FgDecFrame fr;
    fr.byte_off = inside ? o0 : 0; fr.out_off = 0; fr.bytes = len; fr.n = 0; fr.hdr_bytes = 0; fr.channels = 0; fr.ca = 0; fr.bps = 0;
    const auto pb = [&](uint32_t i) -> uint32_t { return (uint32_t)p[i]; };
    fgdev::HdrFields h;
    uint32_t bad = fgdev::fg_dec_header_fields<true>(pb, len, h);
    if (!bad) {
        fr.bps = fgdev::fg_hdr_bps(h.bpc, si_bps);
        if (h.cac < 8) { fr.channels = h.cac + 1; fr.ca = 0; } else { fr.channels = 2; fr.ca = h.cac - 7; }
        if (si_channels && fr.channels != si_channels) bad = 1;
        fr.n = h.n; fr.hdr_bytes = h.hdr_bytes;
    }
    if (bad) { fr.n = 0; fr.channels = si_channels ? si_channels : 1; fr.bytes = 0; }
    frames[f] = fr;
    if (write_err) { results[f].err = bad ? 1 : 0; results[f].crc = 0; }      // (else: parser and CRC pass, beside this one, own the words)
}
__global__ void __launch_bounds__(256)
fg_dec_headers_kernel(const uint8_t *stream, u64 stream_len, const u64 *offsets, uint32_t nframes, uint32_t si_channels, uint32_t si_bps,
                      FgDecFrame *frames, FgDecResult *results, uint32_t write_err)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    fg_dec_parse_header(stream, stream_len, offsets, f, si_channels, si_bps, frames, results, write_err != 0);
}

// ---------------------------------------------------------------- frame index from the bytes alone
// One pass over the stream, 16 bytes per lane.  A position is a frame start candidate when it carries the sync code
// 0xFFF8 (fixed block size: the only kind libFLAC writes), a header whose fields are legal (format.h:418-462), whose
// channel count and sample size are the stream's, and whose CRC-8 is right.  The header also carries the frame NUMBER, and
// that decides where the offset goes: offsets[number - first_number].  No ordering pass, no compaction -- and the few false
// candidates a compressed stream contains by chance (about 2^-31 of all positions pass the tests above) carry a random
// 31-bit number, which falls outside the table; one that lands inside it collides with the true frame of that number and
// is counted in info[1] (the host then falls back to its serial indexer).  Frames whose slot stays empty show up as
// malformed in the header pass; the CRC-16 pass of the decoder checks every frame that was found.
// info[0]: candidates seen (count mode only: nframes == 0, nothing is filed), info[1]: unresolved collisions, info[2]: candidates with the
// variable-block-size sync code 0xFFF9 (their number is a sample number; not handled here), info[3]: highest slot filled + 1.
// (the header bytes come through `p(i)`, i < 16: the index kernel has them in registers)
template <typename Bytes>
__device__ __forceinline__ bool fg_idx_header(const Bytes &p, u64 avail, uint32_t want_channels, uint32_t want_bps, u64 *number, uint32_t *variable,
                                              const uint8_t *crc8tab)
{
    if (avail < 6 || p(0) != 0xFF || (p(1) & 0xFE) != 0xF8) return false;
    const uint32_t bsc = p(2) >> 4, src = p(2) & 15, cac = p(3) >> 4, bpc = (p(3) >> 1) & 7;
    if (bsc == 0 || src == 15 || cac > 10 || bpc == 3 || (p(3) & 1)) return false;
    const uint32_t BP[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    const uint32_t ch = cac < 8 ? cac + 1 : 2;
    if (want_channels && ch != want_channels) return false;
    if (want_bps && bpc && BP[bpc] != want_bps) return false;
    uint32_t pos = 4;
    const uint32_t x = p(pos++);
    uint32_t extra;
    u64 v;
    if (!(x & 0x80)) { extra = 0; v = x; }
    else if ((x & 0xE0) == 0xC0) { extra = 1; v = x & 0x1F; }
    else if ((x & 0xF0) == 0xE0) { extra = 2; v = x & 0x0F; }
    else if ((x & 0xF8) == 0xF0) { extra = 3; v = x & 0x07; }
    else if ((x & 0xFC) == 0xF8) { extra = 4; v = x & 0x03; }
    else if ((x & 0xFE) == 0xFC) { extra = 5; v = x & 0x01; }
    else if (x == 0xFE) { extra = 6; v = 0; }
    else return false;
    if (pos + extra + 1 > avail) return false;
    for (uint32_t i = 0; i < extra; i++) { const uint32_t c = p(pos++); if ((c & 0xC0) != 0x80) return false; v = (v << 6) | (c & 0x3F); }
    if (bsc == 6) pos += 1; else if (bsc == 7) pos += 2;
    if (src == 12) pos += 1; else if (src == 13 || src == 14) pos += 2;
    if (pos + 1 > avail) return false;
    uint32_t c8 = 0;
    for (uint32_t i = 0; i < pos; i++) c8 = crc8tab[c8 ^ p(i)];          // CRC-8, polynomial 0x07 (format.h:446-449)
    if (c8 != p(pos)) return false;
    *number = v; *variable = p(1) & 1;
    return true;
}

__global__ void __launch_bounds__(256)
fg_dec_index_kernel(const uint8_t *stream, u64 len, uint32_t channels, uint32_t bps, u64 first_number, uint32_t nframes, u64 *offsets,
                    unsigned long long *info, u64 *alt, const FgDecRange *ranges, uint32_t nranges, u64 *stamp)
{
    // (stamp: the start-of-call wall-clock stamp when this is the call's first kernel -- the tables were left empty by the last call)
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
    // groups of 16 bytes aligned in memory (one 16-byte load per lane and step); a wave walks the stream in steps of
    // gridDim.x * 4 KiB (launching one short-lived wave per KiB would be bound by the dispatch rate, not by HBM)
    const uintptr_t sa = (uintptr_t)stream;
    const u64 mis = (u64)(sa & 15);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nframes != 0) offsets[nframes] = len;      // the end of the last frame
    // CRC-8 table of the header check: what a candidate costs decides the length of this pass (one in 150 waves' steps
    // holds one, and the whole wave waits for it)
    __shared__ uint8_t crc8tab[256];
    constexpr uint32_t FG_IXQ = 512;                // candidates a workgroup can park (it sees about six)
    __shared__ uint32_t qn;
    __shared__ uint32_t qw[FG_IXQ * 6];
    if (threadIdx.x == 0) qn = 0;
    {
        uint32_t c8 = threadIdx.x;
        for (int b = 0; b < 8; b++) c8 = (c8 & 0x80) ? (((c8 << 1) ^ 0x07) & 0xFF) : ((c8 << 1) & 0xFF);
        crc8tab[threadIdx.x] = (uint8_t)c8;
    }
    __syncthreads();
    // FG_IXU groups per lane are requested before the first is looked at: one 16-byte load in flight per lane does not
    // keep HBM busy (the pass is latency-bound that way), four do
    constexpr int FG_IXU = 4;
    const u64 stride = (u64)gridDim.x * 256;
    // Phase 1 of a step, per group (inlined four times, small): the SWAR test; when any lane of the wave has a candidate the
    // wave also fetches the 16 bytes behind every group from the neighbouring lane (a header is at most 16 bytes long).
    // sync code = a 0xFF byte followed by 0xF8 / 0xF9: bytes equal to 0xFF whose successor has its top five bits set and
    // bit 1 and 2 clear (SWAR over the four words and their one-byte-shifted neighbours; 1 group in 2000 gets past this).
    // Returns the candidate positions of this lane, one bit per byte of the group; W[0..7] = the group and its successor.
    auto scan = [&](const uint32_t (&w)[5], const uint32_t (&n63)[4], uint32_t (&W)[8]) -> uint32_t {
        uint32_t hit = 0, hitw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t a = w[k], b = (w[k] >> 8) | (w[k + 1] << 24);           // b = the successor of every byte of a
            const uint32_t x = ~a, isff = (x - 0x01010101u) & ~x & 0x80808080u;    // 0x80 in every byte of a that is 0xFF (a borrow can flag a byte wrongly: the bytes decide later)
            const uint32_t y = (b & 0xFEFEFEFEu) ^ 0xF8F8F8F8u, isf8 = (y - 0x01010101u) & ~y & 0x80808080u;
            hitw[k] = isff & isf8;
            hit |= hitw[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) W[k] = 0;
        if (!__any(hit != 0)) return 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { W[k] = w[k]; W[4 + k] = (uint32_t)__shfl_down((int)w[k], 1); }
        if ((threadIdx.x & 63) == 63) { W[4] = n63[0]; W[5] = n63[1]; W[6] = n63[2]; W[7] = n63[3]; }
        uint32_t cand = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) cand |= ((((hitw[k] >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu) << (4 * k);
        return cand;
    };
    // Phase 2, once per step: the candidates of the four groups.  One copy of the header check, reached through rolled
    // loops (the group's words are picked with selects): with a copy per group and per byte position this kernel's code
    // outgrew the instruction cache, and the rarely taken candidate path then ran at the speed of instruction fetches
    // from memory -- 40 us of a 55 us pass.  Nothing here goes back to the stream: a chain of dependent byte loads per
    // candidate, the whole wave waiting, was the other half of that.
    // A candidate whose header checks out files its claim: the smallest position goes into the table, the largest (+ 1)
    // beside it, and the claims are counted (fg_dec_index_resolve_kernel picks between two, three fail).  None of the three
    // atomics returns anything: a returning one holds the whole wave for a round trip to memory, and there is one per frame.
    auto check = [&](u64 pos, const uint32_t (&Hh)[4]) {
        const auto hb = [&](uint32_t qq) -> uint32_t {
            const uint32_t wv = qq < 8 ? (qq < 4 ? Hh[0] : Hh[1]) : (qq < 12 ? Hh[2] : Hh[3]);
            return (wv >> (8 * (qq & 3))) & 0xFF;
        };
        u64 number;
        uint32_t variable;
        if (!fg_idx_header(hb, len - pos, channels, bps, &number, &variable, crc8tab)) return;
        // (thousands of atomics on one address take longer than the whole pass over the bytes: the candidates are only
        // counted when that is what the call is for)
        // sync code 0xFFF9 (variable block size: the number is a sample number) is not filed; a fixed-block-size stream
        // holds such byte sequences by chance, so they only count as evidence when nothing else is found
        if (variable) { atomicAdd(&info[2], 1ull); return; }
        if (nframes == 0) {
            // counting pass: the candidates, and the largest frame number that a stream of this length can hold (no frame is
            // shorter than 9 bytes) -- the table is sized by the larger of the two
            atomicAdd(&info[0], 1ull);
            if (number >= first_number && number - first_number < len / 9 + 1) atomicMax(&info[4], (unsigned long long)(number - first_number + 1));
            return;
        }
        u64 slot;
        if (nranges) {
            // several streams laid back to back: the stream this position lies in numbers its frames from its own first number
            // and files them from its own slot on (binary search over the streams' first bytes; this path runs once per frame)
            uint32_t lo = 0, hi = nranges - 1;
            while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (ranges[mid].byte_start <= pos) lo = mid; else hi = mid - 1; }
            const FgDecRange rg = ranges[lo];
            if (number < rg.first_number || number - rg.first_number >= rg.nframes) return;
            slot = rg.slot_base + (number - rg.first_number);
        }
        else {
            if (number < first_number) return;
            slot = number - first_number;
        }
        if (slot >= nframes) return;
        atomicMin((unsigned long long *)&offsets[slot], (unsigned long long)pos);
        atomicMax((unsigned long long *)&alt[slot], (unsigned long long)pos + 1);
        atomicAdd((uint32_t *)(alt + nframes) + slot, 1u);
    };
    // Phase 2, once per step: the candidates of the four groups are only COLLECTED -- position and the 16 bytes from it on go
    // into a queue in LDS -- and the header checks run after the pass, one candidate per lane (phase 3): in the wave that
    // finds it a candidate keeps 63 lanes waiting, and one wave step in 150 has one.  One rolled copy of this code (the
    // group's words are picked with selects): with a copy per group and per byte position the kernel outgrew the
    // instruction cache and the rarely taken path ran at the speed of instruction fetches from memory.  Nothing here goes
    // back to the stream: a chain of dependent byte loads per candidate was the other half of a 55 us pass.
    auto candidates = [&](u64 g0, const uint32_t (&cnd)[FG_IXU], const uint32_t (&Wall)[FG_IXU][8]) {
#pragma unroll 1
        for (uint32_t u = 0; u < (uint32_t)FG_IXU; u++) {
            uint32_t cand = u == 0 ? cnd[0] : u == 1 ? cnd[1] : u == 2 ? cnd[2] : cnd[3];
            if (!__any(cand != 0)) continue;
            uint32_t W[9];
#pragma unroll
            for (int k = 0; k < 8; k++) W[k] = u == 0 ? Wall[0][k] : u == 1 ? Wall[1][k] : u == 2 ? Wall[2][k] : Wall[3][k];
            W[8] = 0;
            const u64 gstart = (g0 + (u64)u * stride) * 16;
#pragma unroll 1
            while (cand) {
                const uint32_t i = (uint32_t)__ffs((int)cand) - 1;
                cand &= cand - 1;
                const uint32_t b0 = (W[i >> 2] >> (8 * (i & 3))) & 0xFF;
                const u64 o = gstart + i;
                if (b0 != 0xFF || o < mis || o + 1 >= mis + len) continue;
                const u64 pos = o - mis;
                // the 16 bytes from position i on: words q .. q + 4 of the window, funnelled by r bytes
                const uint32_t q = i >> 2, r = i & 3;
                uint32_t V[5];
#pragma unroll
                for (int j = 0; j < 5; j++) V[j] = q == 0 ? W[j] : q == 1 ? W[j + 1] : q == 2 ? W[j + 2] : W[j + 3];
                uint32_t Hh[4];
#pragma unroll
                for (int j = 0; j < 4; j++) Hh[j] = __builtin_amdgcn_alignbyte(V[j + 1], V[j], r);
                if (((Hh[0] >> 8) & 0xFE) != 0xF8) continue;
                const uint32_t e = atomicAdd(&qn, 1u);
                if (e < FG_IXQ) {
                    uint32_t *qe = &qw[e * 6];
                    qe[0] = (uint32_t)pos; qe[1] = (uint32_t)(pos >> 32); qe[2] = Hh[0]; qe[3] = Hh[1]; qe[4] = Hh[2]; qe[5] = Hh[3];
                }
                // (queue full -- 512 candidates in the 48 KB of a workgroup, a stretch that is all sync codes: counted as
                // unresolved, the call fails loudly and the host indexer takes over; a second copy of the header check
                // here would double this kernel's cold code)
                else atomicAdd(&info[1], 1ull);
            }
        }
    };
    // (a group that is not wholly inside the stream -- the first and the last one -- is read byte by byte afterwards; its
    // 16-byte load goes to a group that is, so that all loads of a step are issued back to back with no branch between them)
    const u64 gsafe = mis ? 16 : 0;
    const bool anysafe = gsafe + 16 <= mis + len;
    auto bytes_of = [&](u64 gstart, uint32_t (&d)[4]) {
        d[0] = d[1] = d[2] = d[3] = 0;
        if (gstart < mis + len && gstart + 16 > mis) {
            const uint8_t *gp = stream - mis + gstart;
#pragma unroll 1
            for (uint32_t k = 0; k < 4; k++) {              // (rolled: stream edges only, keep it out of the way)
                uint32_t v = 0;
#pragma unroll 1
                for (uint32_t i = 0; i < 4; i++) {
                    const u64 o = gstart + 4 * k + i;
                    if (o >= mis && o < mis + len) v |= (uint32_t)gp[4 * k + i] << (8 * i);
                }
                d[0] = k == 0 ? v : d[0]; d[1] = k == 1 ? v : d[1]; d[2] = k == 2 ? v : d[2]; d[3] = k == 3 ? v : d[3];
            }
        }
    };
    const bool last_lane = (threadIdx.x & 63) == 63;
    for (u64 g0 = (u64)blockIdx.x * 256 + threadIdx.x; g0 * 16 < mis + len + 16 * 64; g0 += stride * FG_IXU) {
        uint32_t w[FG_IXU][5], nx63[FG_IXU][4], Wall[FG_IXU][8], cnd[FG_IXU];
        uint4 va[FG_IXU], vb[FG_IXU];
        bool in_a[FG_IXU], in_b[FG_IXU];
#pragma unroll
        for (int u = 0; u < FG_IXU; u++) {
            const u64 gstart = (g0 + (u64)u * stride) * 16;       // offset of the group relative to the aligned base (stream - mis)
            in_a[u] = gstart >= mis && gstart + 16 <= mis + len;
            // what lies behind the group (the successor of its last byte; the rest of a header that starts in it) is in the
            // next lane's registers -- except for the last lane of the wave, which loads its own next group along with
            // this one (a load in the candidate path would hold the wave for its whole latency; the other lanes repeat
            // their own address here)
            in_b[u] = last_lane && gstart + 16 >= mis && gstart + 32 <= mis + len;
            const uint8_t *base = stream - mis;
            va[u] = anysafe ? *(const uint4 *)(base + (in_a[u] ? gstart : gsafe)) : make_uint4(0, 0, 0, 0);
            vb[u] = anysafe ? *(const uint4 *)(base + (in_b[u] ? gstart + 16 : in_a[u] ? gstart : gsafe)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < FG_IXU; u++) {
            const u64 gstart = (g0 + (u64)u * stride) * 16;
            w[u][0] = va[u].x; w[u][1] = va[u].y; w[u][2] = va[u].z; w[u][3] = va[u].w;
            if (!in_a[u]) { uint32_t d[4]; bytes_of(gstart, d); w[u][0] = d[0]; w[u][1] = d[1]; w[u][2] = d[2]; w[u][3] = d[3]; }
            nx63[u][0] = vb[u].x; nx63[u][1] = vb[u].y; nx63[u][2] = vb[u].z; nx63[u][3] = vb[u].w;
            if (last_lane && !in_b[u]) bytes_of(gstart + 16, nx63[u]);
            const uint32_t nxt = (uint32_t)__shfl_down((int)w[u][0], 1);
            w[u][4] = last_lane ? nx63[u][0] : nxt;
            cnd[u] = scan(w[u], nx63[u], Wall[u]);
        }
        if (__any((cnd[0] | cnd[1] | cnd[2] | cnd[3]) != 0)) candidates(g0, cnd, Wall);
    }
    // Phase 3: the parked candidates, one per lane
    __syncthreads();
    const uint32_t nq = qn < FG_IXQ ? qn : FG_IXQ;
    for (uint32_t e = threadIdx.x; e < nq; e += 256) {
        const uint32_t *qe = &qw[e * 6];
        const uint32_t Hh[4] = {qe[2], qe[3], qe[4], qe[5]};
        check(((u64)qe[1] << 32) | qe[0], Hh);
    }
}

// Exclusive scan of the block sizes -> out_off; totals[0] = total samples, totals[1] = max block size.  Frames that would
// end past `cap` samples are rejected here (bytes = 0), so the decode kernels can be queued before the host has seen the
// total: it finds totals[0] > cap afterwards and reports the short buffer.
// (one workgroup of 1024 threads, tiles of 8192 frames: block sizes fetched into LDS, eight neighbours per thread, one
// workgroup scan per tile)
#define FG_DSCAN_TILE 8192
#define FG_DSCAN_PER (FG_DSCAN_TILE / 1024)
// one tile: block sizes into LDS, eight neighbours per thread, one workgroup scan; returns the tile's total
__device__ __forceinline__ u64 fg_dec_scan_tile(FgDecFrame *frames, uint32_t nframes, uint32_t t0, u64 carry, u64 cap, uint32_t *sz, u64 *wtot,
                                                uint32_t &m, bool write)
{
    const uint32_t tid = threadIdx.x;
    uint32_t nb[FG_DSCAN_PER];
#pragma unroll
    for (uint32_t j = 0; j < FG_DSCAN_PER; j++) {
        const uint32_t b = t0 + j * 1024 + tid;
        nb[j] = b < nframes ? frames[b].n : 0;
    }
#pragma unroll
    for (uint32_t j = 0; j < FG_DSCAN_PER; j++) {
        m = nb[j] > m ? nb[j] : m;
        sz[j * 1024 + tid] = nb[j];
    }
    __syncthreads();
    uint32_t v[FG_DSCAN_PER];
    u64 mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < FG_DSCAN_PER; j += 4) {
        const uint4 t = *(const uint4 *)&sz[tid * FG_DSCAN_PER + j];
        v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
        mine += (u64)t.x + t.y + t.z + t.w;
    }
    u64 total;
    u64 run = carry + fgdev::block_scan_excl_u64(mine, wtot, &total);
    if (write) {
#pragma unroll
        for (uint32_t j = 0; j < FG_DSCAN_PER; j++) {
            const uint32_t b = t0 + tid * FG_DSCAN_PER + j;
            if (b < nframes) {
                if (run + v[j] > cap) { frames[b].bytes = 0; frames[b].out_off = 0; }
                else frames[b].out_off = run;
            }
            run += v[j];
        }
    }
    return total;
}

__global__ void __launch_bounds__(1024)
fg_dec_scan_kernel(FgDecFrame *frames, uint32_t nframes, u64 *totals, u64 cap)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t maxn;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_DSCAN_TILE];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) maxn = 0;
    uint32_t m = 0;
    u64 carry = 0;
    for (uint32_t t0 = 0; t0 < nframes; t0 += FG_DSCAN_TILE) {
        __syncthreads();
        carry += fg_dec_scan_tile(frames, nframes, t0, carry, cap, sz, wtot, m, true);
    }
    atomicMax(&maxn, m);
    __syncthreads();
    if (tid == 0) { totals[0] = carry; totals[1] = maxn; }
}

// More than one tile (batches of many streams): one workgroup per tile.  First every tile's total and largest block
// (tsum[t], tsum[ntiles + t]), then every workgroup adds up the totals in front of its tile and scans it; workgroup 0 also
// writes the grand total and the maximum.
__global__ void __launch_bounds__(1024)
fg_dec_scan_sums_kernel(FgDecFrame *frames, uint32_t nframes, u64 *tsum, uint32_t ntiles)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t maxn;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_DSCAN_TILE];
    if (threadIdx.x == 0) maxn = 0;
    __syncthreads();
    uint32_t m = 0;
    const u64 total = fg_dec_scan_tile(frames, nframes, blockIdx.x * FG_DSCAN_TILE, 0, 0, sz, wtot, m, false);
    atomicMax(&maxn, m);
    __syncthreads();
    if (threadIdx.x == 0) { tsum[blockIdx.x] = total; tsum[ntiles + blockIdx.x] = maxn; }
}

__global__ void __launch_bounds__(1024)
fg_dec_scan_tiles_kernel(FgDecFrame *frames, uint32_t nframes, u64 *totals, u64 cap, const u64 *tsum, uint32_t ntiles)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t maxn;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_DSCAN_TILE];
    const uint32_t tid = threadIdx.x, t = blockIdx.x;
    if (tid == 0) maxn = 0;
    u64 before = 0, all = 0;
    uint32_t mx = 0;
    for (uint32_t u = tid; u < ntiles; u += 1024) {
        const u64 x = tsum[u];
        all += x;
        if (u < t) before += x;
        const uint32_t y = (uint32_t)tsum[ntiles + u];
        mx = y > mx ? y : mx;
    }
    u64 base, grand;
    (void)fgdev::block_scan_excl_u64(before, wtot, &base);
    __syncthreads();
    (void)fgdev::block_scan_excl_u64(all, wtot, &grand);
    if (t == 0) atomicMax(&maxn, mx);
    __syncthreads();
    if (t == 0 && tid == 0) { totals[0] = grand; totals[1] = maxn; }
    uint32_t m = 0;
    (void)fg_dec_scan_tile(frames, nframes, t * FG_DSCAN_TILE, base, cap, sz, wtot, m, true);
}

// First index at which two int32 arrays differ (0xFFFFFFFFFFFFFFFF if none): the encoder's verify pass compares what the
// decoder made of the fresh frames with the PCM that went in.
__global__ void __launch_bounds__(256)
fg_compare_kernel(const int32_t *a, const int32_t *b, u64 n, unsigned long long *first)
{
    u64 best = ~(u64)0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        if (a[i] != b[i]) { best = i; break; }
    if (best != ~(u64)0) atomicMin(first, (unsigned long long)best);
}

// int32 samples -> int16 (the stream decoder's block delivery for streams of at most 16 bits per sample): eight per thread
__global__ void __launch_bounds__(256) fg_narrow16_kernel(const int32_t *in, int16_t *out, u64 n)
{
    const u64 stride = (u64)gridDim.x * 256 * 8;
    for (u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const int4 a = *(const int4 *)(in + i), b = *(const int4 *)(in + i + 4);
            uint4 o;
            o.x = ((uint32_t)a.x & 0xFFFFu) | ((uint32_t)a.y << 16); o.y = ((uint32_t)a.z & 0xFFFFu) | ((uint32_t)a.w << 16);
            o.z = ((uint32_t)b.x & 0xFFFFu) | ((uint32_t)b.y << 16); o.w = ((uint32_t)b.z & 0xFFFFu) | ((uint32_t)b.w << 16);
            *(uint4 *)(out + i) = o;
        }
        else for (u64 j = i; j < n; j++) out[j] = (int16_t)in[j];
    }
}

}  // namespace

// (both buffers 16-byte aligned)
extern "C" int fg_launch_narrow16(const int32_t *d_in, int16_t *d_out, uint64_t n, hipStream_t stream)
{
    if (n == 0) return 0;
    uint64_t nb = (n + 2047) / 2048;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(fg_narrow16_kernel, dim3((uint32_t)nb), dim3(256), 0, stream, d_in, d_out, (u64)n);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_compare(const int32_t *d_a, const int32_t *d_b, uint64_t n, unsigned long long *d_first, hipStream_t stream)
{
    if (hipMemsetAsync(d_first, 0xFF, 8, stream) != hipSuccess) return -1;
    if (n == 0) return 0;
    uint64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(fg_compare_kernel, dim3((uint32_t)nb), dim3(256), 0, stream, d_a, d_b, (u64)n, d_first);
    return (int)hipGetLastError();
}

// 64-bit words the totals array of fg_launch_dec_headers needs for `nframes` frames (totals + the scan's tile totals)
extern "C" size_t fg_dec_scan_words(uint32_t nframes) { return 4 + 2 * (size_t)((nframes + FG_DSCAN_TILE - 1) / FG_DSCAN_TILE) + 2; }

extern "C" int fg_launch_dec_headers(const uint8_t *d_stream, unsigned long long stream_len, const unsigned long long *d_offsets, uint32_t nframes,
                                     uint32_t si_channels, uint32_t si_bps, FgDecFrame *d_frames, FgDecResult *d_results,
                                     unsigned long long *d_totals, unsigned long long cap_samples, hipStream_t stream, int write_err)
{
    if (nframes == 0) return 0;
    // (the header pass inside the one-workgroup scan kernel was tried in round 4 and costs far more than the launch it saves: a header
    // is a chain of dependent byte loads, and 7032 of them want 7032 threads, not 1024 -- decode launch 0.342 -> 0.405 ms)
    hipLaunchKernelGGL(fg_dec_headers_kernel, dim3((nframes + 255) / 256), dim3(256), 0, stream, d_stream, (u64)stream_len, d_offsets, nframes,
                       si_channels, si_bps, d_frames, d_results, write_err ? 1u : 0u);
    const uint32_t ntiles = (nframes + FG_DSCAN_TILE - 1) / FG_DSCAN_TILE;
    if (ntiles == 1)
        hipLaunchKernelGGL(fg_dec_scan_kernel, dim3(1), dim3(1024), 0, stream, d_frames, nframes, d_totals, (u64)cap_samples);
    else {
        // (the tile totals lie behind the three totals: the caller sizes the array with fg_dec_scan_words())
        u64 *tsum = (u64 *)d_totals + 4;
        hipLaunchKernelGGL(fg_dec_scan_sums_kernel, dim3(ntiles), dim3(1024), 0, stream, d_frames, nframes, tsum, ntiles);
        hipLaunchKernelGGL(fg_dec_scan_tiles_kernel, dim3(ntiles), dim3(1024), 0, stream, d_frames, nframes, (u64 *)d_totals, (u64)cap_samples,
                           (const u64 *)tsum, ntiles);
    }
    return (int)hipGetLastError();
}

// A compressed stream holds, by chance, a few byte sequences that pass for a frame header (roughly one per 300 MB for 16-bit
// stereo), and their one- or two-byte frame number usually names an existing frame.  Frames lie in the stream in the order
// of their numbers, so of two claims for slot k the true one is the one between the positions of frames k-1 and k+1.
// hdrrec (when given): the packed header record (fg_dec_hdr.h) of the frame every slot ends up with -- what lets the wave parser
// start from the offsets alone (FgDecSelf); 0 for an empty or a contested slot.  The index pass has checked the header, CRC-8
// included, when it filed the claim: the fields are read again here, the length rules wait for the parser (it knows the length).
__global__ void __launch_bounds__(256)
fg_dec_index_resolve_kernel(u64 *offsets, const u64 *alt, uint32_t nframes, u64 len, unsigned long long *info, const uint8_t *stream, uint32_t *hdrrec,
                            unsigned long long *gate, unsigned long long epoch)
{
    // (gate, round 5: the fork of the decode launch without an event on the main stream -- the workgroup of this kernel that ends
    // last raises gate[0], a word in the host's pinned memory, to the call's epoch; the host, which has queued the parser behind this
    // kernel, queues the side streams' kernels when it sees it (flacgpu_dec_api.cpp).  The ticket is the upper half of info[4]; it
    // wraps to zero with the last workgroup.  What the side streams' kernels read of this kernel's work is the offsets, with
    // agent-scope loads: the index pass put them there with atomics, and the one store this kernel adds goes through to memory
    // before its workgroup takes the ticket.  The header records are the parser's, behind the kernel's end.)
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *cnt = (const uint32_t *)(alt + nframes);       // claims per slot
    // info[3] = highest slot filled + 1 (one atomic per wave)
    {
        uint32_t top = (k < nframes && offsets[k] != ~(u64)0) ? k + 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)top, o); top = t > top ? t : top; }
        if ((threadIdx.x & 63) == 0 && top) atomicMax(&info[3], (unsigned long long)top);
    }
    if (k < nframes) {
    u64 fin = offsets[k];
    bool contested = false;
    const uint32_t c = cnt[k];
    if (c >= 2) {
        const u64 a = fin, b = alt[k] - 1;                   // the smallest and the largest position claiming the slot
        if (a != b) {                                         // (one position found twice cannot happen; harmless)
            if (c > 2) { atomicAdd(&info[1], 1ull); contested = true; }
            else {
                // neighbours with a single claim (a run of contested slots is not resolved here)
                const bool pv = k == 0 || cnt[k - 1] == 1, nx = k + 1 >= nframes || cnt[k + 1] == 1;
                const u64 lo = k == 0 ? 0 : offsets[k - 1], hi = k + 1 >= nframes ? len : offsets[k + 1];
                if (!pv || !nx) { atomicAdd(&info[1], 1ull); contested = true; }
                else {
                    const bool aok = (k == 0 || a > lo) && a < hi, bok = (k == 0 || b > lo) && b < hi;
                    if (aok == bok) { atomicAdd(&info[1], 1ull); contested = true; }
                    else if (bok) {
                        // (the one store of this kernel the side streams' kernels read: through to memory, and there before the ticket)
                        __hip_atomic_store(&offsets[k], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __threadfence();
                        fin = b;
                    }
                }
            }
        }
    }
    if (hdrrec) {
        uint32_t rec = 0;
        if (!contested && fin < len) {
            const uint8_t *p = stream + fin;
            const u64 avail = len - fin;
            const auto pb = [&](uint32_t i) -> uint32_t { return i < avail ? (uint32_t)p[i] : 0u; };
            fgdev::HdrFields h;
            if (!fgdev::fg_dec_header_fields<false>(pb, 0x7FFFFFFFu, h)) rec = fgdev::fg_hdr_pack(h);
        }
        hdrrec[k] = rec;
    }
    }
    if (gate) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t *ticket = (uint32_t *)(info + 4) + 1;
            // (the last ticket: the acquire fence pairs with the fences the other workgroups' writers took in front of their
            // tickets, the system-scope release store then carries all of it to the host and to the kernels it queues)
            if (atomicInc(ticket, gridDim.x - 1) == gridDim.x - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(gate, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// offsets[0 .. nframes] empty (all ones: identity of the minimum), alt[0 .. nframes) zero (identity of the maximum), the claim
// counts behind alt zero, the four counters zero: one launch instead of four fills
// (stamp, when given: the start-of-call wall-clock stamp, see fg_signal_kernel)
__global__ void fg_dec_index_init_kernel(u64 *offsets, u64 *alt, unsigned long long *info, uint32_t nframes, u64 *stamp)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && stamp) stamp[0] = wall_clock64();
    if (k <= nframes) offsets[k] = ~(u64)0;
    if (k < nframes) { alt[k] = 0; ((uint32_t *)(alt + nframes))[k] = 0; }
    if (k < 6) info[k] = 0;          // (the four counters, the count pass's maximum, the gate's timeout word)
}
// The join of the decode launch through a word in memory instead of an event (round 5; flacgpu_dec_api.cpp): a one-thread kernel
// behind the side streams' work raises join[0] to the call's epoch, and the restore kernel looks at it before it reads what that
// work left (flac_dec_wave.hip).
// (RELEASE at agent scope: the kernels in front of this one on its stream have ended -- their writes happen-before this kernel's
// start --, and the restore kernel takes an agent-scope acquire fence behind the load that sees the epoch: release/acquire on this
// word orders everything header pass, scan and CRC pass wrote before everything the restore kernel reads of it.)
__global__ void fg_dec_raise_kernel(unsigned long long *word, unsigned long long epoch)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int fg_launch_dec_raise(unsigned long long *d_word, unsigned long long epoch, hipStream_t stream)
{
    hipLaunchKernelGGL(fg_dec_raise_kernel, dim3(1), dim3(64), 0, stream, d_word, epoch);
    return (int)hipGetLastError();
}
// One wave that does nothing for `ticks` of the 100 MHz wall clock (at most a second): what a test queues in front of the side
// streams' kernels to make the restore kernel's wait for its join word turn (flacgpu_dec_api.cpp, test-hooks build only).
__global__ void fg_dec_spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64(), lim = ticks > 100000000ull ? 100000000ull : ticks;
    while (wall_clock64() - t0 < lim) __builtin_amdgcn_s_sleep(64);
}
extern "C" int fg_launch_dec_spin(unsigned long long ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(fg_dec_spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_dec_index_init(unsigned long long *d_offsets, unsigned long long *d_alt, unsigned long long *d_info, uint32_t nframes,
                                        unsigned long long *d_stamp, hipStream_t stream)
{
    hipLaunchKernelGGL(fg_dec_index_init_kernel, dim3((nframes + 256) / 256), dim3(256), 0, stream, (u64 *)d_offsets, (u64 *)d_alt, d_info, nframes,
                       (u64 *)d_stamp);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_dec_index(const uint8_t *d_stream, unsigned long long len, uint32_t channels, uint32_t bps, unsigned long long first_number,
                                   uint32_t nframes, unsigned long long *d_offsets, unsigned long long *d_info, unsigned long long *d_alt,
                                   const FgDecRange *d_ranges, uint32_t nranges, hipStream_t stream, uint32_t *d_hdrrec, unsigned long long *d_stamp,
                                   unsigned long long *d_gate, unsigned long long epoch)
{
    if (len == 0) return 0;
    // (every lane takes four groups a step; at most 8 workgroups of 4 waves per CU -- every wave slot of the chip, once --
    // and as many steps for every workgroup: the grid is sized so that the steps come out even)
    const unsigned long long groups = (len + 15 + 15) / 16;
    const unsigned long long need = (groups + 1023) / 1024;
    const unsigned long long steps = (need + 2047) / 2048;
    unsigned long long wgs = (need + steps - 1) / steps;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(fg_dec_index_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_stream, (u64)len, channels, bps,
                       (u64)first_number, nframes, (u64 *)d_offsets, d_info, (u64 *)d_alt, d_ranges, nranges, (u64 *)d_stamp);
    if (nframes)
        hipLaunchKernelGGL(fg_dec_index_resolve_kernel, dim3((nframes + 255) / 256), dim3(256), 0, stream, (u64 *)d_offsets, (const u64 *)d_alt,
                           nframes, (u64)len, d_info, d_stream, d_hdrrec, d_gate, epoch);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_decode_slow(const uint8_t *d_stream, const FgDecFrame *d_frames, const uint32_t *d_frame_list, uint32_t nlist,
                                     int32_t *d_pcm, FgDecResult *d_results, const uint16_t *d_crctab, int32_t *d_scratch,
                                     uint32_t interleave, hipStream_t stream)
{
    if (nlist == 0) return 0;
    const uint32_t nwg = (nlist + 63) / 64;
    hipLaunchKernelGGL(fg_decode_slow_kernel, dim3(nwg), dim3(64), 0, stream, d_stream, d_frames, d_frame_list, nlist, d_scratch, d_pcm,
                       d_results, d_crctab, interleave);
    return (int)hipGetLastError();
}
