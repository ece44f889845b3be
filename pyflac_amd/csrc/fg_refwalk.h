// fg_refwalk.h -- what a client of libFLAC 1.4.3 observes while the decoder works its way through damaged data.
//
// The GPU decoder finds frames by structure (sync code, header CRC-8, numbering, CRC-16) and decodes them all at once; libFLAC
// reads one bit after the other, and what it reports for a damaged stretch -- which error statuses, in which order, and which
// intact frames behind the damage it loses -- follows from where that serial reader happens to stand: an error inside a frame
// leaves it at the field that tripped it, a frame whose garbage parses to some end is followed by a CRC-16 that does not
// match, and only while the damaged frame's sync code is still inside the 8 KiB read buffer can the decoder step back to
// it and search on from there (stream_decoder.c read_frame_, bitreader.c FLAC__bitreader_rewind_to_after_last_seen_framesync;
// behaviour described in /root/reference/pyflac/include/FLAC/stream_decoder.h:1451-1456).  This header restates that reader
// for the host: it is run over damaged stretches only (every frame the GPU decodes cleanly is delivered as it is), produces the
// error callbacks in libFLAC's order and says at which frame the decoder is back in step.  No samples are made here.
//
// Round 4: the reader is SIMULATED, not looked up.  libFLAC's bit reader is a buffer of 1024 words of 64 bits with a front B,
// an end E, a read position and the position of the last frame sync seen.  A read that needs more bits than the buffer holds
// refills it: the whole words in front of the read position are dropped (B moves up, and the frame-sync marker is forgotten
// if anything was dropped), then the client is asked for what fits.  After an error behind a frame header the decoder steps
// back to just behind the sync code while the marker holds and to the FRONT OF THE BUFFER otherwise -- and since stepping back
// re-reads bytes that are already buffered, whether the marker holds depends on the path the reader took, not on the stream
// position alone (round 3 looked both up in a static map of refill points and could walk in a circle: a refill 2..7 bytes
// behind a damaged frame's sync code sent it to a buffer front at or before that sync code again and again, where libFLAC --
// whose second reading of the frame needs no refill -- steps to sync + 2).  `Reader` keeps B, E, position and marker and
// performs every refill at the read that causes it; `RefWindows` only says what the client answers and gives the state the
// reader is in when a walk starts behind a run of clean frames (consumed in order: there the static replay holds).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace fgref {

enum { ST_LOST_SYNC = 0, ST_BAD_HEADER = 1, ST_CRC_MISMATCH = 2, ST_UNPARSEABLE = 3 };
static const uint64_t NONE = ~0ull;

struct StreamFacts {                 // what STREAMINFO told the decoder (all zero without one)
    bool have = false;
    uint32_t min_blocksize = 0, max_blocksize = 0, sample_rate = 0, channels = 0, bps = 0;
    uint64_t total_samples = 0;
};

// What the client's read callback answers, and the refill points of a reader that consumes everything in order.
// `chunk_end[i]`: absolute stream offset at which an answer of the client ended SHORT of the request (increasing): the client
// had nothing more at that moment, and a request of libFLAC's (at most 8 KiB) would have ended there too.  Answers that filled
// the request leave no mark; `data_end` is the end of everything read so far.
struct RefWindows {
    std::vector<uint64_t> chunk_end;
    uint64_t data_end = 0;
    bool eof = false;                    // data_end is the end of the stream
    uint64_t max_read = 0;               // != 0: a client that answers min(request, max_read) from wherever it stands (the test harness)
    // replay state (in-order consumption)
    uint64_t B = 0, E = 0;               // buffer base and end of buffered data
    std::vector<uint64_t> wend;          // wend[k]: end of the buffered data after refill k (the reader refills when it needs byte wend[k]:
                                         // the bytes of an incomplete last word are looked at before the client is asked for more)
    std::vector<uint64_t> base;          // base[k]: front of the buffer after refill k (whole consumed words dropped)

    // the reader was emptied (a flush) and reads on from absolute offset `at`
    void restart(uint64_t at) { chunk_end.clear(); wend.clear(); base.clear(); B = E = at; data_end = at; eof = false; }

    // bytes the client delivers to a request of `want` bytes made with `e` bytes of the stream handed over; *unknown: not
    // enough has been read yet to tell (and the stream has not ended)
    uint64_t answer(uint64_t e, uint64_t want, bool *unknown) const
    {
        *unknown = false;
        const uint64_t have = data_end > e ? data_end - e : 0;
        if (max_read) {
            const uint64_t w = want < max_read ? want : max_read;
            if (have >= w) return w;
            if (!eof) { *unknown = true; return 0; }
            return have;
        }
        std::vector<uint64_t>::const_iterator it = std::upper_bound(chunk_end.begin(), chunk_end.end(), e);
        if (it != chunk_end.end()) { const uint64_t a = *it - e; return a < want ? a : want; }
        if (have >= want) return want;
        if (!eof) { *unknown = true; return 0; }
        return have;
    }

    // replay refills until byte `p` is buffered (or nothing more can be read); returns the window index -- the refill after which
    // the reader consumed p --, -1 if p is not reachable with what has been read so far
    long window_of(uint64_t p)
    {
        while (wend.empty() || p >= wend.back()) {
            const uint64_t nB = B + ((E - B) / 8) * 8;
            const uint64_t tail = E - nB;                      // bytes kept (an incomplete word)
            bool unknown = false;
            const uint64_t got = answer(E, 8192 - tail, &unknown);
            if (got == 0) return -1;
            B = nB; E += got;
            base.push_back(B); wend.push_back(E);
            if (wend.size() > (1u << 26)) return -1;
        }
        // (binary search: wend is increasing)
        size_t lo = 0, hi = wend.size() - 1;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (p < wend[mid]) hi = mid; else lo = mid + 1; }
        return (long)lo;
    }
};

// libFLAC's bit reader over the bytes the decoder holds: d[0, len) at absolute offset abs0, and up to eight bytes in front of
// them (`pre`: a buffer front may lie a few bytes before the damaged frame the walk starts at).
struct Reader {
    const uint8_t *d = nullptr;
    uint64_t len = 0, abs0 = 0;
    uint8_t pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t npre = 0;
    const RefWindows *win = nullptr;
    uint64_t B = 0, E = 0;           // front and end of the buffered data (absolute byte offsets)
    uint64_t pos = 0;                // read position, in bits from the start of the stream
    uint64_t marker = NONE;          // byte offset the decoder steps back to (behind the last sync code seen), NONE: forgotten
    bool need_more = false;          // a refill could not be answered from what has been read so far
    bool at_eof = false;             // the client reported the end of the stream

    uint8_t byte(uint64_t a) const
    {
        if (a >= abs0) return a - abs0 < len ? d[a - abs0] : 0;
        const uint64_t back = abs0 - a;
        return back <= npre ? pre[npre - back] : 0;
    }
    uint32_t bit(uint64_t p) const { return (byte(p >> 3) >> (7 - (p & 7))) & 1; }
    uint64_t whole() const { return B + ((E - B) / 8) * 8; }        // end of the buffer's complete words

    // bitreader_read_from_client_: consumed whole words go (and with them the marker), then the client is asked
    bool refill()
    {
        const uint64_t cw = ((pos >> 3) - B) / 8;
        if (cw) { marker = NONE; B += cw * 8; }
        const uint64_t want = 8192 - (E - B);
        if (want == 0) { at_eof = true; return false; }        // (not reached: the read position is never 8 KiB behind the end)
        bool unknown = false;
        const uint64_t got = win->answer(E, want, &unknown);
        if (unknown) { need_more = true; return false; }
        if (got == 0) { at_eof = true; return false; }
        E += got;
        return true;
    }
    // FLAC__bitreader_read_raw_uint32 (n <= 32) / _uint64 (two reads)
    bool get32(uint32_t n, uint64_t *v)
    {
        *v = 0;
        if (n == 0) return true;
        while (E * 8 - pos < n) if (!refill()) return false;
        uint64_t x = 0;
        for (uint32_t i = 0; i < n; i++) x = (x << 1) | bit(pos + i);
        pos += n; *v = x;
        return true;
    }
    bool get(uint32_t n, uint64_t *v)
    {
        if (n <= 32) return get32(n, v);
        uint64_t hi, lo;
        if (!get32(n - 32, &hi) || !get32(32, &lo)) return false;
        *v = (hi << 32) | lo;
        return true;
    }
    // FLAC__bitreader_read_unary_unsigned: zeros are consumed as they are seen, the client is asked when the buffer is used up
    bool unary(uint64_t *z)
    {
        uint64_t c = 0;
        for (;;) {
            const uint64_t e = E * 8;
            while (pos < e) {
                if ((pos & 7) == 0 && pos + 8 <= e && byte(pos >> 3) == 0) { pos += 8; c += 8; continue; }
                const uint32_t b = bit(pos);
                pos++;
                if (b) { *z = c; return true; }
                c++;
            }
            if (!refill()) return false;
        }
    }
    // FLAC__bitreader_read_rice_signed_block: 0 done, 1 a code no 32-bit residual can hold (the caller reports LOST_SYNC; only
    // codes that lie inside the buffer's whole words are tested), 2 a read failed.  With a parameter the block reader keeps its position in registers while it works on the buffer's whole
    // words and writes it back only where it falls to the generic readers -- a unary part or low bits that reach behind the
    // whole words -- and at the end: after the error exit the reader stands where it last did that.
    int rice_block(uint32_t cnt, uint32_t k)
    {
        uint64_t q, x;
        if (k == 0) {
            for (uint32_t i = 0; i < cnt; i++) if (!unary(&q)) return 2;
            return 0;
        }
        const uint64_t limit = 0xFFFFFFFFu >> k;
        uint64_t rpos = pos;
        bool fast = (rpos >> 3) < whole();
        uint32_t i = 0;
        while (i < cnt) {
            if (fast) {
                const uint64_t wb = whole() * 8;
                uint64_t u = rpos;
                while (u < wb) {
                    if ((u & 7) == 0 && u + 8 <= wb && byte(u >> 3) == 0) { u += 8; continue; }
                    if (bit(u)) break;
                    u++;
                }
                if (u >= wb) { pos = wb; fast = false; continue; }                             // incomplete_msbs
                if (u - rpos > limit) return 1;                                               // (position not written back)
                const uint64_t au = u + 1;
                if (au + k > wb) {                                                           // incomplete_lsbs
                    pos = wb;
                    if (!get32((uint32_t)(au + k - wb), &x)) return 2;
                    i++;
                    rpos = pos; fast = (rpos >> 3) < whole();
                    continue;
                }
                rpos = au + k; i++;
                continue;
            }
            if (!unary(&q)) return 2;
            // (no limit test on this path: a code of any length goes through, as in libFLAC's process_tail)
            if (!get32(k, &x)) return 2;
            i++;
            rpos = pos; fast = (rpos >> 3) < whole();
        }
        pos = rpos;
        return 0;
    }
    bool aligned() const { return (pos & 7) == 0; }
    // FLAC__bitreader_rewind_to_after_last_seen_framesync
    void rewind() { pos = (marker != NONE ? marker : B) * 8; }
};

inline uint8_t crc8(const uint8_t *p, size_t n)
{
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1)); }
    return c;
}
inline uint16_t crc16_step(uint16_t c, uint8_t v)
{
    c ^= (uint16_t)(v << 8);
    for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
    return c;
}
inline uint16_t crc16(const uint8_t *p, size_t n)
{
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) c = crc16_step(c, p[i]);
    return c;
}

struct Header { uint32_t blocksize, channels, ca, bps; uint64_t sample_number; bool is_sample_number; uint64_t frame_number; };

// read_frame_ behind a sync code found at absolute byte `s` (the reader stands behind it).  Returns:
//   0  the frame is intact (its CRC-16 matches): the reader stands behind it
//   1  header trouble: the search goes on from where the reader stands (no stepping back; a 0xFF that was read ahead is looked at again)
//   2  trouble behind the header, or the end of the stream inside the frame's body: the decoder steps back (Reader::rewind)
//   3  the stream ended inside the header: the decoder stops there
//   4  more data is needed to tell
// and appends the error statuses it reports to `errs`.
inline int read_frame(Reader &br, uint64_t s, const StreamFacts &si, uint32_t fixed_blocksize, std::vector<uint32_t> &errs, Header *hout)
{
    uint8_t raw[16];
    uint32_t rl = 0;
    raw[rl++] = br.byte(s); raw[rl++] = br.byte(s + 1);
    bool unparseable = (raw[1] & 0x02) != 0;
    uint64_t x = 0;
    bool in_header = true;
    // inside the header read_frame_ just returns when the data runs out (3: no stepping back, no search)
    auto fail = [&]() { return br.need_more ? 4 : (in_header ? 3 : 2); };
    auto again = [&]() { br.pos -= 8; errs.push_back(ST_BAD_HEADER); return 1; };      // (the byte is looked at again as a possible start of a sync code)
    for (int i = 0; i < 2; i++) {
        if (!br.get(8, &x)) return fail();
        if (x == 0xFF) return again();
        raw[rl++] = (uint8_t)x;
    }
    Header h;
    memset(&h, 0, sizeof h);
    uint32_t bs_hint = 0, sr_hint = 0;
    switch (raw[2] >> 4) {
    case 0: unparseable = true; break;
    case 1: h.blocksize = 192; break;
    case 2: case 3: case 4: case 5: h.blocksize = 576u << ((raw[2] >> 4) - 2); break;
    case 6: case 7: bs_hint = raw[2] >> 4; break;
    default: h.blocksize = 256u << ((raw[2] >> 4) - 8); break;
    }
    switch (raw[2] & 0x0F) {
    case 0: if (!si.have) unparseable = true; break;
    case 12: case 13: case 14: sr_hint = raw[2] & 0x0F; break;
    case 15: errs.push_back(ST_BAD_HEADER); return 1;
    default: break;
    }
    {
        const uint32_t c = raw[3] >> 4;
        if (c & 8) { h.channels = 2; if ((c & 7) > 2) unparseable = true; else h.ca = (c & 7) + 1; }
        else { h.channels = c + 1; h.ca = 0; }
    }
    switch ((raw[3] & 0x0E) >> 1) {
    case 0: if (si.have) h.bps = si.bps; else unparseable = true; break;
    case 1: h.bps = 8; break; case 2: h.bps = 12; break; case 3: unparseable = true; break;
    case 4: h.bps = 16; break; case 5: h.bps = 20; break; case 6: h.bps = 24; break; default: h.bps = 32; break;
    }
    if (raw[3] & 0x01) unparseable = true;
    // the UTF-8 coded number (36 bits at most for sample numbers, 31 for frame numbers)
    const bool variable = (raw[1] & 0x01) || (si.have && si.min_blocksize != si.max_blocksize);
    {
        if (!br.get(8, &x)) return fail();
        raw[rl++] = (uint8_t)x;
        uint64_t v = 0;
        uint32_t extra = 0;
        bool bad = false;
        if (!(x & 0x80)) { v = x; }
        else if ((x & 0xC0) && !(x & 0x20)) { v = x & 0x1F; extra = 1; }
        else if ((x & 0xE0) && !(x & 0x10)) { v = x & 0x0F; extra = 2; }
        else if ((x & 0xF0) && !(x & 0x08)) { v = x & 0x07; extra = 3; }
        else if ((x & 0xF8) && !(x & 0x04)) { v = x & 0x03; extra = 4; }
        else if ((x & 0xFC) && !(x & 0x02)) { v = x & 0x01; extra = 5; }
        else if (variable && (x & 0xFE) && !(x & 0x01)) { v = 0; extra = 6; }
        else bad = true;
        for (uint32_t i = 0; i < extra && !bad; i++) {
            if (!br.get(8, &x)) return fail();
            raw[rl++] = (uint8_t)x;
            if (!(x & 0x80) || (x & 0x40)) { bad = true; break; }
            v = (v << 6) | (x & 0x3F);
        }
        if (bad) return again();
        if (variable) { h.is_sample_number = true; h.sample_number = v; } else { h.is_sample_number = false; h.frame_number = v; }
    }
    if (bs_hint) {
        if (!br.get(8, &x)) return fail();
        raw[rl++] = (uint8_t)x;
        uint64_t v = x;
        if (bs_hint == 7) {
            if (!br.get(8, &x)) return fail();
            raw[rl++] = (uint8_t)x;
            v = (v << 8) | x;
        }
        h.blocksize = (uint32_t)v + 1;
        if (h.blocksize > 65535) return again();      // (a block of 65536 samples is not a valid one)
    }
    if (sr_hint) {
        if (!br.get(8, &x)) return fail();
        raw[rl++] = (uint8_t)x;
        if (sr_hint != 12) {
            if (!br.get(8, &x)) return fail();
            raw[rl++] = (uint8_t)x;
        }
    }
    if (!br.get(8, &x)) return fail();
    if (crc8(raw, rl) != (uint8_t)x) { errs.push_back(ST_BAD_HEADER); return 1; }
    if (!h.is_sample_number) {
        if (fixed_blocksize) h.sample_number = (uint64_t)fixed_blocksize * h.frame_number;
        else if (si.have) {
            if (si.min_blocksize == si.max_blocksize) h.sample_number = (uint64_t)si.min_blocksize * h.frame_number;
            else unparseable = true;
        }
        else h.sample_number = (uint64_t)h.blocksize * h.frame_number;
    }
    if (unparseable) { errs.push_back(ST_UNPARSEABLE); return 1; }
    *hout = h;
    in_header = false;

    // ---- subframes
    bool searching = false;          // an error put the decoder back into the search state
    for (uint32_t ch = 0; ch < h.channels && !searching; ch++) {
        uint32_t bps = h.bps;
        if ((h.ca == 1 && ch == 1) || (h.ca == 2 && ch == 0) || (h.ca == 3 && ch == 1)) bps++;
        if (!br.get(8, &x)) return fail();
        uint32_t t = (uint32_t)x;
        const bool wasted_flag = t & 1;
        t &= 0xFE;
        if (wasted_flag) {
            uint64_t u;
            if (!br.unary(&u)) return fail();
            if (u + 1 >= bps) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            bps -= (uint32_t)u + 1;
        }
        if (t & 0x80) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
        auto residual = [&](uint32_t order) -> int {          // 0 ok, 1 error reported, 2 a read failed
            if (!br.get(2, &x)) return 2;
            const uint32_t method = (uint32_t)x;
            if (method > 1) { errs.push_back(ST_UNPARSEABLE); return 1; }
            if (!br.get(4, &x)) return 2;
            const uint32_t po = (uint32_t)x;
            if ((h.blocksize >> po) < order || (h.blocksize & ((1u << po) - 1)) != 0) { errs.push_back(ST_LOST_SYNC); return 1; }
            const uint32_t plen = method ? 5 : 4, esc = method ? 31 : 15;
            const uint32_t psz = h.blocksize >> po;
            for (uint32_t p = 0; p < (1u << po); p++) {
                if (!br.get(plen, &x)) return 2;
                const uint32_t k = (uint32_t)x;
                const uint32_t cnt = (po == 0) ? h.blocksize - order : (p == 0 ? psz - order : psz);
                if (k < esc) {
                    const int r = br.rice_block(cnt, k);
                    if (r == 1) { errs.push_back(ST_LOST_SYNC); return 1; }
                    if (r) return 2;
                }
                else {
                    if (!br.get(5, &x)) return 2;
                    const uint32_t raw_bits = (uint32_t)x;
                    if (raw_bits) for (uint32_t i = 0; i < cnt; i++) if (!br.get(raw_bits, &x)) return 2;
                }
            }
            return 0;
        };
        if (t == 0) { if (!br.get(bps, &x)) return fail(); }
        else if (t == 2) { for (uint32_t i = 0; i < h.blocksize; i++) if (!br.get(bps, &x)) return fail(); }
        else if (t < 16) { errs.push_back(ST_UNPARSEABLE); searching = true; }
        else if (t <= 24) {
            const uint32_t order = (t >> 1) & 7;
            if (h.blocksize <= order) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            for (uint32_t i = 0; i < order; i++) if (!br.get(bps, &x)) return fail();
            const int r = residual(order);
            if (r == 2) return fail();
            if (r == 1) searching = true;
        }
        else if (t < 64) { errs.push_back(ST_UNPARSEABLE); searching = true; }
        else {
            const uint32_t order = ((t >> 1) & 31) + 1;
            if (h.blocksize <= order) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            for (uint32_t i = 0; i < order; i++) if (!br.get(bps, &x)) return fail();
            if (!br.get(4, &x)) return fail();
            if (x == 15) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            const uint32_t prec = (uint32_t)x + 1;
            if (!br.get(5, &x)) return fail();
            if (x & 0x10) { errs.push_back(ST_LOST_SYNC); searching = true; break; }       // negative shift
            for (uint32_t i = 0; i < order; i++) if (!br.get(prec, &x)) return fail();
            const int r = residual(order);
            if (r == 2) return fail();
            if (r == 1) searching = true;
        }
    }
    // read_zero_padding_ runs whatever state the subframes left
    if (!br.aligned()) {
        const uint32_t nb = 8 - (uint32_t)(br.pos & 7);
        if (!br.get(nb, &x)) return fail();
        if (x != 0) { errs.push_back(ST_LOST_SYNC); searching = true; }
    }
    if (!searching) {
        const uint64_t body_end = br.pos / 8;
        if (!br.get(16, &x)) return fail();
        uint16_t c = 0;
        for (uint64_t a = s; a < body_end; a++) c = crc16_step(c, br.byte(a));
        if (c == (uint16_t)x) return 0;
        errs.push_back(ST_CRC_MISMATCH);
    }
    return 2;
}

// The decoder's walk from byte `p` of d[0, len) (absolute stream offset of d[0]: `abs0`) until it delivers a frame again.
// Appends the error statuses in order.  Returns the start of the frame that decodes again (its header in *h, its end in
// *frame_end), or len when the stream ends first (`*ended`), or UINT64_MAX when more data is needed to tell (not final).
struct Walker {
    const uint8_t *d = nullptr;
    uint64_t len = 0, abs0 = 0;
    uint8_t pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // the bytes in front of d[0]
    uint32_t npre = 0;
    bool final = false;
    StreamFacts si;
    uint32_t fixed_blocksize = 0;
    RefWindows *win = nullptr;
    // the reader's buffer as the last walk left it: it still holds when the next walk starts inside it (no refill since)
    bool have_state = false;
    uint64_t sB = 0, sE = 0, sPos = 0;

    uint64_t run(uint64_t p, std::vector<uint32_t> &errs, Header *h, bool *ended, uint64_t *frame_end)
    {
        *ended = false;
        Reader br;
        br.d = d; br.len = len; br.abs0 = abs0; br.npre = npre; memcpy(br.pre, pre, 8); br.win = win;
        const uint64_t pa = abs0 + p;
        if (have_state && pa >= sPos && pa <= sE) { br.B = sB; br.E = sE; br.pos = pa * 8; }
        else {
            // behind a run of clean frames, consumed in order: the static replay says where the buffer stood when the reader took
            // the bytes in front of pa -- the last two of them in one read (the frame's CRC-16)
            const uint64_t origin = win->wend.empty() ? win->B : win->base[0];
            const uint64_t back = pa - origin >= 2 ? 2 : pa - origin;
            const long w = back ? win->window_of(pa - back) : -1;
            if (w >= 0) {
                br.B = win->base[(size_t)w]; br.E = win->wend[(size_t)w]; br.pos = (pa - back) * 8;
                uint64_t x;
                if (!br.get32((uint32_t)back * 8, &x)) { br.B = br.E = pa; br.pos = pa * 8; br.need_more = br.at_eof = false; }
            }
            else { br.B = br.E = pa; br.pos = pa * 8; }
        }
        br.marker = NONE;
        uint64_t last_s = NONE, same = 0;
        for (;;) {
            // ---- frame_sync_: one LOST_SYNC for the first byte that is no part of a sync code
            if (!br.aligned()) br.pos = (br.pos + 7) & ~7ull;
            bool first = true;
            uint64_t s = NONE, x = 0;
            for (;;) {
                if (!br.get32(8, &x)) break;
                if (x == 0xFF) {
                    const uint64_t at = br.pos / 8 - 1;
                    if (!br.get32(8, &x)) break;
                    if (x == 0xFF) br.pos -= 8;                            // (the second 0xFF may start the code)
                    else if ((x >> 1) == 0x7C) { s = at; br.marker = br.pos / 8; break; }
                }
                if (first) { errs.push_back(ST_LOST_SYNC); first = false; }
            }
            if (s == NONE) {
                if (br.need_more && !final) return UINT64_MAX;
                *ended = true;
                return len;
            }
            // (a walk that comes back to the same sync code in the same state would never end; libFLAC's does not, this is a guard)
            if (s == last_s) { if (++same > 64) { br.marker = NONE; br.B = s + 2; } } else { last_s = s; same = 0; }
            // ---- read_frame_
            const int r = read_frame(br, s, si, fixed_blocksize, errs, h);
            if (fg_tune("FG_REFWALK_DEBUG"))
                fprintf(stderr, "refwalk: sync %llu r=%d stands %llu.%u B=%llu E=%llu marker=%lld eof=%d nerr=%zu\n", (unsigned long long)s, r,
                        (unsigned long long)(br.pos / 8), (unsigned)(br.pos & 7), (unsigned long long)br.B, (unsigned long long)br.E, (long long)br.marker, (int)br.at_eof, errs.size());
            if (r == 4) { if (!final) return UINT64_MAX; *ended = true; return len; }
            if (r == 3) { *ended = true; return len; }
            if (r == 0 && s >= abs0) {
                *frame_end = br.pos / 8 - abs0;
                have_state = true; sB = br.B; sE = br.E; sPos = br.pos / 8;
                return s - abs0;
            }
            if (r == 1) continue;
            // trouble behind the header: back to just behind the sync code while no refill has moved the buffer since, to the
            // front of the buffer otherwise
            br.rewind();
        }
    }
};

}  // namespace fgref
