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
// The read buffer: 1024 words of 64 bits.  It is refilled when every complete word has been consumed; the refill moves the
// unconsumed tail to the front -- forgetting the last frame sync position -- and asks the client for what fits.  Since every
// byte is consumed in order, the refill points depend on the sizes the client's read callback returned and on nothing else:
// RefWindows replays them from the recorded chunk ends.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace fgref {

enum { ST_LOST_SYNC = 0, ST_BAD_HEADER = 1, ST_CRC_MISMATCH = 2, ST_UNPARSEABLE = 3 };

struct StreamFacts {                 // what STREAMINFO told the decoder (all zero without one)
    bool have = false;
    uint32_t min_blocksize = 0, max_blocksize = 0, sample_rate = 0, channels = 0, bps = 0;
    uint64_t total_samples = 0;
};

// Refill points of libFLAC's bit reader.  `chunk_end[i]`: absolute stream offset at which the i-th answer of the client's read
// callback ended (increasing); an answer that filled the request completely does not end a chunk.
struct RefWindows {
    std::vector<uint64_t> chunk_end;     // absolute offsets, increasing; the last one is the end of what has been read so far
    bool eof = false;                    // the last chunk end is the end of the stream
    // replay state
    uint64_t B = 0, E = 0;               // buffer base (multiple of 8) and end of buffered data
    size_t ci = 0;                       // chunk that serves the next read
    std::vector<uint64_t> wend;          // wend[k]: end of the buffered data after refill k (the reader refills when it needs byte wend[k]:
                                         // the bytes of an incomplete last word are looked at before the client is asked for more)
    std::vector<uint64_t> base;          // base[k]: front of the buffer after refill k (whole consumed words dropped)

    // the reader was emptied (a flush) and reads on from absolute offset `at`
    void restart(uint64_t at) { chunk_end.clear(); wend.clear(); base.clear(); ci = 0; B = E = at; }

    // replay refills until byte `p` is buffered (or nothing more can be read); returns the window index -- the refill after which
    // the reader consumed p --, -1 if p is not reachable with what has been read so far
    long window_of(uint64_t p)
    {
        while (wend.empty() || p >= wend.back()) {
            const uint64_t nB = B + ((E - B) / 8) * 8;
            const uint64_t tail = E - nB;                      // bytes kept (an incomplete word)
            const uint64_t want = 8192 - tail;
            while (ci < chunk_end.size() && chunk_end[ci] <= E) ci++;
            if (ci >= chunk_end.size()) return -1;
            const uint64_t avail = chunk_end[ci] - E;
            const uint64_t got = avail < want ? avail : want;
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

struct Bits {
    const uint8_t *d;
    uint64_t len;        // bytes available
    uint64_t pos;        // bit position
    bool eof = false;    // a read ran off the end
    bool get(uint32_t n, uint64_t *v)      // n <= 64
    {
        if (pos + n > len * 8) { eof = true; return false; }       // (the reader asks for more before it consumes any of the field)
        uint64_t x = 0;
        for (uint32_t i = 0; i < n; i++) { x = (x << 1) | ((d[(pos + i) >> 3] >> (7 - ((pos + i) & 7))) & 1); }
        pos += n; *v = x;
        return true;
    }
    bool unary(uint32_t *z)
    {
        uint32_t c = 0;
        for (;;) {
            if (pos >= len * 8) { eof = true; return false; }
            const uint32_t b = (d[pos >> 3] >> (7 - (pos & 7))) & 1;
            pos++;
            if (b) break;
            c++;
        }
        *z = c;
        return true;
    }
    bool aligned() const { return (pos & 7) == 0; }
};

inline uint8_t crc8(const uint8_t *p, size_t n)
{
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1)); }
    return c;
}
inline uint16_t crc16(const uint8_t *p, size_t n)
{
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= (uint16_t)(p[i] << 8); for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1)); }
    return c;
}

struct Header { uint32_t blocksize, channels, ca, bps; uint64_t sample_number; bool is_sample_number; uint64_t frame_number; };

// One pass of read_frame_ from the byte behind a sync code found at byte `s`.  Returns:
//   0  the frame is intact (its CRC-16 matches): *end = the byte behind it
//   1  header trouble: the search goes on from *end (no stepping back), *cached = a 0xFF that was read ahead
//   2  trouble behind the header (or the end of the data inside the frame): *end = where the reader stands (byte-aligned by the
//      search that follows); the decoder steps back to s + 2 when it still can
//   3  the data ended inside the header (*hit_eof): the decoder stops there
// and appends the error statuses it reports to `errs`.
inline int read_frame(const uint8_t *d, uint64_t len, uint64_t s, const StreamFacts &si, uint32_t fixed_blocksize, std::vector<uint32_t> &errs,
                      uint64_t *end, bool *cached, Header *hout, bool *hit_eof, RefWindows *win = nullptr, uint64_t abs0 = 0)
{
    Bits br{d, len, (s + 2) * 8};
    *cached = false; *hit_eof = false;
    uint8_t raw[16];
    uint32_t rl = 0;
    raw[rl++] = d[s]; raw[rl++] = d[s + 1];
    bool unparseable = (raw[1] & 0x02) != 0;
    uint64_t x = 0;
    bool in_header = true;
    // where the reader stood when the data ran out; inside the header read_frame_ just returns (3: no stepping back, no search)
    auto fail_eof = [&]() { *hit_eof = true; *end = br.pos / 8; return in_header ? 3 : 2; };
    for (int i = 0; i < 2; i++) {
        if (!br.get(8, &x)) return fail_eof();
        if (x == 0xFF) { *cached = true; errs.push_back(ST_BAD_HEADER); *end = br.pos / 8 - 1; return 1; }      // (the 0xFF is looked at again)
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
    case 15: errs.push_back(ST_BAD_HEADER); *end = br.pos / 8; return 1;
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
        if (!br.get(8, &x)) return fail_eof();
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
            if (!br.get(8, &x)) return fail_eof();
            raw[rl++] = (uint8_t)x;
            if (!(x & 0x80) || (x & 0x40)) { bad = true; break; }
            v = (v << 6) | (x & 0x3F);
        }
        if (bad) {
            // (the byte that broke the code is looked at again as a possible start of a sync code)
            *cached = true;
            errs.push_back(ST_BAD_HEADER);
            *end = br.pos / 8 - 1;
            return 1;
        }
        if (variable) { h.is_sample_number = true; h.sample_number = v; } else { h.is_sample_number = false; h.frame_number = v; }
    }
    if (bs_hint) {
        if (!br.get(8, &x)) return fail_eof();
        raw[rl++] = (uint8_t)x;
        uint64_t v = x;
        if (bs_hint == 7) {
            if (!br.get(8, &x)) return fail_eof();
            raw[rl++] = (uint8_t)x;
            v = (v << 8) | x;
        }
        h.blocksize = (uint32_t)v + 1;
        if (h.blocksize > 65535) {      // (a block of 65536 samples is not a valid one)
            *cached = true; errs.push_back(ST_BAD_HEADER); *end = br.pos / 8 - 1; return 1;
        }
    }
    if (sr_hint) {
        if (!br.get(8, &x)) return fail_eof();
        raw[rl++] = (uint8_t)x;
        if (sr_hint != 12) {
            if (!br.get(8, &x)) return fail_eof();
            raw[rl++] = (uint8_t)x;
        }
    }
    if (!br.get(8, &x)) return fail_eof();
    if (crc8(raw, rl) != (uint8_t)x) { errs.push_back(ST_BAD_HEADER); *end = br.pos / 8; return 1; }
    if (!h.is_sample_number) {
        if (fixed_blocksize) h.sample_number = (uint64_t)fixed_blocksize * h.frame_number;
        else if (si.have) {
            if (si.min_blocksize == si.max_blocksize) h.sample_number = (uint64_t)si.min_blocksize * h.frame_number;
            else unparseable = true;
        }
        else h.sample_number = (uint64_t)h.blocksize * h.frame_number;
    }
    if (unparseable) { errs.push_back(ST_UNPARSEABLE); *end = br.pos / 8; return 1; }
    *hout = h;
    in_header = false;

    // ---- subframes
    bool searching = false;          // an error put the decoder back into the search state
    for (uint32_t ch = 0; ch < h.channels && !searching; ch++) {
        uint32_t bps = h.bps;
        if ((h.ca == 1 && ch == 1) || (h.ca == 2 && ch == 0) || (h.ca == 3 && ch == 1)) bps++;
        if (!br.get(8, &x)) return fail_eof();
        uint32_t t = (uint32_t)x;
        const bool wasted_flag = t & 1;
        t &= 0xFE;
        if (wasted_flag) {
            uint32_t u;
            if (!br.unary(&u)) return fail_eof();
            if (u + 1 >= bps) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            bps -= u + 1;
        }
        if (t & 0x80) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
        auto residual = [&](uint32_t order) -> int {          // 0 ok, 1 error reported, 2 end of data
            if (!br.get(2, &x)) return 2;
            const uint32_t method = (uint32_t)x;
            if (method > 1) { errs.push_back(ST_UNPARSEABLE); return 1; }
            if (!br.get(4, &x)) return 2;
            const uint32_t po = (uint32_t)x;
            if ((h.blocksize >> po) < order || (h.blocksize & ((1u << po) - 1)) != 0) { errs.push_back(ST_LOST_SYNC); return 1; }
            const uint32_t plen = method ? 5 : 4, esc = method ? 31 : 15;
            const uint32_t psz = h.blocksize >> po;
            // (read_residual_partitioned_rice_: partition 0 of order 0 may be shorter than the predictor -- handled by the caller's test)
            for (uint32_t p = 0; p < (1u << po); p++) {
                if (!br.get(plen, &x)) return 2;
                const uint32_t k = (uint32_t)x;
                const uint32_t cnt = (po == 0) ? h.blocksize - order : (p == 0 ? psz - order : psz);
                if (k < esc) {
                    // FLAC__bitreader_read_rice_signed_block: a code whose unary part exceeds what a 32-bit residual can hold
                    // (UINT32_MAX >> k) ends the block with LOST_SYNC.  The block reader keeps its position in registers and
                    // writes it back only where it falls to the generic readers -- a code that touches the bytes behind the whole
                    // words of the buffer -- so after such an error the reader stands where it last did that (`stale`).
                    const uint32_t limit = 0xFFFFFFFFu >> k;
                    uint64_t stale = br.pos, cur_wend = 0, cur_W = 0;
                    for (uint32_t i = 0; i < cnt; i++) {
                        const uint64_t a = br.pos;
                        uint32_t q;
                        if (!br.unary(&q)) return 2;
                        const uint64_t au = br.pos;
                        bool precise = false;
                        if (win) {
                            const uint64_t ab = abs0 + a / 8;
                            if (ab >= cur_wend) {
                                const long w = win->window_of(ab);
                                if (w >= 0) { cur_wend = win->wend[(size_t)w]; const uint64_t b = win->base[(size_t)w]; cur_W = b + ((cur_wend - b) / 8) * 8; }
                                else { cur_wend = UINT64_MAX; cur_W = UINT64_MAX; }
                            }
                            precise = ab >= cur_W || abs0 + (au + k - 1) / 8 >= cur_W;
                        }
                        if (q > limit) { errs.push_back(ST_LOST_SYNC); br.pos = precise ? au : stale; return 1; }
                        // (the block reader takes the rest of the word it stands in before it asks for the low bits that lie
                        // behind it: the reader then stands at the end of the whole words)
                        if (k && !br.get(k, &x)) { br.pos = br.len * 8; return 2; }
                        if (precise) stale = br.pos;
                    }
                }
                else {
                    if (!br.get(5, &x)) return 2;
                    const uint32_t raw_bits = (uint32_t)x;
                    if (raw_bits) for (uint32_t i = 0; i < cnt; i++) if (!br.get(raw_bits, &x)) return 2;
                }
            }
            return 0;
        };
        if (t == 0) { if (!br.get(bps, &x)) return fail_eof(); }
        else if (t == 2) { for (uint32_t i = 0; i < h.blocksize; i++) if (!br.get(bps, &x)) return fail_eof(); }
        else if (t < 16) { errs.push_back(ST_UNPARSEABLE); searching = true; }
        else if (t <= 24) {
            const uint32_t order = (t >> 1) & 7;
            if (h.blocksize <= order) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            for (uint32_t i = 0; i < order; i++) if (!br.get(bps, &x)) return fail_eof();
            const int r = residual(order);
            if (r == 2) return fail_eof();
            if (r == 1) searching = true;
        }
        else if (t < 64) { errs.push_back(ST_UNPARSEABLE); searching = true; }
        else {
            const uint32_t order = ((t >> 1) & 31) + 1;
            if (h.blocksize <= order) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            for (uint32_t i = 0; i < order; i++) if (!br.get(bps, &x)) return fail_eof();
            if (!br.get(4, &x)) return fail_eof();
            if (x == 15) { errs.push_back(ST_LOST_SYNC); searching = true; break; }
            const uint32_t prec = (uint32_t)x + 1;
            if (!br.get(5, &x)) return fail_eof();
            if (x & 0x10) { errs.push_back(ST_LOST_SYNC); searching = true; break; }       // negative shift
            for (uint32_t i = 0; i < order; i++) if (!br.get(prec, &x)) return fail_eof();
            const int r = residual(order);
            if (r == 2) return fail_eof();
            if (r == 1) searching = true;
        }
    }
    // read_zero_padding_ runs whatever state the subframes left
    if (!br.aligned()) {
        const uint32_t nb = 8 - (uint32_t)(br.pos & 7);
        if (!br.get(nb, &x)) return fail_eof();
        if (x != 0) { errs.push_back(ST_LOST_SYNC); searching = true; }
    }
    if (!searching) {
        const uint64_t body_end = br.pos / 8;
        if (!br.get(16, &x)) return fail_eof();
        if (crc16(d + s, (size_t)(body_end - s)) == (uint16_t)x) { *end = br.pos / 8; return 0; }
        errs.push_back(ST_CRC_MISMATCH);
    }
    *end = br.pos / 8;
    return 2;
}

// The decoder's walk from byte `p` of d[0, len) (absolute stream offset of d[0]: `abs0`) until it delivers a frame again.
// `cached0`: a 0xFF was read ahead at p (the search looks at it first).  Appends the error statuses in order.  Returns the start of
// the frame that decodes again (its header in *h), or len when the data ends first (`*ended`), or UINT64_MAX when more data is
// needed to tell (not final).
struct Walker {
    const uint8_t *d = nullptr;
    uint64_t len = 0, abs0 = 0;
    bool final = false;
    StreamFacts si;
    uint32_t fixed_blocksize = 0;
    RefWindows *win = nullptr;
    uint64_t eof_front = 0;          // front of the reader's buffer once refills at the end of the stream have moved it
    bool eof_shifted = false;

    uint64_t run(uint64_t p, bool cached0, std::vector<uint32_t> &errs, Header *h, bool *ended, uint64_t *frame_end)
    {
        *ended = false;
        bool cached = cached0;
        (void)cached;
        for (;;) {
            // ---- frame_sync_: (the reader is byte-aligned here) one LOST_SYNC for the first byte that is no part of a sync code
            bool first = true;
            uint64_t s = UINT64_MAX;
            while (p < len) {
                if (d[p] == 0xFF) {
                    if (p + 1 >= len) { if (!final) return UINT64_MAX; break; }
                    if (d[p + 1] == 0xFF) { if (first) { errs.push_back(ST_LOST_SYNC); first = false; } p++; continue; }       // (the second 0xFF may start the code)
                    if ((d[p + 1] >> 1) == 0x7C) { s = p; break; }
                    // 0xFF followed by something else: both bytes are consumed
                    if (first) { errs.push_back(ST_LOST_SYNC); first = false; }
                    p += 2;
                    continue;
                }
                if (first) { errs.push_back(ST_LOST_SYNC); first = false; }
                p++;
            }
            if (s == UINT64_MAX) {
                if (!final) return UINT64_MAX;
                *ended = true;
                return len;
            }
            // ---- read_frame_
            if (!final && len - s < (1u << 20) + 65536 * 8 * 5) {
                // (a damaged frame may parse far: wait until enough data is there, or the end of the stream -- the caller pulls more)
            }
            uint64_t end = 0;
            bool c2 = false, eofhit = false;
            const size_t nerr0 = errs.size();
            const int r = read_frame(d, len, s, si, fixed_blocksize, errs, &end, &c2, h, &eofhit, win, abs0);
            if (eofhit && !final) { errs.resize(nerr0); return UINT64_MAX; }
            if (r == 3) { *ended = true; return len; }
            if (r == 0) { *frame_end = end; return s; }
            if (r == 1) { if (getenv("FG_REFWALK_DEBUG")) fprintf(stderr, "refwalk: sync %llu r=1 end=%llu nerr=%zu\n", (unsigned long long)s, (unsigned long long)end, errs.size()); p = end; continue; }
            // trouble behind the header: step back to just behind the sync code while the buffer still holds it
            // FLAC__bitreader_rewind_to_after_last_seen_framesync: back to just behind the sync code while no refill has moved the
            // buffer since -- and to the FRONT OF THE BUFFER otherwise (what the last refill kept: from the word the reader stood in
            // at that moment).  Running off the data means the reader asked its client for more: that attempt shifts the buffer too.
            uint64_t back = s + 2;
            if (win) {
                const long w0 = win->window_of(abs0 + s + 1);
                const uint64_t lastp = abs0 + (end ? end - 1 : 0);
                const long w1 = eofhit ? -1 : win->window_of(lastp);
                if (eofhit) {
                    // the refill that found no more data drops the whole words the reader has consumed: the front of the buffer
                    // moves to the word the reader stands in, and that invalidates the marker.  With nothing to drop (the reader
                    // still stands in the first word) the marker stays -- if no earlier refill has taken it.
                    const long we = win->window_of(abs0 + (len ? len - 1 : 0));
                    const uint64_t b1 = we >= 0 ? win->base[(size_t)we] : 0;
                    if (eof_front < b1) eof_front = b1;
                    const uint64_t stand = abs0 + end;
                    const uint64_t nf = stand >= eof_front ? eof_front + ((stand - eof_front) / 8) * 8 : eof_front;
                    if (nf > eof_front) { eof_front = nf; eof_shifted = true; back = nf - abs0; }
                    else if (eof_shifted || (w0 >= 0 && we >= 0 && win->base[(size_t)w0] == win->base[(size_t)we])) back = s + 2;
                    else back = eof_front - abs0;
                }
                else if (w0 >= 0 && w1 >= 0 && win->base[(size_t)w1] > win->base[(size_t)w0]) back = win->base[(size_t)w1] - abs0;
            }
            if (getenv("FG_REFWALK_DEBUG")) fprintf(stderr, "refwalk: sync %llu r=%d end=%llu eof=%d back=%llu nerr=%zu\n", (unsigned long long)s, r, (unsigned long long)end, (int)eofhit, (unsigned long long)back, errs.size());
            p = back;
        }
    }
};

}  // namespace fgref
