// flacgpu_dec_api.cpp -- the libFLAC stream-decoder entry points pyFLAC binds
// (reference: pyflac/builder/decoder.py:387-475), backed by the HIP frame decoder, plus the batch decode
// entry points of include/flacgpu.h.
//
// Host responsibilities: pulling bytes through the read callback,
// metadata parsing, locating frame boundaries (sync code + header CRC-8, confirmed by the frame CRC-16),
// delivering frames in order through the write callback and reporting errors.  All subframe decoding, CRC-16
// checking and channel reconstruction runs on the GPU; there is no CPU decode fallback.
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <time.h>
#include <thread>

#include <algorithm>

#include "fg_host.h"
#include "fg_refwalk.h"

extern "C" {
const char *const FLAC__StreamDecoderStateString[] = {
    "FLAC__STREAM_DECODER_SEARCH_FOR_METADATA", "FLAC__STREAM_DECODER_READ_METADATA", "FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC",
    "FLAC__STREAM_DECODER_READ_FRAME", "FLAC__STREAM_DECODER_END_OF_STREAM", "FLAC__STREAM_DECODER_OGG_ERROR",
    "FLAC__STREAM_DECODER_SEEK_ERROR", "FLAC__STREAM_DECODER_ABORTED", "FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR",
    "FLAC__STREAM_DECODER_UNINITIALIZED"};
const char *const FLAC__StreamDecoderInitStatusString[] = {
    "FLAC__STREAM_DECODER_INIT_STATUS_OK", "FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER",
    "FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS", "FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR",
    "FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE", "FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED"};
const char *const FLAC__StreamDecoderErrorStatusString[] = {
    "FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC", "FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER",
    "FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH", "FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM",
    "FLAC__STREAM_DECODER_ERROR_STATUS_BAD_METADATA"};
}

#define HIPOK(call) ((call) == hipSuccess)

// ------------------------------------------------------------------ frame header parsing on the host
namespace {

struct HostHeader {
    uint32_t n, sample_rate, channels, ca, bps, hdr_bytes, variable;
    uint64_t number;
};

// Returns true when p[0..avail) starts with a syntactically valid frame header whose CRC-8 matches.
bool parse_header(const uint8_t *p, size_t avail, const FLAC__StreamMetadata_StreamInfo *si, HostHeader *h)
{
    if (avail < 6 || p[0] != 0xFF || (p[1] & 0xFE) != 0xF8) return false;
    const uint32_t bsc = p[2] >> 4, src = p[2] & 15, cac = p[3] >> 4, bpc = (p[3] >> 1) & 7;
    if (bsc == 0 || src == 15 || cac > 10 || bpc == 3 || (p[3] & 1)) return false;
    size_t pos = 4;
    uint32_t x = p[pos++], extra;
    uint64_t num;
    if (!(x & 0x80)) { num = x; extra = 0; }
    else if ((x & 0xE0) == 0xC0) { num = x & 0x1F; extra = 1; }
    else if ((x & 0xF0) == 0xE0) { num = x & 0x0F; extra = 2; }
    else if ((x & 0xF8) == 0xF0) { num = x & 0x07; extra = 3; }
    else if ((x & 0xFC) == 0xF8) { num = x & 0x03; extra = 4; }
    else if ((x & 0xFE) == 0xFC) { num = x & 0x01; extra = 5; }
    else if (x == 0xFE) { num = 0; extra = 6; }
    else return false;
    if (pos + extra + 5 > avail + 0 && pos + extra >= avail) return false;
    for (uint32_t i = 0; i < extra; i++) {
        if (pos >= avail || (p[pos] & 0xC0) != 0x80) return false;
        num = (num << 6) | (p[pos++] & 0x3F);
    }
    uint32_t n;
    switch (bsc) {
    case 1: n = 192; break;
    case 2: case 3: case 4: case 5: n = 576u << (bsc - 2); break;
    case 6: if (pos + 1 > avail) return false; n = (uint32_t)p[pos] + 1; pos += 1; break;
    case 7: if (pos + 2 > avail) return false; n = (((uint32_t)p[pos] << 8) | p[pos + 1]) + 1; pos += 2; break;
    default: n = 256u << (bsc - 8); break;
    }
    static const uint32_t SR[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    uint32_t sr = si ? si->sample_rate : 0;
    if (src >= 1 && src <= 11) sr = SR[src];
    else if (src == 12) { if (pos + 1 > avail) return false; sr = (uint32_t)p[pos] * 1000; pos += 1; }
    else if (src == 13) { if (pos + 2 > avail) return false; sr = ((uint32_t)p[pos] << 8) | p[pos + 1]; pos += 2; }
    else if (src == 14) { if (pos + 2 > avail) return false; sr = (((uint32_t)p[pos] << 8) | p[pos + 1]) * 10; pos += 2; }
    if (pos + 1 > avail) return false;
    if (fg_crc8(p, pos) != p[pos]) return false;
    pos++;
    static const uint32_t BP[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    h->n = n; h->sample_rate = sr; h->bps = bpc ? BP[bpc] : (si ? si->bits_per_sample : 0);
    if (cac < 8) { h->channels = cac + 1; h->ca = 0; } else { h->channels = 2; h->ca = cac - 7; }
    h->hdr_bytes = (uint32_t)pos; h->variable = p[1] & 1; h->number = num;
    if (si && si->channels && h->channels != si->channels) return false;
    if (si && si->bits_per_sample && h->bps != si->bits_per_sample) return false;
    return true;
}

// Parse "fLaC" + metadata blocks.  Returns bytes consumed (audio offset), 0 if more data is needed,
// -1 on a malformed stream.
int64_t parse_metadata(const uint8_t *d, uint64_t len, FLAC__StreamMetadata_StreamInfo *si, bool *have_si)
{
    if (len < 4) return 0;
    if (memcmp(d, "fLaC", 4)) return -1;
    uint64_t pos = 4;
    for (;;) {
        if (pos + 4 > len) return 0;
        const uint32_t last = d[pos] >> 7, type = d[pos] & 0x7F;
        const uint32_t l = ((uint32_t)d[pos + 1] << 16) | ((uint32_t)d[pos + 2] << 8) | d[pos + 3];
        if (pos + 4 + l > len) return 0;
        if (type == 0 && l >= 34) {
            const uint8_t *s = d + pos + 4;
            si->min_blocksize = (s[0] << 8) | s[1]; si->max_blocksize = (s[2] << 8) | s[3];
            si->min_framesize = (s[4] << 16) | (s[5] << 8) | s[6];
            si->max_framesize = (s[7] << 16) | (s[8] << 8) | s[9];
            si->sample_rate = ((uint32_t)s[10] << 12) | ((uint32_t)s[11] << 4) | (s[12] >> 4);
            si->channels = ((s[12] >> 1) & 7) + 1;
            si->bits_per_sample = ((((uint32_t)s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si->total_samples = ((uint64_t)(s[13] & 15) << 32) | ((uint64_t)s[14] << 24) | ((uint64_t)s[15] << 16) |
                                ((uint64_t)s[16] << 8) | s[17];
            memcpy(si->md5sum, s + 18, 16);
            *have_si = true;
        }
        pos += 4 + l;
        if (last) break;
    }
    return (int64_t)pos;
}

// Incremental frame indexer over a growing byte buffer.  A frame [a, b) is accepted when a valid header
// starts at a, and either a valid header starts at b or b is the end of the stream, and the CRC-16 over
// [a, b) (stored CRC included) is zero.
struct Indexer {
    uint64_t frame_start = 0;     // start of the frame being delimited (valid header already confirmed)
    bool in_frame = false;
    uint64_t scan = 0;            // next byte to examine
    uint16_t crc = 0;             // running CRC-16 over [frame_start, crc_pos)
    uint64_t crc_pos = 0;         // the CRC is brought up to date only where a sync code shows up (crc_to)
    uint64_t max_len = 0;         // longest a frame with the current header can be (anything longer is damaged)
    uint64_t cur_number = 0;      // frame / sample number of the frame being delimited
    uint32_t cur_n = 0, cur_variable = 0;
    bool tail_checked = false;    // the resync search already ran over the final, damaged frame
    // Fast mode (clean streams): a frame boundary is accepted where a valid header CONTINUES THE NUMBERING, without the
    // running CRC-16 (the GPU pass checks the CRC-16 of every frame anyway).  Anything irregular -- a frame longer than its
    // header allows, a frame the GPU pass rejects -- makes the decoder restore its snapshot and repeat the round in careful
    // mode (CRC-16 at every candidate, the resynchronisation rules below), so damaged streams behave exactly as before.
    bool fast = true, fast_failed = false;
    std::vector<uint64_t> bounds; // accepted frame boundaries: frames are [bounds[i], bounds[i+1])
    std::vector<uint32_t> errors; // FLAC__StreamDecoderErrorStatus to report, in stream order
    std::vector<uint64_t> error_pos;

    // CRC-16 (x^16 + x^15 + x^2 + 1, MSB first), eight bytes per step: T[k][x] = CRC of byte x followed by k zero bytes
    static const uint16_t (*tables())[256]
    {
        static uint16_t T[8][256];
        static bool ok = false;
        if (!ok) {
            for (int i = 0; i < 256; i++) {
                uint16_t c = (uint16_t)(i << 8);
                for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
                T[0][i] = c;
            }
            for (int k = 1; k < 8; k++)
                for (int i = 0; i < 256; i++) T[k][i] = (uint16_t)((T[k - 1][i] << 8) ^ T[0][T[k - 1][i] >> 8]);
            ok = true;
        }
        return T;
    }
    void crc_to(const uint8_t *d, uint64_t upto)
    {
        const uint16_t (*T)[256] = tables();
        uint16_t c = crc;
        uint64_t p = crc_pos;
        while (p + 8 <= upto) {
            const uint8_t *b = d + p;
            c = (uint16_t)(T[7][(uint8_t)((c >> 8) ^ b[0])] ^ T[6][(uint8_t)(c ^ b[1])] ^ T[5][b[2]] ^ T[4][b[3]] ^ T[3][b[4]] ^
                           T[2][b[5]] ^ T[1][b[6]] ^ T[0][b[7]]);
            p += 8;
        }
        for (; p < upto; p++) c = (uint16_t)((c << 8) ^ T[0][(c >> 8) ^ d[p]]);
        crc = c; crc_pos = upto;
    }
    void open_frame(const HostHeader &h)
    {
        // verbatim subframes (one extra bit for a side channel) + subframe headers with a long wasted-bits unary + footer
        const uint64_t bps = h.bps ? h.bps : 32;
        max_len = h.hdr_bytes + (uint64_t)h.channels * (((uint64_t)h.n * (bps + 1) + 7) / 8 + 8) + 2 + 16;
        cur_number = h.number; cur_n = h.n; cur_variable = h.variable;
    }
    // Examine data[0..len).  `final` = no more data will arrive.
    void feed(const uint8_t *d, uint64_t len, bool final, const FLAC__StreamMetadata_StreamInfo *si)
    {
        static uint16_t tab[256];
        static bool tab_ok = false;
        if (!tab_ok) {
            for (int i = 0; i < 256; i++) {
                uint16_t c = (uint16_t)(i << 8);
                for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
                tab[i] = c;
            }
            tab_ok = true;
        }
        HostHeader h;
        for (;;) {
            // A frame longer than its header allows, or one that reaches the end of the data without a clean CRC-16,
            // is damaged: resynchronise (below).
            const bool at_end = scan >= len;
            if (at_end && final && in_frame && !tail_checked && !fast) crc_to(d, len);
            // (tail_checked: the search below already ran to the end of the final data for this frame and found nothing --
            // without it a last frame longer than max_len asked for the same search forever)
            const bool want_resync = in_frame && !tail_checked && ((scan - frame_start > max_len) || (at_end && final && !fast && crc != 0));
            if (want_resync && fast) { fast_failed = true; return; }
            if (at_end && !want_resync) break;
            if (!in_frame) {
                // search for a frame start
                if (d[scan] == 0xFF && scan + 1 < len && (d[scan + 1] & 0xFE) == 0xF8) {
                    if (len - scan < 16 && !final) return;             // need the whole header
                    if (parse_header(d + scan, len - scan, si, &h)) {
                        in_frame = true; frame_start = scan; crc = 0; crc_pos = scan;
                        open_frame(h);
                        if (bounds.empty() || bounds.back() != scan) {
                            bounds.push_back(scan);
                        }
                        scan += h.hdr_bytes;
                        continue;
                    }
                }
                else if (d[scan] == 0xFF && scan + 1 >= len && !final) return;
                if (errors.empty() || error_pos.back() + 1 != scan || errors.back() != FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC) {
                    errors.push_back(FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC);
                    error_pos.push_back(scan);
                }
                else error_pos.back() = scan;
                scan++;
                continue;
            }
            // inside a frame: a boundary is a position where a valid header starts and the CRC-16 of everything since the
            // frame start is zero.  Headers start with 0xFF: jump from one 0xFF to the next (memchr) and bring the CRC up to
            // date only where the sync code is complete.
            if (!at_end && !want_resync) {
                const uint64_t lim = std::min<uint64_t>(len, frame_start + max_len + 1);
                const uint8_t *q = scan < lim ? (const uint8_t *)memchr(d + scan, 0xFF, (size_t)(lim - scan)) : nullptr;
                if (!q) { scan = lim; continue; }
                scan = (uint64_t)(q - d);
                if (scan + 1 >= len) {
                    if (!final) return;
                    scan++;
                    continue;
                }
                if ((d[scan + 1] & 0xFE) == 0xF8 && scan >= frame_start + 9) {
                    if (len - scan < 16 && !final) return;
                    bool boundary;
                    if (fast) {
                        const uint64_t step = cur_variable ? cur_n : 1;
                        boundary = parse_header(d + scan, len - scan, si, &h) && h.variable == cur_variable && h.number == cur_number + step;
                    }
                    else {
                        crc_to(d, scan);
                        boundary = crc == 0 && parse_header(d + scan, len - scan, si, &h);
                    }
                    if (boundary) {
                        bounds.push_back(scan);       // closes the current frame, opens the next
                        frame_start = scan; crc = 0; crc_pos = scan;
                        open_frame(h);
                        scan += h.hdr_bytes;
                        continue;
                    }
                }
                scan++;
                continue;
            }
            // Resynchronisation: no position with a clean CRC-16 inside the longest possible frame means the frame is
            // damaged.  It is cut at the next header that continues the numbering (the GPU pass then reports the CRC
            // mismatch and delivers silence for it, like libFLAC) instead of swallowing the frames behind it.
            if (want_resync) {
                uint64_t p2 = frame_start + 2;
                bool found = false, starved = false;
                for (; p2 + 1 < len; p2++) {
                    if (d[p2] != 0xFF || (d[p2 + 1] & 0xFE) != 0xF8) continue;
                    if (len - p2 < 16 && !final) { starved = true; break; }
                    if (!parse_header(d + p2, len - p2, si, &h)) continue;
                    const uint64_t step = cur_variable ? cur_n : 1;
                    if (h.variable == cur_variable && h.number >= cur_number + step && h.number <= cur_number + 64 * step) { found = true; break; }
                }
                if (starved || (!found && !final)) return;          // wait for more data (scan stays, the test repeats)
                if (found) {
                    // The damage may sit in the NEXT frame's header only, with this frame intact: look for the end of
                    // an intact frame (running CRC-16 zero) before p2, preferring one followed by a sync code.  The
                    // bytes between it and p2 become a frame of their own, which the GPU pass rejects (bad header).
                    // A wrong guess costs nothing: the parse kernel rejects a frame whose contents do not end at its
                    // boundary.
                    uint16_t c2 = 0;
                    uint64_t cut_sync = 0, cut_any = 0;
                    for (uint64_t e = frame_start; e < p2; e++) {
                        c2 = (uint16_t)((c2 << 8) ^ tab[(c2 >> 8) ^ d[e]]);
                        if (c2 == 0 && e + 1 >= frame_start + 9 && e + 1 < p2 && e + 1 - frame_start <= max_len) {
                            cut_any = e + 1;
                            if (!cut_sync && d[e + 1] == 0xFF && e + 2 < len && (d[e + 2] & 0xFC) == 0xF8) cut_sync = e + 1;
                        }
                    }
                    // (c2 == 0 here: the whole stretch up to the next header is ONE intact frame that is merely longer than
                    // any libFLAC encoder would make it -- escape-coded partitions with more raw bits than the sample size,
                    // Rice codes longer than verbatim samples; a zero of the running CRC inside it is then a coincidence,
                    // one per 64 KiB)
                    const uint64_t cut = c2 == 0 ? 0 : (cut_sync ? cut_sync : cut_any);
                    if (cut) bounds.push_back(cut);
                    bounds.push_back(p2);
                    frame_start = p2; crc = 0; crc_pos = p2; open_frame(h);
                    scan = p2 + h.hdr_bytes;
                    continue;
                }
                tail_checked = true;   // final and nothing follows: the damaged frame runs to the end of the data
                scan = len;
                continue;
            }
        }
        if (final && in_frame) {
            // last frame ends at the end of the data (its CRC is checked on the GPU like every other frame)
            bounds.push_back(len);
            in_frame = false;
        }
    }
};

}  // namespace

extern "C" int flacgpu_decode_frames_dev(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, const uint64_t *d_frame_offsets,
                                         uint32_t nframes, uint32_t channels_hint, uint32_t bps_hint, void *d_pcm,
                                         uint64_t pcm_capacity_samples, void *h_frame_status, flacgpu_decode_stats *stats);

// Test entry of fg_refwalk.h (no device involved): the error statuses and the frames libFLAC 1.4.3 would deliver for a whole
// stream read in answers of `read_size` bytes.  frames[2 i] = sample number, frames[2 i + 1] = block size of the i-th frame
// that decodes; returns the number of errors (negative: not a FLAC stream), *nframes the number of frames.
extern "C" int64_t flacgpu_refwalk_probe(const uint8_t *stream, uint64_t len, uint32_t read_size, uint32_t *errors, uint64_t errors_cap,
                                         uint64_t *frames, uint64_t frames_cap, uint64_t *nframes)
{
    FLAC__StreamMetadata_StreamInfo si;
    memset(&si, 0, sizeof si);
    bool have = false;
    const int64_t a = parse_metadata(stream, len, &si, &have);
    if (a <= 0) return -1;
    fgref::RefWindows win;
    win.max_read = read_size ? read_size : 1;      // (the test harness answers min(request, read_size) from wherever it stands)
    win.data_end = len;
    win.eof = true;
    fgref::Walker w;
    w.d = stream; w.len = len; w.abs0 = 0; w.final = true; w.win = &win;
    w.si.have = have; w.si.min_blocksize = si.min_blocksize; w.si.max_blocksize = si.max_blocksize; w.si.sample_rate = si.sample_rate;
    w.si.channels = si.channels; w.si.bps = si.bits_per_sample; w.si.total_samples = si.total_samples;
    std::vector<uint32_t> errs;
    uint64_t p = (uint64_t)a, nf = 0;
    for (;;) {
        fgref::Header h;
        bool ended = false;
        uint64_t fend = 0;
        const uint64_t s = w.run(p, errs, &h, &ended, &fend);
        if (ended || s >= len) break;
        if (nf < frames_cap / 2) { frames[2 * nf] = h.sample_number; frames[2 * nf + 1] = h.blocksize; }
        nf++;
        if (!w.fixed_blocksize && !h.is_sample_number) w.fixed_blocksize = (have && si.min_blocksize == si.max_blocksize) ? si.min_blocksize : h.blocksize;
        p = fend;
    }
    *nframes = nf;
    for (size_t i = 0; i < errs.size() && i < errors_cap; i++) errors[i] = errs[i];
    return (int64_t)errs.size();
}

extern "C" int64_t flacgpu_index_frames(const uint8_t *stream, uint64_t len, uint64_t *frame_offsets, uint64_t capacity,
                                        FLAC__StreamMetadata_StreamInfo *streaminfo, uint64_t *audio_offset)
{
    FLAC__StreamMetadata_StreamInfo si;
    memset(&si, 0, sizeof si);
    bool have = false;
    const int64_t a = parse_metadata(stream, len, &si, &have);
    if (a <= 0) return -1;
    if (streaminfo) *streaminfo = si;
    if (audio_offset) *audio_offset = (uint64_t)a;
    Indexer ix;
    ix.fast = false;          // (no GPU pass behind this entry point: every boundary is confirmed by the frame's CRC-16)
    ix.feed(stream + a, len - a, true, have ? &si : nullptr);
    const uint64_t nfr = ix.bounds.size() > 0 ? ix.bounds.size() - 1 : 0;
    if (frame_offsets) {
        if (ix.bounds.size() > capacity) return -2;
        for (size_t i = 0; i < ix.bounds.size(); i++) frame_offsets[i] = ix.bounds[i] + (uint64_t)a;
    }
    return (int64_t)nfr;
}

// ------------------------------------------------------------------ batch decode of device-resident frames
// What the stream decoder needs to fill FLAC__Frame.subframes[] (format.h:285-396): the parse kernel's per-subframe records,
// the Rice parameters, the warm-up samples; with level 2 also the whole residual plane (frame-planar, warm-up samples in
// place) for the `residual` / verbatim `data` pointers.
struct DecDetail {
    int level = 1;
    std::vector<FgDecSub> subs;
    std::vector<uint16_t> rparams;      // FG_DEC_RPARAMS per subframe
    std::vector<int32_t> warm;          // 32 per subframe
    std::vector<int32_t> planes;        // level 2: [frame][channel][n] as the kernels keep it
};
#define FG_DEC_RPARAMS 256

// h_offsets == nullptr: the frame index is made on the device from the bytes (fg_dec_index_kernel); nframes is then the
// number of frames the stream is known to hold (STREAMINFO: total samples / block size), or 0 to have them counted first.
#ifndef FG_NO_DEFER_INIT
#define FG_NO_DEFER_INIT 0
#endif
// One attempt.  *again = true (with false returned): nothing is wrong with the input, the call is to be made once more under what this
// attempt left in the context -- events instead of the fork / join words (a wait on one of them ran out), or 32-bit residual planes
// (a value beyond 16 bits showed up); decode_frames_impl below does that, in a loop.
static bool decode_frames_once(flacgpu_ctx *c, const void *d_stream, uint64_t len, const uint64_t *h_offsets, uint32_t nframes,
                               uint32_t channels_hint, uint32_t bps_hint, void *d_pcm, uint64_t cap_samples, int interleave,
                               FgDecResult *h_status, std::vector<FgDecFrame> *h_frames, flacgpu_decode_stats *st,
                               bool offsets_on_device, uint64_t first_number, uint64_t *d_offsets_out,
                               DecDetail *detail, uint64_t status_capacity, const FgDecRange *h_ranges,
                               uint32_t nranges, bool *again)
{
    *again = false;
    memset(st, 0, sizeof *st);
    if (!HIPOK(hipSetDevice(c->device))) { fg_set_error("hipSetDevice failed"); return false; }
    const bool index_here = h_offsets == nullptr;
    if (index_here && len != 0) {
        if (!c->ensure_pinned_res(64)) return false;
        unsigned long long *hinfo = (unsigned long long *)c->h_res;
        if (!c->dec_info.ensure(64)) return false;
        unsigned long long *d_info = (unsigned long long *)c->dec_info.p;
        if (nframes == 0) {
            c->idx_clean_n = 0;
            // count the candidates (an upper bound of the frames; false candidates are a handful), then see which slots fill
            if (!HIPOK(hipMemsetAsync(d_info, 0, 64, c->stream)) ||
                fg_launch_dec_index((const uint8_t *)d_stream, len, channels_hint, bps_hint, first_number, 0, nullptr, d_info, nullptr, nullptr, 0, c->stream, nullptr, nullptr) != 0 ||
                !HIPOK(hipMemcpyAsync(hinfo, d_info, 64, hipMemcpyDeviceToHost, c->stream)) || !HIPOK(hipStreamSynchronize(c->stream))) {
                fg_set_error("frame index kernel failed"); return false;
            }
            if (hinfo[0] == 0 && hinfo[2]) { fg_set_error("variable block size stream: use flacgpu_index_frames"); return false; }
            // the table holds every frame number seen: a stream with a damaged header has one candidate less than frames, and its
            // last frame must not fall off the table (hinfo[4] = the largest plausible number + 1)
            const unsigned long long want = hinfo[0] > hinfo[4] ? hinfo[0] : hinfo[4];
            if (want > 0x7FFFFFFFull) { fg_set_error("too many frames"); return false; }
            const uint32_t bound = (uint32_t)want;
            if (bound == 0) { st->nframes = 0; return true; }
            c->idx_clean_n = 0;
            if (!c->dec_off.ensure(((size_t)bound + 4) * 8)) return false;
            if (!c->dec_info.ensure(64 + (size_t)bound * 12 + 16)) return false;       // counters, second claims, claim counts
            d_info = (unsigned long long *)c->dec_info.p;
            if (fg_launch_dec_index_init((unsigned long long *)c->dec_off.p, d_info + 8, d_info, bound, nullptr, c->stream) != 0 ||
                fg_launch_dec_index((const uint8_t *)d_stream, len, channels_hint, bps_hint, first_number, bound, (unsigned long long *)c->dec_off.p, d_info, d_info + 8, nullptr, 0, c->stream, nullptr, nullptr) != 0 ||
                !HIPOK(hipMemcpyAsync(hinfo, d_info, 32, hipMemcpyDeviceToHost, c->stream)) || !HIPOK(hipStreamSynchronize(c->stream))) {
                fg_set_error("frame index kernel failed"); return false;
            }
            nframes = (uint32_t)hinfo[3];
            if (nframes == 0) { st->nframes = 0; return true; }
        }
    }
    st->nframes = nframes;
    if (nframes == 0) return true;
    if (channels_hint == 0) { fg_set_error("channel count required"); return false; }
    const uint32_t npad = (nframes + 63) & ~63u;
    if (!c->dec_frames.ensure((size_t)npad * sizeof(FgDecFrame)) || !c->dec_results.ensure((size_t)npad * sizeof(FgDecResult)) ||
        !c->dec_off.ensure(((size_t)nframes + 1 + fg_dec_scan_words(nframes)) * 8))
        return false;
    unsigned long long *d_off = (unsigned long long *)c->dec_off.p;
    if (!index_here) c->idx_clean_n = 0;        // (the caller's offsets go into the table)
    unsigned long long *d_tot = d_off + nframes + 1;
    // end of call and timing as in the encoder (fg_ctx.cpp): level 0 = stamp kernel in front, export kernel at the end (status
    // words, totals and stamps into pinned memory, the host polls a sequence number), no events; levels 1, 2 = HIP events
    // (level 3: as level 0, plus one event in front of the first kernel and one behind the last -- fg_ctx.cpp)
    const bool lean = (c->stage_timing == 0 || c->stage_timing == 3) && !h_frames && !detail;
    const bool ev2 = lean && c->stage_timing == 3;
    if (ev2 && !HIPOK(hipEventRecord(c->ev[0], c->stream))) return false;
    if (lean) {     // (with the index made here, its first kernel takes the stamp)
        if (!index_here && fg_launch_stamp((unsigned long long *)c->stamp.p, c->stream) != 0) { fg_set_error("stamp kernel launch failed"); return false; }
    }
    else if (!HIPOK(hipEventRecord(c->ev[0], c->stream))) return false;
    // The wave parser on its own (FgDecSelf): the index pass leaves a header record per frame, the parser starts as soon as the
    // offsets are settled, header pass and scan of the block sizes move to the side stream in front of the CRC pass, and the
    // restore kernel waits for the frame table.  For one stream whose index is made here, in a call without events, with a
    // residual scratch sized by the caller's capacity; not when the planes are handed out (subframe detail).
    const uint32_t C = channels_hint ? channels_hint : 2;
    // (+ one block of the largest size: the parts of the plane are placed at frame number x stride, FgDecSelf)
    // (... and 16 bytes a frame: a frame's part is rounded up to 16 bytes, which adds up when block size x channels is no multiple
    // of four -- without them the frames behind the point where the sum passes one block's slack all went to the generic decoder)
    const uint64_t cap_bytes = cap_samples * C * 4 + 256 + 65536ull * C * 4 + (uint64_t)nframes * 16;
    const bool queued = cap_bytes <= c->dec_scratch.cap || cap_bytes <= 32 * len + (1u << 20) + 65536ull * C * 4 + (uint64_t)nframes * 16;
    static const bool self_off = fg_sel("FLACGPU_DEC_SELF") && atoi(fg_sel("FLACGPU_DEC_SELF")) == 0;
    bool selfstart = lean && queued && index_here && nranges == 0 && !detail && !self_off && c->stream2 != nullptr;
    uint32_t *d_hrec = nullptr;
    unsigned long long *d_poff = nullptr;
    // (round 5: no event on the main stream between the resolve kernel and the parser, none in front of the restore kernel.  Fork: the
    // resolve kernel raises a word in pinned memory, and the HOST -- it has the parser queued by then and would only wait for the end
    // of the call -- queues the side streams' kernels when it sees it: they start some 8 us behind the parser, whose workgroups are
    // placed by then (let go earlier, the CRC pass takes the wave slots the parser needs to hold all its frames at once; a kernel that
    // waits for the word on the side streams slows the index pass beside it by half).  Join: the side streams' last kernel raises a
    // word in device memory that the restore kernel, queued behind all of it, looks at before it reads what they left.
    // FLACGPU_DEC_GATE=0 in a test-hooks build keeps the events; =2 mutes the join word: the restore kernel's bounded wait times out,
    // the call is repeated with events and the context keeps them; FLACGPU_DEC_DELAY_US=n (test-hooks build) queues a wave that idles
    // for n microseconds in front of the side streams' kernels, so that the restore kernel's wait for the join word does turn --
    // flacgpu_decode_stats.join_late_workgroups counts the workgroups that waited)
    static const int gate_sel = fg_sel("FLACGPU_DEC_GATE") ? atoi(fg_sel("FLACGPU_DEC_GATE")) : 1;
    static const long side_delay_us = fg_sel("FLACGPU_DEC_DELAY_US") ? atol(fg_sel("FLACGPU_DEC_DELAY_US")) : 0;
    bool use_gate = selfstart && gate_sel != 0 && !c->gate_off && len != 0;      // (no bytes: no resolve kernel to raise the word)
    if (selfstart) {
        if (!c->dec_poff.ensure((size_t)npad * 8) || !c->dec_hrec.ensure((size_t)npad * 4)) return false;
        d_poff = (unsigned long long *)c->dec_poff.p; d_hrec = (uint32_t *)c->dec_hrec.p;
    }
    const unsigned long long gate_epoch = use_gate ? ++c->gate_epoch : 0;
    if (index_here) {
        // one pass over the bytes: every frame header found puts its position into the slot of its frame number
        if (!c->dec_info.ensure(64 + (size_t)nframes * 12 + 16)) return false;      // counters, second claims, claim counts
        unsigned long long *d_info = (unsigned long long *)c->dec_info.p;
        // (a buffer this has not been done for -- a new one, or the 4 KB the count pass's ensure(64) above got on a context's first
        // call, which a small table then fits into without a new allocation --: the join word behind the counters must not hold
        // anything that looks like an epoch.  A new buffer never has the address of the one it replaces: DevBuf::ensure allocates first)
        if (c->dec_info.p != c->gate_words_of) {
            if (!HIPOK(hipMemsetAsync(d_info, 0, 64, c->stream)) || !HIPOK(hipStreamSynchronize(c->stream))) return false;
            c->gate_words_of = c->dec_info.p;
        }
        const FgDecRange *d_ranges = nullptr;
        if (nranges) {
            // several streams in one buffer: every stream files its frames from its own slot on
            if (!c->dec_ranges.ensure((size_t)nranges * sizeof(FgDecRange)) ||
                !HIPOK(hipMemcpyAsync(c->dec_ranges.p, h_ranges, (size_t)nranges * sizeof(FgDecRange), hipMemcpyHostToDevice, c->stream))) return false;
            d_ranges = (const FgDecRange *)c->dec_ranges.p;
        }
        // The tables the index pass files its claims in are emptied by the call that used them last, BEHIND its end-of-call signal
        // (below): a call that finds them as that call left them starts with the index pass itself (its first workgroup takes the
        // start-of-call stamp) -- the init kernel and the turn-around behind it were 7 us in front of every decode launch.
        const bool tables_clean = c->idx_clean_n == nframes && c->idx_clean_off == (void *)d_off && c->idx_clean_info == (void *)d_info;
        c->idx_clean_n = 0;
        unsigned long long *const d_stamp = lean ? (unsigned long long *)c->stamp.p : nullptr;
        if ((!tables_clean && fg_launch_dec_index_init(d_off, d_info + 8, d_info, nframes, d_stamp, c->stream) != 0) ||
            fg_launch_dec_index((const uint8_t *)d_stream, len, channels_hint, bps_hint, first_number, nframes, d_off, d_info, d_info + 8, d_ranges, nranges, c->stream, d_hrec,
                                tables_clean ? d_stamp : nullptr, use_gate ? c->h_sig + 1 : nullptr, gate_epoch) != 0) {
            fg_set_error("frame index kernel launch failed"); return false;
        }
        // (the end of the last frame, offsets[nframes] = len, is set by the index kernel)
        if (d_offsets_out && !HIPOK(hipMemcpyAsync(d_offsets_out, d_off, ((size_t)nframes + 1) * 8, hipMemcpyDeviceToDevice, c->stream))) return false;
        if (!lean) (void)hipEventRecord(c->ev[3], c->stream);
    }
    else if (!HIPOK(hipMemcpyAsync(d_off, h_offsets, ((size_t)nframes + 1) * 8, offsets_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream))) { fg_set_error("copy of the frame offsets failed"); return false; }
    // (a stream of its own for header pass and scan: the CRC pass, on the second stream, starts from the offsets beside them --
    // queued behind them it found the chip full of parser waves and ended after the parser)
    // The parser is queued FIRST, header pass, scan and CRC pass behind it in host order (below): queued in front of it the CRC
    // pass took the wave slots the parser needs to have all its frames resident at once, and the host's calls for the side
    // streams sat between the index pass and the parser.
    hipStream_t hstream = c->stream;
    if (selfstart) {
        hstream = c->gstream[0] ? c->gstream[0] : c->stream2;
        if (!use_gate) selfstart = HIPOK(hipEventRecord(c->evx[0], c->stream));
        if (!selfstart) { hstream = c->stream; use_gate = false; }
    }
    if (!selfstart && fg_launch_dec_headers((const uint8_t *)d_stream, len, d_off, nframes, channels_hint, bps_hint, (FgDecFrame *)c->dec_frames.p,
                                            (FgDecResult *)c->dec_results.p, d_tot, cap_samples, c->stream, 1) != 0) {
        fg_set_error("header kernel launch failed"); return false;
    }
    if (!c->ensure_pinned_res(64 + ((size_t)nframes + 2) * sizeof(FgDecResult))) return false;
    unsigned long long *tot = (unsigned long long *)c->h_res;
    tot[0] = tot[1] = 0;
    // The residual scratch is sized by the caller's capacity when that is a sane bound (the scan kernel rejects frames past
    // it), so the decode kernels are queued without waiting for the header pass; a wildly generous capacity (no STREAMINFO:
    // 65535 samples per frame) waits for the real total instead of allocating for it.  (cap_bytes, queued: above)
    if (!queued) {
        if (!HIPOK(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, c->stream)) || !HIPOK(hipStreamSynchronize(c->stream))) {
            fg_set_error("header pass failed"); return false;
        }
        if (tot[0] > cap_samples) { st->total_samples = tot[0]; fg_set_error("PCM output buffer too small"); return false; }
    }
    if (!c->dec_scratch.ensure(queued ? (size_t)cap_bytes : (size_t)std::max<uint64_t>(tot[0], 1) * C * 4 + 256) ||
        !c->dec_subs.ensure((size_t)npad * C * sizeof(FgDecSub)))
        return false;
    if (!lean && !HIPOK(hipEventRecord(c->ev[1], c->stream))) return false;
    const int wide = (bps_hint == 0 || bps_hint > 16) ? 1 : 0;
    // The kernels of a decode launch (the lane-serial decoders of rounds 1 and 2 and the fused kernel left the tree in round 5, the
    // orders of CRC pass and restore kernel tried in round 4 -- FLACGPU_DEC_CRC_LATE -- in round 6): the wave parser
    // (flac_dec_wave.hip) on the main stream; the CRC-16 pass, which only needs the stream and the frame positions, beside it on a
    // second stream; the restore kernel behind both, merging the CRC verdict into the frame status and, in a call without events,
    // sending the status words to the host's pinned copy on the way.  FLACGPU_DEC_WAVE=2 in a test-hooks build counts the parser's
    // batches and sync rounds.
    static const bool count_rounds = fg_sel("FLACGPU_DEC_WAVE") && atoi(fg_sel("FLACGPU_DEC_WAVE")) >= 2;
    // 16-bit residual plane between the wave parser and its restore kernel (streams of up to 16 bits; flac_dec_wave.hip P16).  Not
    // when the planes themselves are handed out (subframe detail: FLAC__Frame.subframes[].residual) or read by the warm-up kernel.
    static const bool p16_off = fg_sel("FLACGPU_DEC_P16") && atoi(fg_sel("FLACGPU_DEC_P16")) == 0;
    // A frame with a value beyond 16 bits ends the parse with status 6: the call is repeated with 32-bit planes, and so are the
    // next calls of this context (a stream that does it once does it again: full-scale noise, a side channel at full scale).
    const int plane16 = (!wide && !detail && !p16_off && c->dec_p16_hold == 0) ? 1 : 0;
    if (c->dec_p16_hold) c->dec_p16_hold--;
    bool forked = false;
    FgDecResult *const h_rows_pinned = (FgDecResult *)((char *)c->h_res + 64);
    if (!selfstart) {
        // (the frame table is there: the CRC pass starts beside the parser from an event behind the header pass)
        forked = HIPOK(hipEventRecord(c->evx[0], c->stream)) && HIPOK(hipStreamWaitEvent(c->stream2, c->evx[0], 0));
        if (fg_launch_decode_crc((const uint8_t *)d_stream, (const FgDecFrame *)c->dec_frames.p, nframes, (FgDecResult *)c->dec_results.p,
                                 (const uint16_t *)c->crctab.p, forked ? c->stream2 : c->stream, nullptr, len) != 0) {
            fg_set_error("decode kernel launch failed"); return false;
        }
        if (forked && !HIPOK(hipEventRecord(c->evx[1], c->stream2))) return false;
    }
    uint16_t *d_rparams = nullptr;
    if (detail && detail->level >= 1) {
        if (!c->dec_rparams.ensure((size_t)npad * C * FG_DEC_RPARAMS * 2) || !c->dec_warm.ensure((size_t)npad * C * 32 * 4)) return false;
        d_rparams = (uint16_t *)c->dec_rparams.p;
        // (the subframe records of frames the fast decoder does not take stay zero: flags bit 12 = valid)
        if (!HIPOK(hipMemsetAsync(c->dec_subs.p, 0, (size_t)npad * C * sizeof(FgDecSub), c->stream))) return false;
    }
    {
        // wave-parallel parse (flac_dec_wave.hip): one frame per wavefront
        unsigned long long *d_cnt = nullptr;
        FgDecSelf self;
        self.offsets = d_off; self.hdrrec = d_hrec; self.planeoff = d_poff; self.plane_cap_bytes = c->dec_scratch.cap;
        self.si_bps = bps_hint; self.reserved = 0;
        // ([5]: lower half a wait on the join word that ran out, upper half the workgroups that had to wait; [7]: the join word)
        unsigned long long *const d_gw = (unsigned long long *)c->dec_info.p;
        if (count_rounds && c->dec_prof.ensure(64)) { d_cnt = (unsigned long long *)c->dec_prof.p; (void)hipMemsetAsync(d_cnt, 0, 64, c->stream); }
        if (fg_launch_decode_wparse((const uint8_t *)d_stream, len, (const FgDecFrame *)c->dec_frames.p, nframes, (int32_t *)c->dec_scratch.p,
                                    (FgDecSub *)c->dec_subs.p, (FgDecResult *)c->dec_results.p, wide, d_rparams, d_cnt, c->stream, plane16,
                                    selfstart ? &self : nullptr) != 0) {
            fg_set_error("decode kernel launch failed"); return false;
        }
        if (selfstart) {
            // beside the parser, from the index pass's end: header pass + scan on one stream, the CRC pass (from the offsets) on
            // another; the restore kernel reads the frame table and the verdicts and waits for both
            // (one wait in front of the restore kernel, not two: the header stream waits for the CRC stream's event before it
            // records its own -- every wait on the main stream is some 5 us of idle GPU)
            if (use_gate) {
                // The fork: wait here, not on the GPU, for the word the resolve kernel's last workgroup raises.  The index pass of
                // one stream takes some 50 us: the first 250 us are spent looking at the word; behind them (a batch of streams,
                // a GPU shared with somebody else) the thread sleeps between looks -- a decoder's thread must not keep a core busy
                // for as long as the GPU takes --, and behind 3 ms the main stream is waited for, which is slower and as good.
                volatile unsigned long long *const fw = c->h_sig + 1;
                const auto t0 = std::chrono::steady_clock::now();
                bool seen = false, sleeping = false;
                for (unsigned it = 0;; it++) {
                    if (__atomic_load_n(fw, __ATOMIC_ACQUIRE) == gate_epoch) { seen = true; break; }
                    if (!sleeping && (it & 63) != 63) continue;
                    const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                    if (us > 3000) break;
                    if (us > 250) { sleeping = true; struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
                }
                if (!seen && !HIPOK(hipStreamSynchronize(c->stream))) { fg_set_error("decode kernel failed"); return false; }
            }
            if (!(use_gate || HIPOK(hipStreamWaitEvent(c->stream2, c->evx[0], 0))) ||
                (side_delay_us > 0 && fg_launch_dec_spin((unsigned long long)side_delay_us * 100ull, c->stream2) != 0) ||
                fg_launch_decode_crc((const uint8_t *)d_stream, (const FgDecFrame *)c->dec_frames.p, nframes, (FgDecResult *)c->dec_results.p,
                                     (const uint16_t *)c->crctab.p, c->stream2, d_off, len) != 0 ||
                !HIPOK(hipEventRecord(c->evx[1], c->stream2))) { fg_set_error("decode kernel launch failed"); return false; }
            if (!(use_gate || HIPOK(hipStreamWaitEvent(hstream, c->evx[0], 0))) ||
                (side_delay_us > 0 && hstream != c->stream2 && fg_launch_dec_spin((unsigned long long)side_delay_us * 100ull, hstream) != 0) ||
                fg_launch_dec_headers((const uint8_t *)d_stream, len, d_off, nframes, channels_hint, bps_hint, (FgDecFrame *)c->dec_frames.p,
                                      (FgDecResult *)c->dec_results.p, d_tot, cap_samples, hstream, 0) != 0 ||
                (hstream != c->stream2 && !HIPOK(hipStreamWaitEvent(hstream, c->evx[1], 0))) ||
                !(use_gate ? (gate_sel == 2 || fg_launch_dec_raise(d_gw + 7, gate_epoch, hstream) == 0) : HIPOK(hipEventRecord(c->evx[2], hstream)))) { fg_set_error("header kernel launch failed"); return false; }
            // (evx[2] stands for both side streams.  With the gate the restore kernel looks at the join word itself: no wait on the main stream)
            if (!use_gate && !HIPOK(hipStreamWaitEvent(c->stream, c->evx[2], 0))) return false;
        }
        if (d_cnt) {
            unsigned long long hc[3] = {0, 0, 0};
            if (HIPOK(hipMemcpyAsync(hc, d_cnt, 24, hipMemcpyDeviceToHost, c->stream)) && HIPOK(hipStreamSynchronize(c->stream)))
                fprintf(stderr, "[flacgpu dec wave] %u frames: %llu batches, %llu sync rounds, %llu long codes\n", nframes, hc[0], hc[1], hc[2]);
        }
    }
    if (forked && !HIPOK(hipStreamWaitEvent(c->stream, c->evx[1], 0))) return false;
    // (the status words straight to the host's pinned copy in a call without events: this kernel has the last word on them)
    const bool rows_sent = lean;
    if (fg_launch_decode_wrestore((const FgDecFrame *)c->dec_frames.p, nframes, C, (const int32_t *)c->dec_scratch.p, (const FgDecSub *)c->dec_subs.p,
                                  (int32_t *)d_pcm, (FgDecResult *)c->dec_results.p, interleave ? 1u : 0u, wide, c->stream, plane16,
                                  rows_sent ? h_rows_pinned : nullptr, selfstart ? d_poff : nullptr,
                                  use_gate ? (unsigned long long *)c->dec_info.p + 7 : nullptr, gate_epoch) != 0) {
        fg_set_error("decode kernel launch failed"); return false;
    }
    if (!lean && !HIPOK(hipEventRecord(c->ev[2], c->stream))) return false;
    if (detail && detail->level >= 1) {
        if (fg_launch_decode_warmup((const FgDecFrame *)c->dec_frames.p, nframes, C, (const FgDecSub *)c->dec_subs.p, (const int32_t *)c->dec_scratch.p,
                                    (int32_t *)c->dec_warm.p, c->stream) != 0) { fg_set_error("decode kernel launch failed"); return false; }
        detail->subs.resize((size_t)nframes * C); detail->rparams.resize((size_t)nframes * C * FG_DEC_RPARAMS); detail->warm.resize((size_t)nframes * C * 32);
        if (!HIPOK(hipMemcpyAsync(detail->subs.data(), c->dec_subs.p, detail->subs.size() * sizeof(FgDecSub), hipMemcpyDeviceToHost, c->stream)) ||
            !HIPOK(hipMemcpyAsync(detail->rparams.data(), c->dec_rparams.p, detail->rparams.size() * 2, hipMemcpyDeviceToHost, c->stream)) ||
            !HIPOK(hipMemcpyAsync(detail->warm.data(), c->dec_warm.p, detail->warm.size() * 4, hipMemcpyDeviceToHost, c->stream))) return false;
    }
    FgDecResult *res = (FgDecResult *)((char *)c->h_res + 64);
    unsigned long long *hinfo2 = (unsigned long long *)((char *)c->h_res + 32);
    if (lean) {
        const unsigned long long seq = ++c->sig_seq;
        if (ev2 && !HIPOK(hipEventRecord(c->ev[2], c->stream))) return false;
        if (fg_launch_signal(d_tot, 2, index_here ? (const unsigned long long *)c->dec_info.p : nullptr, index_here ? 6 : 0,
                             (const unsigned long long *)c->stamp.p, c->h_sig, seq, c->stream) != 0 || !c->wait_signal(seq)) {
            fg_set_error("decode kernel failed"); return false;
        }
        tot[0] = c->h_sig[2]; tot[1] = c->h_sig[3];
        for (int k = 0; k < 4; k++) hinfo2[k] = index_here ? c->h_sig[4 + k] : 0;
        if (use_gate) {
            st->join_late_workgroups = (uint32_t)(c->h_sig[9] >> 32);
            if ((c->h_sig[9] & 0xFFFFFFFFull) != 0) {
                // a wait on the join word ran out (never seen outside tools that serialise kernels across streams): the restore
                // kernel's workgroups that gave up wrote nothing, the side streams' kernels may still be running -- wait for them
                // (they read the index tables, which must not be emptied under them), then once more, and from now on, with events
                c->gate_off = true;
                if (!HIPOK(hipStreamSynchronize(c->stream2)) || !HIPOK(hipStreamSynchronize(hstream)) || !HIPOK(hipStreamSynchronize(c->stream))) {
                    fg_set_error("decode kernel failed"); return false;
                }
                *again = true;
                return false;
            }
        }
        // (the index tables for the next call of this shape: emptied behind the signal, while the host is on its way back -- and
        // behind the look at the timeout word above: the side streams' kernels have ended when the join word was seen)
        if (index_here && !FG_NO_DEFER_INIT && fg_launch_dec_index_init(d_off, (unsigned long long *)c->dec_info.p + 8, (unsigned long long *)c->dec_info.p, nframes, nullptr, c->stream) == 0) {
            c->idx_clean_n = nframes; c->idx_clean_off = (void *)d_off; c->idx_clean_info = c->dec_info.p;
        }
    }
    else {
        if (queued && !HIPOK(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, c->stream))) return false;
        if (!HIPOK(hipMemcpyAsync(res, c->dec_results.p, (size_t)nframes * sizeof(FgDecResult), hipMemcpyDeviceToHost, c->stream))) return false;
        if (h_frames) {
            h_frames->resize(nframes);
            if (!HIPOK(hipMemcpyAsync(h_frames->data(), c->dec_frames.p, (size_t)nframes * sizeof(FgDecFrame), hipMemcpyDeviceToHost, c->stream))) return false;
        }
        if (index_here && !HIPOK(hipMemcpyAsync(hinfo2, c->dec_info.p, 32, hipMemcpyDeviceToHost, c->stream))) return false;
        if (!HIPOK(fg_stream_wait(c->stream))) { fg_set_error("decode kernel failed"); return false; }
    }
    if (index_here) {
        if (!lean) (void)hipEventElapsedTime(&st->index_ms, c->ev[0], c->ev[3]);
        if (hinfo2[1]) { fg_set_error("ambiguous frame sync (several headers claim one frame number and their order does not decide): use flacgpu_index_frames"); return false; }
    }
    st->total_samples = tot[0];
    st->max_blocksize = (uint32_t)tot[1];
    if (tot[0] > cap_samples) { fg_set_error("PCM output buffer too small"); return false; }
    if (detail && detail->level >= 2 && tot[0]) {
        detail->planes.resize((size_t)tot[0] * C);
        if (!HIPOK(hipMemcpy(detail->planes.data(), c->dec_scratch.p, detail->planes.size() * 4, hipMemcpyDeviceToHost))) return false;
    }
    if (plane16) {
        bool wide_values = false;
        for (uint32_t i = 0; i < nframes; i++) if (res[i].err == 6) { wide_values = true; break; }
        if (wide_values) {
            c->dec_p16_hold = 256;
            *again = true;
            return false;
        }
    }
    {
        // frames outside the register-resident decoder's envelope (predictor order > 12, ...) go through the generic kernel
        std::vector<uint32_t> redo;
        for (uint32_t i = 0; i < nframes; i++) if (res[i].err == 3) redo.push_back(i);
        st->generic_frames = (uint32_t)redo.size();
        if (!redo.empty()) {
            if (!c->dec_redo.ensure(redo.size() * 4)) return false;   // own buffer: `descs` caches the encoder's block list
            if (!HIPOK(hipMemcpyAsync(c->dec_redo.p, redo.data(), redo.size() * 4, hipMemcpyHostToDevice, c->stream))) return false;
            if (fg_launch_decode_slow((const uint8_t *)d_stream, (const FgDecFrame *)c->dec_frames.p, (const uint32_t *)c->dec_redo.p, (uint32_t)redo.size(),
                                      (int32_t *)d_pcm, (FgDecResult *)c->dec_results.p, (const uint16_t *)c->crctab.p, (int32_t *)c->dec_scratch.p,
                                      interleave ? 1u : 0u, c->stream) != 0) { fg_set_error("decode kernel launch failed"); return false; }
            if (!HIPOK(hipMemcpyAsync(res, c->dec_results.p, (size_t)nframes * sizeof(FgDecResult), hipMemcpyDeviceToHost, c->stream)) ||
                !HIPOK(hipStreamSynchronize(c->stream))) { fg_set_error("decode kernel failed"); return false; }
        }
    }
    if (ev2) { (void)hipEventSynchronize(c->ev[2]); (void)hipEventElapsedTime(&st->total_gpu_ms, c->ev[0], c->ev[2]); }      // (see fg_ctx.cpp)
    else if (lean) st->total_gpu_ms = (float)((double)(c->h_sig[11] - c->h_sig[10]) / c->wall_khz);     // (decode_kernel_ms, index_ms: levels 1, 2)
    else {
        (void)hipEventElapsedTime(&st->decode_kernel_ms, c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&st->total_gpu_ms, c->ev[0], c->ev[2]);
    }
    uint32_t bad = 0;
    for (uint32_t i = 0; i < nframes; i++) if (res[i].err) bad++;
    st->error_frames = bad;
    st->plane_bits = plane16 ? 16u : 32u;
    if (index_here && bad == nframes && ((unsigned long long *)((char *)c->h_res + 32))[2]) {
        // nothing was found under the fixed-block-size sync code, but headers with the variable-block-size one were
        fg_set_error("variable block size stream: use flacgpu_index_frames"); return false;
    }
    st->channels = C; st->bits_per_sample = bps_hint;
    if (h_status) memcpy(h_status, res, (size_t)std::min<uint64_t>(nframes, status_capacity) * sizeof(FgDecResult));
    return true;
}
static bool decode_frames_impl(flacgpu_ctx *c, const void *d_stream, uint64_t len, const uint64_t *h_offsets, uint32_t nframes,
                               uint32_t channels_hint, uint32_t bps_hint, void *d_pcm, uint64_t cap_samples, int interleave,
                               FgDecResult *h_status, std::vector<FgDecFrame> *h_frames, flacgpu_decode_stats *st,
                               bool offsets_on_device = false, uint64_t first_number = 0, uint64_t *d_offsets_out = nullptr,
                               DecDetail *detail = nullptr, uint64_t status_capacity = ~0ull, const FgDecRange *h_ranges = nullptr,
                               uint32_t nranges = 0)
{
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    // (at most three attempts: the first, one with events in place of the words in memory, one with 32-bit planes)
    for (int attempt = 0; attempt < 3; attempt++) {
        bool again = false;
        if (decode_frames_once(c, d_stream, len, h_offsets, nframes, channels_hint, bps_hint, d_pcm, cap_samples, interleave, h_status, h_frames, st,
                               offsets_on_device, first_number, d_offsets_out, detail, status_capacity, h_ranges, nranges, &again)) return true;
        if (!again) return false;
    }
    fg_set_error("decode call did not settle after three attempts");
    return false;
}


extern "C" int flacgpu_decode_frames_dev(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, const uint64_t *d_frame_offsets,
                                         uint32_t nframes, uint32_t channels_hint, uint32_t bps_hint, void *d_pcm,
                                         uint64_t pcm_capacity_samples, void *h_frame_status, flacgpu_decode_stats *stats)
{
    flacgpu_decode_stats local;
    if (!stats) stats = &local;
    if (!ctx) { fg_set_error("null context"); return -1; }
    return decode_frames_impl(ctx, d_stream, len, d_frame_offsets, nframes, channels_hint, bps_hint, d_pcm, pcm_capacity_samples, 1,
                              (FgDecResult *)h_frame_status, nullptr, stats, true) ? 0 : -1;
}

extern "C" int flacgpu_decode_stream_dev(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, uint32_t nframes_hint, uint64_t first_frame_number,
                                         uint32_t channels, uint32_t bps, void *d_pcm, uint64_t pcm_capacity_samples, void *h_frame_status,
                                         uint64_t status_capacity, void *d_frame_offsets_out, flacgpu_decode_stats *stats)
{
    flacgpu_decode_stats local;
    if (!stats) stats = &local;
    if (!ctx) { fg_set_error("null context"); return -1; }
    if (channels == 0) { fg_set_error("channel count required"); return -1; }
    return decode_frames_impl(ctx, d_stream, len, nullptr, nframes_hint, channels, bps, d_pcm, pcm_capacity_samples, 1,
                              (FgDecResult *)h_frame_status, nullptr, stats, true, first_frame_number, (uint64_t *)d_frame_offsets_out,
                              nullptr, status_capacity) ? 0 : -1;
}

// Several fixed-block-size streams laid back to back in one device buffer, decoded from their bytes alone in one launch (the
// decode side of BASELINE config 5: the frame index of every stream is made on the GPU).
extern "C" int flacgpu_decode_streams_dev(flacgpu_ctx *ctx, const void *d_bytes, uint64_t len, const flacgpu_stream_range *ranges, uint32_t nranges,
                                          uint32_t channels, uint32_t bps, void *d_pcm, uint64_t pcm_capacity_samples, void *h_frame_status,
                                          uint64_t status_capacity, void *d_frame_offsets_out, flacgpu_decode_stats *stats)
{
    flacgpu_decode_stats local;
    if (!stats) stats = &local;
    if (!ctx) { fg_set_error("null context"); return -1; }
    if (channels == 0) { fg_set_error("channel count required"); return -1; }
    if (!ranges || nranges == 0) { fg_set_error("no streams"); return -1; }
    std::vector<FgDecRange> rg(nranges);
    uint64_t at = 0, frames = 0;
    for (uint32_t i = 0; i < nranges; i++) {
        if (ranges[i].byte_offset != at) { fg_set_error("the streams must lie back to back from the start of the buffer"); return -1; }
        if (ranges[i].nframes == 0) { fg_set_error("frame count of every stream required (STREAMINFO: total samples / block size)"); return -1; }
        rg[i].byte_start = at; rg[i].first_number = ranges[i].first_frame_number; rg[i].slot_base = (uint32_t)frames; rg[i].nframes = ranges[i].nframes;
        at += ranges[i].byte_length; frames += ranges[i].nframes;
        if (frames > 0x7FFFFFFFull) { fg_set_error("too many frames"); return -1; }
    }
    if (at != len) { fg_set_error("the streams must cover the whole buffer"); return -1; }
    return decode_frames_impl(ctx, d_bytes, len, nullptr, (uint32_t)frames, channels, bps, d_pcm, pcm_capacity_samples, 1,
                              (FgDecResult *)h_frame_status, nullptr, stats, true, 0, (uint64_t *)d_frame_offsets_out, nullptr,
                              status_capacity, rg.data(), nranges) ? 0 : -1;
}

extern "C" int flacgpu_decode_frames(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, const uint64_t *h_frame_offsets,
                                     uint32_t nframes, uint32_t channels_hint, uint32_t bps_hint, void *d_pcm,
                                     uint64_t pcm_capacity_samples, void *h_frame_status, flacgpu_decode_stats *stats)
{
    flacgpu_decode_stats local;
    if (!stats) stats = &local;
    if (!ctx) { fg_set_error("null context"); return -1; }
    return decode_frames_impl(ctx, d_stream, len, h_frame_offsets, nframes, channels_hint, bps_hint, d_pcm, pcm_capacity_samples, 1,
                              (FgDecResult *)h_frame_status, nullptr, stats) ? 0 : -1;
}

// ------------------------------------------------------------------ libFLAC-style stream decoder
namespace {

// Landing buffer of the decoded PCM on the host: pinned (the copy from the device runs at PCIe speed instead of through the
// runtime's staging buffers), grow-only and never initialised (value-initialising 200 MB -- what a std::vector does on
// resize -- took a quarter of a one-shot decode), and handed from a finished decoder to the next one through a one-slot
// pool, so that a sequence of decoders pins and faults the pages in once.
struct HostPcm {
    int32_t *p = nullptr;
    size_t cap = 0;         // in samples
    bool pinned = false;
    void drop() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; pinned = false; }
    static std::mutex &pool_mu() { static std::mutex m; return m; }
    static HostPcm &pool() { static HostPcm spare; return spare; }
    bool ensure(size_t n)
    {
        if (n <= cap) return true;
        {
            std::lock_guard<std::mutex> lk(pool_mu());
            HostPcm &sp = pool();
            if (sp.cap >= n) { drop(); p = sp.p; cap = sp.cap; pinned = sp.pinned; sp.p = nullptr; sp.cap = 0; sp.pinned = false; return true; }
        }
        drop();
        void *np = nullptr;
        if (hipHostMalloc(&np, n * sizeof(int32_t), hipHostMallocDefault) == hipSuccess) { p = (int32_t *)np; pinned = true; }
        else { (void)hipGetLastError(); p = (int32_t *)malloc(n * sizeof(int32_t)); pinned = false; }
        cap = p ? n : 0;
        return p != nullptr;
    }
    void give_back()
    {
        std::lock_guard<std::mutex> lk(pool_mu());
        HostPcm &sp = pool();
        if (cap > sp.cap) { sp.drop(); sp.p = p; sp.cap = cap; sp.pinned = pinned; p = nullptr; cap = 0; pinned = false; }
        else drop();
    }
    int32_t *data() const { return p; }
};

// The bytes not yet consumed.  What the decoder needs of std::vector<uint8_t>, without its value-initialisation: the read callback
// is handed up to 16 MiB of room at a time, and zeroing that room first cost as much as the whole decode of a 600 s stream.
struct ByteBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf &) = delete;
    ByteBuf &operator=(const ByteBuf &) = delete;
    ~ByteBuf() { free(p); }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint8_t *begin() { return p; }
    uint8_t &operator[](size_t i) { return p[i]; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    // one allocation of a finished decoder is kept for the next one (its pages are already there)
    static std::mutex &pool_mu() { static std::mutex m; return m; }
    static ByteBuf &pool() { static ByteBuf spare; return spare; }
    void give_back()
    {
        std::lock_guard<std::mutex> lk(pool_mu());
        ByteBuf &sp = pool();
        if (cap > sp.cap) { free(sp.p); sp.p = p; sp.cap = cap; p = nullptr; cap = 0; n = 0; }
        else if (cap > (1u << 20)) { free(p); p = nullptr; cap = 0; n = 0; }
        else n = 0;
    }
    bool resize(size_t m)          // (new bytes are NOT initialised; false: out of memory, nothing changed)
    {
        if (m > cap && cap < (1u << 20)) {
            std::lock_guard<std::mutex> lk(pool_mu());
            ByteBuf &sp = pool();
            if (sp.cap >= m && sp.cap > cap) {
                if (n) memcpy(sp.p, p, n);
                free(p); p = sp.p; cap = sp.cap; sp.p = nullptr; sp.cap = 0;
            }
        }
        if (m > cap) {
            size_t nc = cap ? cap : 65536;
            while (nc < m) nc *= 2;
            uint8_t *np = (uint8_t *)realloc(p, nc);
            if (!np) return false;
            p = np; cap = nc;
        }
        n = m;
        return true;
    }
    void erase(uint8_t *a, uint8_t *b) { memmove(a, b, (size_t)(p + n - b)); n -= (size_t)(b - a); }
};

// Device buffers of finished stream decoders, kept for the next one (hipMalloc / hipFree of a few hundred megabytes cost
// milliseconds): a handful of slots, the largest buffers stay.
struct DevPool {
    struct Slot { DevBuf b; int device = -1; };
    static std::mutex &mu() { static std::mutex m; return m; }
    static Slot *slots() { static Slot s[4]; return s; }
    static bool take(DevBuf &into, size_t bytes, int device)
    {
        std::lock_guard<std::mutex> lk(mu());
        Slot *s = slots();
        int best = -1;
        for (int i = 0; i < 4; i++) if (s[i].b.p && s[i].device == device && s[i].b.cap >= bytes && (best < 0 || s[i].b.cap < s[best].b.cap)) best = i;
        if (best < 0) return false;
        into.release();
        into = s[best].b; s[best].b.p = nullptr; s[best].b.cap = 0; s[best].device = -1;
        return true;
    }
    static void give(DevBuf &from, int device)
    {
        if (!from.p) return;
        std::lock_guard<std::mutex> lk(mu());
        Slot *s = slots();
        int at = -1;
        for (int i = 0; i < 4; i++) if (!s[i].b.p) { at = i; break; }
        if (at < 0) { for (int i = 0; i < 4; i++) if (at < 0 || s[i].b.cap < s[at].b.cap) at = i; if (s[at].b.cap >= from.cap) { from.release(); return; } s[at].b.release(); }
        s[at].b = from; s[at].device = device;
        from.p = nullptr; from.cap = 0;
    }
};
inline bool dev_ensure(DevBuf &b, size_t bytes, int device)
{
    if (b.cap >= bytes) return true;
    if (DevPool::take(b, bytes, device)) return true;
    return b.ensure(bytes);
}

struct DecImpl;
void fill_subframes(DecImpl *d, FLAC__Frame &f, const FgDecFrame &fr, uint32_t fi);

struct DecImpl {
    FLAC__StreamDecoder pub;
    FLAC__StreamDecoderState state;
    FLAC__bool md5_checking;
    FLAC__StreamDecoderReadCallback read_cb;
    FLAC__StreamDecoderWriteCallback write_cb;
    FLAC__StreamDecoderMetadataCallback meta_cb;
    FLAC__StreamDecoderErrorCallback error_cb;
    void *client;
    FILE *file;
    bool own_file;
    bool respond[128];                          // metadata filter by block type (stream_decoder.h:849-973)
    std::vector<uint32_t> app_ids;              // APPLICATION ids whose filter is the opposite of respond[APPLICATION]
    flacgpu_ctx *ctx;
    // stream state
    ByteBuf buf;                    // bytes not yet consumed (from `base` on)
    uint64_t consumed_total;        // stream offset of buf[0]
    bool eof;
    bool have_meta, have_si;
    FLAC__StreamMetadata_StreamInfo si;
    Indexer ix;
    uint64_t frames_delivered_bound;  // index into ix.bounds of the next frame to decode
    uint64_t samples_decoded;
    // decoded frames waiting for delivery
    HostPcm pcm;                      // frame-planar
    std::vector<FgDecFrame> frames;
    std::vector<FgDecResult> status;
    size_t next_frame;
    uint32_t last_blocksize, last_ca;
    uint64_t first_pos;               // offset in buf of the first queued frame
    uint32_t fixed_blocksize;         // block size of a fixed-blocksize stream without a usable STREAMINFO
    FLAC__FrameHeader last_hdr;       // header of the last delivered frame (gap filling)
    bool do_md5;                      // MD5 checking still meaningful for this stream (libFLAC do_md5_checking)
    FgMd5 md5;
    std::vector<int32_t> md5_tmp;
    bool last_set;
    std::vector<int32_t> silence;
    DevBuf d_stream, d_pcm;
    // FLAC__Frame.subframes[] of the frame being delivered
    int subframe_detail = 1;      // flacgpu_stream_decoder_set_subframe_detail
    DecDetail detail;
    // Damaged data (fg_refwalk.h): where the index or the GPU pass finds anything irregular, the frames before that place are
    // delivered and libFLAC's serial reader is replayed on the host from there -- its error statuses, in order, and the frame it
    // decodes next, where the index takes over again.
    fgref::RefWindows win;        // the answers of the read callback, as the refills of libFLAC's 8 KiB reader would have seen them
    fgref::Walker walker;         // (the reader's buffer as the last walk left it)
    // seeking (FLAC__stream_decoder_seek_absolute): possible on a FILE of the library's own and with the client's four callbacks
    FLAC__StreamDecoderSeekCallback seek_cb = nullptr;
    FLAC__StreamDecoderTellCallback tell_cb = nullptr;
    FLAC__StreamDecoderLengthCallback length_cb = nullptr;
    FLAC__StreamDecoderEofCallback eof_cb = nullptr;
    uint64_t audio_start = 0;         // stream offset of the first frame (behind the metadata)
    bool seeking = false;             // frames in front of seek_target are dropped, the frame that holds it is delivered from there
    uint64_t seek_target = 0;
    bool seek_done = false;
    uint8_t pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // the last bytes dropped from the front of buf (libFLAC's buffer front may lie a few bytes before buf[0])
    uint32_t npre = 0;
    bool walk_pending = false;
    uint64_t walk_from = 0;       // offset in buf
    uint64_t ix_origin = 0;       // offset in buf where the index last started outside a frame
    size_t pull_want = 1 << 16;
    double prof_ms[6] = {0, 0, 0, 0, 0, 0};     // FLACGPU_API_PROF: pull, index, decode_available, delivery, erase, metadata
    // block delivery (flacgpu_stream_decoder_set_block_callback)
    flacgpu_block_callback block_cb = nullptr;
    bool round_blocks = false;    // the queued round was decoded for block delivery: pcm holds it interleaved
    uint32_t round_bytes = 4;     // ... as int16 (2) or int32 (4)
    std::vector<flacgpu_block> blocks;
    DevBuf d_pcm16;
    uint64_t walk_stuck_at = UINT64_MAX;
    FLAC__EntropyCodingMethod_PartitionedRiceContents rice_contents[8];
    std::vector<uint32_t> rice_prm[8], rice_raw[8];
};

inline DecImpl *impl(FLAC__StreamDecoder *d) { return reinterpret_cast<DecImpl *>(d); }
inline const DecImpl *impl(const FLAC__StreamDecoder *d) { return reinterpret_cast<const DecImpl *>(d); }

// Drop the first n bytes of buf (they are consumed); the last eight dropped stay readable for the replay of libFLAC's reader.
void drop_front(DecImpl *d, size_t n)
{
    if (n == 0) return;
    uint8_t t[16];
    uint32_t k = 0;
    if (n >= 8) { memcpy(d->pre, d->buf.data() + n - 8, 8); d->npre = 8; }
    else {
        const uint32_t keep = d->npre < 8 - (uint32_t)n ? d->npre : 8 - (uint32_t)n;
        memcpy(t, d->pre + d->npre - keep, keep); k = keep;
        memcpy(t + k, d->buf.data(), n); k += (uint32_t)n;
        memcpy(d->pre, t, k); d->npre = k;
    }
    if (n >= d->buf.size()) d->buf.clear(); else d->buf.erase(d->buf.begin(), d->buf.begin() + n);
    d->consumed_total += n;
}

void reset_stream(DecImpl *d)
{
    d->npre = 0;
    d->buf.clear(); d->consumed_total = 0; d->eof = false; d->have_meta = false; d->have_si = false;
    memset(&d->si, 0, sizeof d->si);
    d->ix = Indexer();
    d->frames_delivered_bound = 0; d->samples_decoded = 0;
    d->frames.clear(); d->status.clear(); d->next_frame = 0; d->last_blocksize = 0; d->last_ca = 0;
    d->do_md5 = d->md5_checking != 0; d->md5.init();
    d->first_pos = 0; d->fixed_blocksize = 0; d->last_set = false; memset(&d->last_hdr, 0, sizeof d->last_hdr);
    d->win = fgref::RefWindows(); d->walker = fgref::Walker(); d->walk_pending = false; d->walk_from = 0; d->ix_origin = 0; d->walk_stuck_at = UINT64_MAX;
    d->pull_want = 1 << 16; d->round_blocks = false;
}

// Pull more bytes.  Returns false on abort.  Sets d->eof at end of stream.  `short_read` reports that the
// callback returned fewer bytes than requested (everything currently available has been delivered).
struct ProfSpan {
    double *acc; std::chrono::steady_clock::time_point t0;
    explicit ProfSpan(double *a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ProfSpan() { *acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1e6; }
};

bool pull(DecImpl *d, bool *short_read)
{
    ProfSpan span(&d->prof_ms[0]);
    // (a client that keeps filling the request is asked for more at a time: 64 KiB .. 16 MiB)
    const size_t want = d->pull_want;
    const size_t old = d->buf.size();
    if (!d->buf.resize(old + want)) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
    size_t got = want;
    *short_read = true;
    if (d->file) {
        got = fread(d->buf.data() + old, 1, want, d->file);
        d->buf.resize(old + got);
        if (got == 0) { d->eof = true; d->win.eof = true; }
        d->win.data_end = d->consumed_total + d->buf.size();
        if (got && got < want) d->win.chunk_end.push_back(d->win.data_end);
        if (got == want && d->pull_want < (16u << 20)) d->pull_want *= 2;
        *short_read = false;
        return true;
    }
    // libFLAC calls the read callback again when it answers CONTINUE with no bytes (bitreader.c refills until it has data
    // or the callback reports end of stream / abort); a client that never produces data spins there too.  The retry is
    // bounded so that a broken callback ends in ABORTED instead of a hang.
    FLAC__StreamDecoderReadStatus rs = FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
    for (uint32_t tries = 0;; tries++) {
        got = want;
        rs = d->read_cb(&d->pub, d->buf.data() + old, &got, d->client);
        if (rs != FLAC__STREAM_DECODER_READ_STATUS_CONTINUE || got != 0) break;
        if (tries >= (1u << 20)) { d->buf.resize(old); d->state = FLAC__STREAM_DECODER_ABORTED; return false; }
        if (tries >= 64) std::this_thread::yield();
    }
    if (rs == FLAC__STREAM_DECODER_READ_STATUS_ABORT) { d->buf.resize(old); d->state = FLAC__STREAM_DECODER_ABORTED; return false; }
    if (got > want) got = want;
    d->buf.resize(old + got);
    // (an answer that fills the request leaves no mark: libFLAC's smaller requests would have been filled as well)
    d->win.data_end = d->consumed_total + d->buf.size();
    if (got && got < want) d->win.chunk_end.push_back(d->win.data_end);
    if (rs == FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM || got == 0) { d->eof = true; d->win.eof = true; }
    *short_read = got < want;
    if (got == want && d->pull_want < (16u << 20)) d->pull_want *= 2;
    return true;
}

// Hand the metadata blocks of d[4 .. end) to the metadata callback, in stream order, filtered the way libFLAC's
// read_metadata_ does (stream_decoder.h:849-973: by type; APPLICATION blocks also by id).  All multi-byte fields are
// big-endian except the VORBIS_COMMENT lengths (format.h:600-880).
void deliver_metadata(DecImpl *d, const uint8_t *buf, uint64_t end)
{
    auto be = [](const uint8_t *p, int n) { uint64_t v = 0; for (int i = 0; i < n; i++) v = (v << 8) | p[i]; return v; };
    auto le32 = [](const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    uint64_t pos = 4;
    while (pos + 4 <= end) {
        const uint32_t last = buf[pos] >> 7, type = buf[pos] & 0x7F;
        const uint32_t len = (uint32_t)be(buf + pos + 1, 3);
        const uint8_t *b = buf + pos + 4;
        pos += 4 + (uint64_t)len;
        if (pos > end) break;
        bool want = d->respond[type];
        if (type == FLAC__METADATA_TYPE_APPLICATION && len >= 4) {
            const uint32_t id = (uint32_t)be(b, 4);
            for (uint32_t x : d->app_ids) if (x == id) { want = !want; break; }
        }
        if (!want) { if (last) break; continue; }
        FLAC__StreamMetadata m;
        memset(&m, 0, sizeof m);
        m.type = (FLAC__MetadataType)type; m.is_last = (FLAC__bool)last; m.length = len;
        // owned storage of the variable-size members; lives until the callback returns
        std::vector<std::vector<uint8_t>> blobs;
        auto keep = [&](const uint8_t *p, size_t n, bool nul) -> FLAC__byte * {
            blobs.emplace_back(p, p + n);
            if (nul) blobs.back().push_back(0);
            if (blobs.back().empty()) blobs.back().push_back(0);
            return blobs.back().data();
        };
        std::vector<FLAC__StreamMetadata_SeekPoint> points;
        std::vector<FLAC__StreamMetadata_VorbisComment_Entry> comments;
        std::vector<FLAC__StreamMetadata_CueSheet_Track> tracks;
        std::vector<std::vector<FLAC__StreamMetadata_CueSheet_Index>> indices;
        bool ok = true;
        switch (type) {
        case FLAC__METADATA_TYPE_STREAMINFO:
            if (len < 34 || !d->have_si) ok = false;
            else m.data.stream_info = d->si;
            break;
        case FLAC__METADATA_TYPE_PADDING:
            break;
        case FLAC__METADATA_TYPE_APPLICATION:
            if (len < 4) { ok = false; break; }
            memcpy(m.data.application.id, b, 4);
            m.data.application.data = len > 4 ? keep(b + 4, len - 4, false) : nullptr;
            break;
        case FLAC__METADATA_TYPE_SEEKTABLE: {
            const uint32_t np = len / 18;
            points.resize(np);
            for (uint32_t i = 0; i < np; i++) {
                points[i].sample_number = be(b + 18 * i, 8);
                points[i].stream_offset = be(b + 18 * i + 8, 8);
                points[i].frame_samples = (uint32_t)be(b + 18 * i + 16, 2);
            }
            m.data.seek_table.num_points = np;
            m.data.seek_table.points = np ? points.data() : nullptr;
            break;
        }
        case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
            // libFLAC tolerates truncated blocks by dropping what does not fit (stream_decoder.c read_metadata_vorbiscomment_)
            uint32_t o = 0;
            FLAC__StreamMetadata_VorbisComment &vc = m.data.vorbis_comment;
            if (len >= 8 && le32(b) <= len - 8) {
                vc.vendor_string.length = le32(b);
                vc.vendor_string.entry = keep(b + 4, vc.vendor_string.length, true);
                o = 4 + vc.vendor_string.length;
                uint32_t nc = le32(b + o);
                o += 4;
                if (nc > 100000) nc = 0;
                for (uint32_t i = 0; i < nc; i++) {
                    if (o + 4 > len) break;
                    const uint32_t l = le32(b + o);
                    if (l > len - o - 4) break;
                    FLAC__StreamMetadata_VorbisComment_Entry e;
                    e.length = l; e.entry = keep(b + o + 4, l, true);
                    comments.push_back(e);
                    o += 4 + l;
                }
                vc.num_comments = (uint32_t)comments.size();
                vc.comments = comments.empty() ? nullptr : comments.data();
            }
            break;
        }
        case FLAC__METADATA_TYPE_CUESHEET: {
            if (len < 396) { ok = false; break; }
            FLAC__StreamMetadata_CueSheet &cs = m.data.cue_sheet;
            memcpy(cs.media_catalog_number, b, 128);
            cs.media_catalog_number[128] = 0;
            cs.lead_in = be(b + 128, 8);
            cs.is_cd = b[136] >> 7;
            const uint32_t nt = b[395];
            uint32_t o = 396;
            tracks.resize(nt);
            indices.resize(nt);
            for (uint32_t t = 0; t < nt && ok; t++) {
                if (o + 36 > len) { ok = false; break; }
                FLAC__StreamMetadata_CueSheet_Track &tr = tracks[t];
                memset(&tr, 0, sizeof tr);
                tr.offset = be(b + o, 8);
                tr.number = b[o + 8];
                memcpy(tr.isrc, b + o + 9, 12);
                tr.isrc[12] = 0;
                tr.type = b[o + 21] >> 7;
                tr.pre_emphasis = (b[o + 21] >> 6) & 1;
                tr.num_indices = b[o + 35];
                o += 36;
                indices[t].resize(tr.num_indices);
                for (uint32_t k = 0; k < tr.num_indices; k++) {
                    if (o + 12 > len) { ok = false; break; }
                    indices[t][k].offset = be(b + o, 8);
                    indices[t][k].number = b[o + 8];
                    o += 12;
                }
                tr.indices = tr.num_indices ? indices[t].data() : nullptr;
            }
            cs.num_tracks = nt;
            cs.tracks = nt ? tracks.data() : nullptr;
            break;
        }
        case FLAC__METADATA_TYPE_PICTURE: {
            FLAC__StreamMetadata_Picture &pc = m.data.picture;
            uint32_t o = 0;
            auto need = [&](uint32_t nbytes) { if ((uint64_t)o + nbytes > len) { ok = false; return false; } return true; };
            if (!need(8)) break;
            pc.type = (FLAC__StreamMetadata_Picture_Type)be(b, 4);
            uint32_t l = (uint32_t)be(b + 4, 4);
            o = 8;
            if (!need(l)) break;
            pc.mime_type = (char *)keep(b + o, l, true);
            o += l;
            if (!need(4)) break;
            l = (uint32_t)be(b + o, 4);
            o += 4;
            if (!need(l)) break;
            pc.description = keep(b + o, l, true);
            o += l;
            if (!need(20)) break;
            pc.width = (uint32_t)be(b + o, 4); pc.height = (uint32_t)be(b + o + 4, 4);
            pc.depth = (uint32_t)be(b + o + 8, 4); pc.colors = (uint32_t)be(b + o + 12, 4);
            pc.data_length = (uint32_t)be(b + o + 16, 4);
            o += 20;
            if (!need(pc.data_length)) break;
            pc.data = pc.data_length ? keep(b + o, pc.data_length, false) : nullptr;
            break;
        }
        default:
            m.data.unknown.data = len ? keep(b, len, false) : nullptr;
            break;
        }
        if (ok) d->meta_cb(&d->pub, &m, d->client);
        if (last) break;
    }
}

bool ensure_metadata(DecImpl *d)
{
    while (!d->have_meta) {
        const int64_t a = parse_metadata(d->buf.data(), d->buf.size(), &d->si, &d->have_si);
        if (a < 0) {
            // libFLAC keeps searching for "fLaC" and reports LOST_SYNC; a stream that never shows it ends in error
            if (d->error_cb) d->error_cb(&d->pub, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, d->client);
            drop_front(d, d->buf.size());
            if (d->eof) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return false; }
            if (d->state == FLAC__STREAM_DECODER_ABORTED) return false;
            bool sr;
            if (!pull(d, &sr)) return false;
            if (d->eof && d->buf.empty()) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return false; }
            continue;
        }
        if (a == 0) {
            if (d->eof) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return false; }
            bool sr;
            if (!pull(d, &sr)) return false;
            continue;
        }
        d->have_meta = true;
        d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
        if (d->meta_cb) deliver_metadata(d, d->buf.data(), (uint64_t)a);
        drop_front(d, (size_t)a);
        d->audio_start = d->consumed_total;
    }
    return true;
}

// Decode every complete frame currently delimited in d->buf on the GPU and queue them for delivery.
bool decode_available(DecImpl *d)
{
    const size_t nb = d->ix.bounds.size();
    if (nb < 2 || d->frames_delivered_bound + 1 >= nb) return true;
    const uint32_t nframes = (uint32_t)(nb - 1 - d->frames_delivered_bound);
    const uint64_t first = d->ix.bounds[d->frames_delivered_bound], last = d->ix.bounds[nb - 1];
    flacgpu_ctx *c = d->ctx;
    (void)hipSetDevice(c->device);
    // FLACGPU_API_PROF=1: wall time of the phases of this call on stderr (tuning aid)
    static const bool api_prof = fg_tune("FLACGPU_API_PROF") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t) {
        if (!api_prof) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[flacgpu api prof] decode_available %-10s %8.3f ms (%u frames)\n", what,
                std::chrono::duration_cast<std::chrono::microseconds>(now - t).count() / 1000.0, nframes);
        t = now;
    };
    auto tl = tp0;
    std::vector<uint64_t> offs(nframes + 1);
    for (uint32_t i = 0; i <= nframes; i++) offs[i] = d->ix.bounds[d->frames_delivered_bound + i] - first;
    if (!dev_ensure(d->d_stream, (size_t)(last - first) + 64, c->device)) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
    if (!HIPOK(hipMemcpy(d->d_stream.p, d->buf.data() + first, (size_t)(last - first), hipMemcpyHostToDevice))) {
        d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false;
    }
    lap("upload", tl);
    const uint32_t C = d->have_si ? d->si.channels : 0;
    // upper bound of the sample count: 65535 per frame is wasteful; use the STREAMINFO max block size when known
    uint64_t cap = (uint64_t)nframes * ((d->have_si && d->si.max_blocksize) ? d->si.max_blocksize : 65535);
    const uint32_t Cb = C ? C : 8;
    if (!dev_ensure(d->d_pcm, (size_t)cap * Cb * 4, c->device)) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
    flacgpu_decode_stats st;
    std::vector<FgDecResult> status(nframes);
    std::vector<FgDecFrame> frames;
    d->detail.level = d->subframe_detail;
    // block delivery: channels interleaved (one buffer for the whole round), no per-subframe records
    const bool blocks = d->block_cb != nullptr && !d->do_md5 && C != 0;
    DecDetail *const detail = (C && d->subframe_detail > 0 && !blocks) ? &d->detail : nullptr;
    bool ok = decode_frames_impl(c, d->d_stream.p, last - first, offs.data(), nframes, C, d->have_si ? d->si.bits_per_sample : 0, d->d_pcm.p,
                                 cap, blocks ? 1 : 0, status.data(), &frames, &st, false, 0, nullptr, detail);
    if (!ok && st.total_samples > cap && st.total_samples <= (uint64_t)nframes * 65535) {
        // A frame header that names a larger block than STREAMINFO's maximum (damage that the CRC-8 let through, or a stream
        // that lies about itself): libFLAC decodes such a frame all the same.  Once more with room for what the headers say.
        cap = st.total_samples;
        if (!dev_ensure(d->d_pcm, (size_t)cap * Cb * 4, c->device)) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
        ok = decode_frames_impl(c, d->d_stream.p, last - first, offs.data(), nframes, C, d->have_si ? d->si.bits_per_sample : 0, d->d_pcm.p,
                                cap, blocks ? 1 : 0, status.data(), &frames, &st, false, 0, nullptr, detail);
    }
    if (!ok) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
    lap("decode", tl);
    const size_t npcm = (size_t)st.total_samples * (C ? C : 2);
    if (!d->pcm.ensure(npcm)) { d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false; }
    lap("buffer", tl);
    d->round_blocks = blocks; d->round_bytes = 4;
    if (blocks) {
        bool narrow = true;
        for (const FgDecFrame &fr : frames) if (fr.bps > 16) { narrow = false; break; }
        if (narrow && npcm) {
            if (!dev_ensure(d->d_pcm16, npcm * 2 + 16, c->device) || fg_launch_narrow16((const int32_t *)d->d_pcm.p, (int16_t *)d->d_pcm16.p, npcm, c->stream) != 0 ||
                !HIPOK(hipMemcpyAsync(d->pcm.data(), d->d_pcm16.p, npcm * 2, hipMemcpyDeviceToHost, c->stream)) || !HIPOK(hipStreamSynchronize(c->stream))) {
                d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false;
            }
            d->round_bytes = 2;
        }
    }
    if (d->round_bytes == 4 && st.total_samples && !HIPOK(hipMemcpy(d->pcm.data(), d->d_pcm.p, npcm * 4, hipMemcpyDeviceToHost))) {
        d->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR; return false;
    }
    lap("download", tl);
    d->frames.swap(frames);
    d->status.swap(status);
    d->next_frame = 0;
    d->first_pos = first;
    d->frames_delivered_bound = nb - 1;
    return true;
}

// Hand one frame to the write callback.  Returns false when the client aborted.
bool write_frame(DecImpl *d, const FLAC__Frame &f, const int32_t *const chan[])
{
    d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
    if (!d->have_si) d->do_md5 = false;
    if (d->do_md5) {
        // signature input: interleaved, little-endian, (bps+7)/8 bytes per sample (format.h:549 md5sum)
        const uint32_t C = f.header.channels, n = f.header.blocksize;
        d->md5_tmp.resize((size_t)n * C);
        for (uint32_t c = 0; c < C; c++) {
            const int32_t *src = chan[c];
            int32_t *dst = d->md5_tmp.data() + c;
            for (uint32_t i = 0; i < n; i++) dst[(size_t)i * C] = src[i];
        }
        d->md5.update_pcm(d->md5_tmp.data(), (uint64_t)n * C, f.header.bits_per_sample);
    }
    if (d->write_cb(&d->pub, &f, chan, d->client) != FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE) {
        d->state = FLAC__STREAM_DECODER_ABORTED;
        return false;
    }
    return true;
}

// Deliver one queued frame.  Returns false when the client aborted.
//
// Damage handling is libFLAC 1.4.3's, callback for callback (tests/golden/damage_vectors.json, recorded from the reference's
// binary): fill_queue ends the queue in front of the first place the index or the GPU pass rejects and replays libFLAC's serial
// reader from there on the host (walk_damage, fg_refwalk.h) -- which error statuses it reports, in which order, and which frame it
// decodes next.  A damaged frame is NOT delivered; when the next good frame's sample number shows a gap behind the last delivered
// frame, frames of silence with the last frame's header fill it (at most 5 s / 50 frames, only between frames of the same
// format), so the time line is kept.  Nothing is filled before the first delivered frame or after the last one.
bool deliver_one(DecImpl *d)
{
    const FgDecFrame &fr = d->frames[d->next_frame];
    const FgDecResult &rs = d->status[d->next_frame];
    d->next_frame++;
    const uint64_t fpos = d->first_pos + fr.byte_off;
    if (rs.err != 0 && rs.err != 3) {
        // (not reached: fill_queue ends the queue in front of a frame the GPU pass rejects and replays libFLAC's reader from there)
        if (d->error_cb) d->error_cb(&d->pub, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, d->client);
        return true;
    }
    FLAC__Frame f;
    memset(&f, 0, sizeof f);
    f.header.blocksize = fr.n;
    f.header.sample_rate = d->si.sample_rate;
    // the sample number comes from the frame's own header (it stays right across frames lost to damage)
    HostHeader hh;
    uint64_t number = d->samples_decoded;
    if (fpos < d->buf.size() && parse_header(d->buf.data() + fpos, d->buf.size() - fpos, d->have_si ? &d->si : nullptr, &hh)) {
        if (hh.sample_rate) f.header.sample_rate = hh.sample_rate;
        if (hh.variable) number = hh.number;
        else {
            const uint32_t fixed = (d->have_si && d->si.min_blocksize == d->si.max_blocksize && d->si.min_blocksize) ? d->si.min_blocksize
                                   : (d->fixed_blocksize ? d->fixed_blocksize : fr.n);
            if (!d->fixed_blocksize) d->fixed_blocksize = fixed;
            number = hh.number * (uint64_t)fixed;
        }
    }
    f.header.channels = fr.channels;
    f.header.channel_assignment = (FLAC__ChannelAssignment)fr.ca;
    f.header.bits_per_sample = fr.bps;
    f.header.number_type = FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER;
    f.header.number.sample_number = number;
    f.footer.crc = (FLAC__uint16)rs.crc;
    if (d->last_set && d->last_hdr.number.sample_number + d->last_hdr.blocksize < number &&
        d->last_hdr.sample_rate == f.header.sample_rate && d->last_hdr.channels == f.header.channels &&
        d->last_hdr.bits_per_sample == f.header.bits_per_sample && d->last_hdr.blocksize >= 16) {
        uint64_t need = number - (d->last_hdr.number.sample_number + d->last_hdr.blocksize);
        FLAC__Frame e;
        memset(&e, 0, sizeof e);
        e.header = d->last_hdr;
        if (need > 5ull * e.header.sample_rate) need = 5ull * e.header.sample_rate;
        if (need > 50ull * e.header.blocksize) need = 50ull * e.header.blocksize;
        d->silence.assign(e.header.blocksize, 0);
        const int32_t *zc[8];
        for (uint32_t c = 0; c < 8; c++) zc[c] = c < e.header.channels ? d->silence.data() : nullptr;
        for (uint32_t c = 0; c < e.header.channels; c++) e.subframes[c].type = FLAC__SUBFRAME_TYPE_CONSTANT;
        while (need) {
            e.header.number.sample_number += e.header.blocksize;
            if (need < e.header.blocksize) e.header.blocksize = (uint32_t)need;
            need -= e.header.blocksize;
            d->samples_decoded = e.header.number.sample_number + e.header.blocksize;
            if (!write_frame(d, e, zc)) return false;
        }
    }
    const int32_t *chan[8];
    for (uint32_t c = 0; c < 8; c++) chan[c] = c < fr.channels ? d->pcm.data() + (size_t)fr.out_off * fr.channels + (size_t)c * fr.n : nullptr;
    if (d->subframe_detail > 0) fill_subframes(d, f, fr, d->next_frame - 1);
    d->last_blocksize = fr.n; d->last_ca = fr.ca;
    d->last_hdr = f.header; d->last_set = true;
    d->samples_decoded = number + fr.n;
    if (d->seeking) {
        // FLAC__stream_decoder_seek_absolute (stream_decoder.h:1412-1450; the reference binary, tests/test_gpu_api.py TestSeek): frames
        // in front of the target are not delivered; the frame that holds it is delivered from the target sample on, under the
        // target's sample number
        if (number + fr.n <= d->seek_target) return true;
        d->seeking = false; d->seek_done = true;
        if (number < d->seek_target) {
            const uint32_t skip = (uint32_t)(d->seek_target - number);
            for (uint32_t c = 0; c < fr.channels; c++) chan[c] += skip;
            f.header.blocksize = fr.n - skip;
            f.header.number.sample_number = d->seek_target;
        }
    }
    return write_frame(d, f, chan);
}

// Block delivery: up to `limit` queued frames in one call of the block callback (the same headers, numbering and gap filling as
// deliver_one).
bool deliver_blocks(DecImpl *d, size_t limit)
{
    d->blocks.clear();
    const size_t left = d->frames.size() - d->next_frame;
    const size_t end = d->next_frame + std::min(left, limit);
    for (; d->next_frame < end; d->next_frame++) {
        const FgDecFrame &fr = d->frames[d->next_frame];
        const FgDecResult &rs = d->status[d->next_frame];
        if (rs.err != 0 && rs.err != 3) continue;          // (not reached, see deliver_one)
        const uint64_t fpos = d->first_pos + fr.byte_off;
        FLAC__FrameHeader h;
        memset(&h, 0, sizeof h);
        h.blocksize = fr.n; h.sample_rate = d->si.sample_rate;
        HostHeader hh;
        uint64_t number = d->samples_decoded;
        if (fpos < d->buf.size() && parse_header(d->buf.data() + fpos, d->buf.size() - fpos, d->have_si ? &d->si : nullptr, &hh)) {
            if (hh.sample_rate) h.sample_rate = hh.sample_rate;
            if (hh.variable) number = hh.number;
            else {
                const uint32_t fixed = (d->have_si && d->si.min_blocksize == d->si.max_blocksize && d->si.min_blocksize) ? d->si.min_blocksize
                                       : (d->fixed_blocksize ? d->fixed_blocksize : fr.n);
                if (!d->fixed_blocksize) d->fixed_blocksize = fixed;
                number = hh.number * (uint64_t)fixed;
            }
        }
        h.channels = fr.channels; h.channel_assignment = (FLAC__ChannelAssignment)fr.ca; h.bits_per_sample = fr.bps;
        h.number_type = FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER; h.number.sample_number = number;
        if (d->last_set && d->last_hdr.number.sample_number + d->last_hdr.blocksize < number && d->last_hdr.sample_rate == h.sample_rate &&
            d->last_hdr.channels == h.channels && d->last_hdr.bits_per_sample == h.bits_per_sample && d->last_hdr.blocksize >= 16) {
            uint64_t need = number - (d->last_hdr.number.sample_number + d->last_hdr.blocksize);
            FLAC__FrameHeader e = d->last_hdr;
            if (need > 5ull * e.sample_rate) need = 5ull * e.sample_rate;
            if (need > 50ull * e.blocksize) need = 50ull * e.blocksize;
            while (need) {
                e.number.sample_number += e.blocksize;
                if (need < e.blocksize) e.blocksize = (uint32_t)need;
                need -= e.blocksize;
                d->blocks.push_back(flacgpu_block{e.number.sample_number, FLACGPU_BLOCK_SILENCE, e.blocksize, e.channels, e.bits_per_sample, e.sample_rate});
            }
        }
        if (d->seeking) {
            // (FLAC__stream_decoder_seek_absolute, see deliver_one)
            d->last_blocksize = fr.n; d->last_ca = fr.ca; d->last_hdr = h; d->last_set = true; d->samples_decoded = number + fr.n;
            if (number + fr.n <= d->seek_target) continue;
            d->seeking = false; d->seek_done = true;
            const uint32_t skip = number < d->seek_target ? (uint32_t)(d->seek_target - number) : 0u;
            d->blocks.push_back(flacgpu_block{number + skip, fr.out_off + skip, fr.n - skip, fr.channels, fr.bps, h.sample_rate});
            continue;
        }
        d->blocks.push_back(flacgpu_block{number, fr.out_off, fr.n, fr.channels, fr.bps, h.sample_rate});
        d->last_blocksize = fr.n; d->last_ca = fr.ca;
        d->last_hdr = h; d->last_set = true;
        d->samples_decoded = number + fr.n;
    }
    if (!d->blocks.empty()) {
        d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
        if (d->block_cb(&d->pub, d->blocks.data(), (uint32_t)d->blocks.size(), d->pcm.data(), d->round_bytes, d->client) != FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE) {
            d->state = FLAC__STREAM_DECODER_ABORTED;
            return false;
        }
    }
    return true;
}

// FLAC__Frame.subframes[] of a delivered frame (format.h:285-396, pyflac/builder/decoder.py:146-231): what the parse kernel
// recorded per subframe.  Pointers stay valid until the next frame is delivered.
void fill_subframes(DecImpl *d, FLAC__Frame &f, const FgDecFrame &fr, uint32_t fi)
{
    const uint32_t C = fr.channels;
    if ((size_t)(fi + 1) * C > d->detail.subs.size()) return;
    for (uint32_t ch = 0; ch < C && ch < 8; ch++) {
        const FgDecSub &sd = d->detail.subs[(size_t)fi * C + ch];
        FLAC__Subframe &sf = f.subframes[ch];
        if (!(sd.flags & (1u << 12))) continue;                 // a frame of the generic decoder: no record
        const uint32_t type = sd.flags & 3, prec = (sd.flags >> 2) & 31, po = (sd.flags >> 7) & 15, method = (sd.flags >> 11) & 1;
        // (Rice parameters are kept for the first 256 partitions: a non-subset stream with a partition order above 8 gets no
        // subframe details rather than details that end in zeros -- include/flacgpu.h says so)
        if (type >= 2 && po > 8) continue;
        const int32_t *plane = d->detail.planes.empty() ? nullptr : d->detail.planes.data() + (size_t)fr.out_off * C + (size_t)ch * fr.n;
        const int32_t *warm = d->detail.warm.data() + ((size_t)fi * C + ch) * 32;
        sf.wasted_bits = sd.wasted;
        auto rice = [&](FLAC__EntropyCodingMethod &ecm) {
            ecm.type = method ? FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 : FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE;
            ecm.data.partitioned_rice.order = po;
            const uint32_t np = 1u << po;
            d->rice_prm[ch].assign(np, 0); d->rice_raw[ch].assign(np, 0);
            const uint16_t *rp = d->detail.rparams.data() + ((size_t)fi * C + ch) * FG_DEC_RPARAMS;
            for (uint32_t p = 0; p < np && p < FG_DEC_RPARAMS; p++) {
                if (rp[p] & 0x8000) { d->rice_prm[ch][p] = method ? 31 : 15; d->rice_raw[ch][p] = (rp[p] >> 8) & 31; }
                else d->rice_prm[ch][p] = rp[p] & 31;
            }
            d->rice_contents[ch].parameters = d->rice_prm[ch].data();
            d->rice_contents[ch].raw_bits = d->rice_raw[ch].data();
            d->rice_contents[ch].capacity_by_order = po;
            ecm.data.partitioned_rice.contents = &d->rice_contents[ch];
        };
        switch (type) {
        case 0:
            sf.type = FLAC__SUBFRAME_TYPE_CONSTANT;
            sf.data.constant.value = sd.q[0];
            break;
        case 1:
            sf.type = FLAC__SUBFRAME_TYPE_VERBATIM;
            sf.data.verbatim.data.int32 = plane;               // null unless the sample arrays were asked for (detail level 2)
            sf.data.verbatim.data_type = FLAC__VERBATIM_SUBFRAME_DATA_TYPE_INT32;
            break;
        case 2:
            sf.type = FLAC__SUBFRAME_TYPE_FIXED;
            sf.data.fixed.order = sd.order;
            for (uint32_t j = 0; j < sd.order && j < 4; j++) sf.data.fixed.warmup[j] = warm[j];
            sf.data.fixed.residual = plane ? plane + sd.order : nullptr;
            rice(sf.data.fixed.entropy_coding_method);
            break;
        default:
            sf.type = FLAC__SUBFRAME_TYPE_LPC;
            sf.data.lpc.order = sd.order;
            sf.data.lpc.qlp_coeff_precision = prec;
            sf.data.lpc.quantization_level = sd.shift;
            for (uint32_t j = 0; j < sd.order && j < 12; j++) { sf.data.lpc.qlp_coeff[j] = sd.q[j]; sf.data.lpc.warmup[j] = warm[j]; }
            sf.data.lpc.residual = plane ? plane + sd.order : nullptr;
            rice(sf.data.lpc.entropy_coding_method);
            break;
        }
    }
}

// Replay libFLAC's reader from d->walk_from (fg_refwalk.h): report what it reports, and let the index take over at the frame it
// decodes next.  Returns false when the stream ended (or the client aborted).
bool walk_damage(DecImpl *d)
{
    std::vector<uint32_t> errs;
    fgref::Header h;
    bool ended = false;
    uint64_t s = 0, fend = 0;
    fgref::Walker w;
    for (;;) {
        errs.clear();
        w = d->walker;
        w.d = d->buf.data(); w.len = d->buf.size(); w.abs0 = d->consumed_total; w.final = d->eof; w.win = &d->win;
        w.npre = d->npre; memcpy(w.pre, d->pre, 8);
        w.si.have = d->have_si; w.si.min_blocksize = d->si.min_blocksize; w.si.max_blocksize = d->si.max_blocksize;
        w.si.sample_rate = d->si.sample_rate; w.si.channels = d->si.channels; w.si.bps = d->si.bits_per_sample;
        w.si.total_samples = d->si.total_samples;
        w.fixed_blocksize = d->fixed_blocksize;
        s = w.run(d->walk_from, errs, &h, &ended, &fend);
        if (s != UINT64_MAX) break;
        // the reader would have read on: more data (a damaged frame can parse on for megabytes)
        bool short_read = false;
        const size_t had = d->buf.size();
        if (!pull(d, &short_read)) return false;
        while (!short_read && !d->eof && d->buf.size() - had < (1u << 20)) if (!pull(d, &short_read)) return false;
    }
    d->walker = w;
    d->walk_pending = false;
    if (fg_tune("FG_REFWALK_DEBUG"))
        fprintf(stderr, "walk_damage: from %llu (abs %llu) -> s=%llu fend=%llu ended=%d errs=%zu buf=%zu eof=%d\n", (unsigned long long)d->walk_from,
                (unsigned long long)(d->consumed_total + d->walk_from), (unsigned long long)s, (unsigned long long)fend, (int)ended, errs.size(), d->buf.size(), (int)d->eof);
    if (d->error_cb) for (uint32_t e : errs) d->error_cb(&d->pub, (FLAC__StreamDecoderErrorStatus)e, d->client);
    if (!ended && errs.empty() && d->consumed_total + s == d->walk_stuck_at) {
        // (not reached: a frame libFLAC's rules accept and the GPU pass rejects.  Report it and go on behind it.)
        if (d->error_cb) d->error_cb(&d->pub, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM, d->client);
        d->walk_pending = true; d->walk_from = fend; d->walk_stuck_at = UINT64_MAX;
        return true;
    }
    d->walk_stuck_at = ended ? UINT64_MAX : d->consumed_total + s;
    d->frames.clear(); d->status.clear(); d->next_frame = 0; d->frames_delivered_bound = 0;
    d->ix = Indexer(); d->ix.fast = false;
    if (ended) {
        drop_front(d, d->buf.size()); d->ix_origin = 0;
        d->state = FLAC__STREAM_DECODER_END_OF_STREAM;
        return false;
    }
    // the frame [s, fend) decodes: it is the next one the GPU pass sees, and the index goes on behind it
    drop_front(d, (size_t)s);
    d->ix.bounds.push_back(0); d->ix.bounds.push_back(fend - s);
    d->ix.scan = fend - s; d->ix.in_frame = false;
    d->ix_origin = fend - s;
    return true;
}

// Make progress: after this call either at least one frame is queued, or the stream has ended / aborted.
bool fill_queue(DecImpl *d)
{
    for (;;) {
        // (libFLAC 1.4.3 does NOT end a stream where STREAMINFO's sample count is reached -- the reference binary delivers every
        // frame of a stream that claims fewer and reports LOST_SYNC for bytes behind the last frame, in stream and in file mode:
        // tests/test_gpu_api.py::TestStreamEnd; round 3 had such a test here, on one of three delivery paths)
        if (d->next_frame < d->frames.size()) return true;
        if (!ensure_metadata(d)) return false;
        if (d->walk_pending) {
            if (!walk_damage(d)) return false;
            if (d->walk_pending) continue;
        }
        // drop the bytes of delivered frames
        if (d->frames_delivered_bound > 0 && !d->ix.bounds.empty()) {
            const uint64_t cut = d->ix.bounds[d->frames_delivered_bound];
            if (cut > 0) {
                { ProfSpan span(&d->prof_ms[4]); drop_front(d, (size_t)cut); }
                std::vector<uint64_t> nbnd;
                for (size_t i = d->frames_delivered_bound; i < d->ix.bounds.size(); i++) nbnd.push_back(d->ix.bounds[i] - cut);
                d->ix.bounds.swap(nbnd);
                d->ix.scan -= cut; d->ix.frame_start -= cut; d->ix.crc_pos -= cut;
                for (auto &p : d->ix.error_pos) p = p >= cut ? p - cut : 0;
                d->ix_origin = d->ix_origin >= cut ? d->ix_origin - cut : 0;
                d->frames_delivered_bound = 0;
            }
        }
        d->frames.clear(); d->status.clear(); d->next_frame = 0;
        // Index the new bytes -- in fast mode first (see Indexer::fast), with a snapshot to return to: if the index or the GPU
        // pass finds anything irregular, the round is repeated in careful mode (and the decoder stays there).
        Indexer snap;
        const bool try_fast = d->ix.fast;
        if (try_fast) snap = d->ix;
        const uint64_t fdb = d->frames_delivered_bound;
        for (int attempt = 0; attempt < 2; attempt++) {
            const auto tf0 = std::chrono::steady_clock::now();
            { ProfSpan span(&d->prof_ms[1]); d->ix.feed(d->buf.data(), d->buf.size(), d->eof, d->have_si ? &d->si : nullptr); }
            if (fg_tune("FLACGPU_API_PROF"))
                fprintf(stderr, "[flacgpu api prof] index feed %8.3f ms (%zu bytes, %s)\n",
                        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tf0).count() / 1000.0, d->buf.size(),
                        d->ix.fast ? "fast" : "careful");
            bool redo = d->ix.fast && (d->ix.fast_failed || !d->ix.errors.empty());
            if (!redo && d->ix.bounds.size() >= 2 && d->frames_delivered_bound + 1 < d->ix.bounds.size()) {
                { ProfSpan span(&d->prof_ms[2]); if (!decode_available(d)) return false; }
                if (d->ix.fast) for (const FgDecResult &r : d->status) if (r.err != 0 && r.err != 3) { redo = true; break; }
            }
            if (!redo) break;
            d->ix = snap; d->ix.fast = false; d->ix.fast_failed = false;
            d->frames.clear(); d->status.clear(); d->next_frame = 0; d->frames_delivered_bound = fdb;
        }
        // Anything irregular: bytes that are no frame where the index looked for one (from ix_origin on), or a frame the GPU pass
        // rejects.  The frames before that place are delivered, then libFLAC's reader is replayed from there.
        {
            uint64_t trouble = UINT64_MAX;
            if (!d->ix.errors.empty()) trouble = d->ix_origin;
            for (size_t i = 0; i < d->frames.size(); i++)
                if (d->status[i].err != 0 && d->status[i].err != 3) { trouble = std::min<uint64_t>(trouble, d->first_pos + d->frames[i].byte_off); break; }
            if (trouble != UINT64_MAX && fg_tune("FG_REFWALK_DEBUG"))
                fprintf(stderr, "fill_queue: trouble at %llu (abs %llu), %zu frames in the round, index errors %zu, origin %llu\n", (unsigned long long)trouble,
                        (unsigned long long)(d->consumed_total + trouble), d->frames.size(), d->ix.errors.size(), (unsigned long long)d->ix_origin);
            if (trouble != UINT64_MAX) {
                size_t keep = 0;
                while (keep < d->frames.size() && d->first_pos + d->frames[keep].byte_off < trouble) keep++;       // (both are frame boundaries)
                d->frames.resize(keep); d->status.resize(keep);
                d->walk_pending = true; d->walk_from = trouble;
            }
        }
        if (!d->frames.empty()) return true;
        if (d->walk_pending) continue;
        if (d->eof) { d->state = FLAC__STREAM_DECODER_END_OF_STREAM; return false; }
        // need more data; keep pulling while the callback keeps filling the request (drains what is available)
        bool short_read = false;
        if (!pull(d, &short_read)) return false;
        while (!short_read && !d->eof && d->buf.size() < (64u << 20)) {
            if (!pull(d, &short_read)) return false;
        }
    }
}

FLAC__StreamDecoderInitStatus init_common(DecImpl *d)
{
    d->ctx = fg_default_ctx();
    if (!d->ctx) return FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR;
    reset_stream(d);
    d->state = FLAC__STREAM_DECODER_SEARCH_FOR_METADATA;
    return FLAC__STREAM_DECODER_INIT_STATUS_OK;
}

}  // namespace

extern "C" {

FLAC__StreamDecoder *FLAC__stream_decoder_new(void)
{
    DecImpl *d = new DecImpl();
    d->pub.protected_ = nullptr; d->pub.private_ = nullptr;
    d->state = FLAC__STREAM_DECODER_UNINITIALIZED;
    d->md5_checking = 0; d->read_cb = nullptr; d->write_cb = nullptr; d->meta_cb = nullptr; d->error_cb = nullptr;
    d->client = nullptr; d->file = nullptr; d->own_file = false; d->ctx = nullptr;
    memset(d->respond, 0, sizeof d->respond); d->respond[FLAC__METADATA_TYPE_STREAMINFO] = true; d->app_ids.clear();
    reset_stream(d);
    return &d->pub;
}

void FLAC__stream_decoder_delete(FLAC__StreamDecoder *dec)
{
    if (!dec) return;
    DecImpl *d = impl(dec);
    if (d->file && d->own_file) fclose(d->file);
    if (d->ctx) (void)hipSetDevice(d->ctx->device);
    d->d_stream.release(); d->d_pcm.release(); d->d_pcm16.release();
    d->pcm.give_back();
    delete d;
}

// Extension: how much of FLAC__Frame.subframes[] the write callback sees.  0: nothing (type fields stay zero), 1 (default):
// type, wasted bits, order, precision, shift, coefficients, warm-up, partition order and Rice parameters, 2: also the
// `residual` / verbatim `data` sample arrays (one more device-to-host copy of the size of the PCM).
FLAC__bool flacgpu_stream_decoder_set_block_callback(FLAC__StreamDecoder *dec, flacgpu_block_callback callback)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    d->block_cb = callback;
    return 1;
}

void flacgpu_stream_decoder_set_subframe_detail(FLAC__StreamDecoder *dec, int level) { impl(dec)->subframe_detail = level < 0 ? 0 : level > 2 ? 2 : level; }

FLAC__bool FLAC__stream_decoder_set_md5_checking(FLAC__StreamDecoder *dec, FLAC__bool value)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    d->md5_checking = value;
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *dec, FLAC__MetadataType type)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED || (uint32_t)type > FLAC__MAX_METADATA_TYPE) return 0;
    d->respond[type] = true;
    if (type == FLAC__METADATA_TYPE_APPLICATION) d->app_ids.clear();
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *dec, const FLAC__byte id[4])
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED || !id) return 0;
    if (d->respond[FLAC__METADATA_TYPE_APPLICATION]) return 1;          // already responding to every id
    d->app_ids.push_back(((uint32_t)id[0] << 24) | ((uint32_t)id[1] << 16) | ((uint32_t)id[2] << 8) | id[3]);
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    for (bool &r : d->respond) r = true;
    d->app_ids.clear();
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *dec, FLAC__MetadataType type)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED || (uint32_t)type > FLAC__MAX_METADATA_TYPE) return 0;
    d->respond[type] = false;
    if (type == FLAC__METADATA_TYPE_APPLICATION) d->app_ids.clear();
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *dec, const FLAC__byte id[4])
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED || !id) return 0;
    if (!d->respond[FLAC__METADATA_TYPE_APPLICATION]) return 1;         // already ignoring every id
    d->app_ids.push_back(((uint32_t)id[0] << 24) | ((uint32_t)id[1] << 16) | ((uint32_t)id[2] << 8) | id[3]);
    return 1;
}
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    for (bool &r : d->respond) r = false;
    d->app_ids.clear();
    return 1;
}

FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *dec) { return impl(dec)->state; }
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *dec) { return FLAC__StreamDecoderStateString[impl(dec)->state]; }
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *dec) { return impl(dec)->md5_checking; }
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *dec) { return impl(dec)->have_si ? impl(dec)->si.total_samples : 0; }
uint32_t FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *dec) { return impl(dec)->si.channels; }
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *dec) { return (FLAC__ChannelAssignment)impl(dec)->last_ca; }
uint32_t FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *dec) { return impl(dec)->si.bits_per_sample; }
uint32_t FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *dec) { return impl(dec)->si.sample_rate; }
uint32_t FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *dec) { return impl(dec)->last_blocksize; }
FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *dec, FLAC__uint64 *position)
{
    const DecImpl *d = impl(dec);
    if (!position || !d->file) return 0;
    *position = d->consumed_total;
    return 1;
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *dec, FLAC__StreamDecoderReadCallback read_callback,
                                                               FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback,
                                                               FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback,
                                                               FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback,
                                                               FLAC__StreamDecoderErrorCallback error_callback, void *client_data)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
    if (!read_callback || !write_callback || !error_callback || (seek_callback && (!tell_callback || !length_callback || !eof_callback)))
        return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
    d->read_cb = read_callback; d->write_cb = write_callback; d->meta_cb = metadata_callback; d->error_cb = error_callback;
    d->seek_cb = seek_callback; d->tell_cb = tell_callback; d->length_cb = length_callback; d->eof_cb = eof_callback;
    d->client = client_data; d->file = nullptr; d->own_file = false;
    return init_common(d);
}
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *, FLAC__StreamDecoderReadCallback, FLAC__StreamDecoderSeekCallback,
                                                                   FLAC__StreamDecoderTellCallback, FLAC__StreamDecoderLengthCallback, FLAC__StreamDecoderEofCallback,
                                                                   FLAC__StreamDecoderWriteCallback, FLAC__StreamDecoderMetadataCallback,
                                                                   FLAC__StreamDecoderErrorCallback, void *)
{
    return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *dec, FILE *file, FLAC__StreamDecoderWriteCallback write_callback,
                                                             FLAC__StreamDecoderMetadataCallback metadata_callback,
                                                             FLAC__StreamDecoderErrorCallback error_callback, void *client_data)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
    if (!file || !write_callback || !error_callback) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
    d->read_cb = nullptr; d->write_cb = write_callback; d->meta_cb = metadata_callback; d->error_cb = error_callback;
    d->client = client_data; d->file = file; d->own_file = false;
    return init_common(d);
}
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *, FILE *, FLAC__StreamDecoderWriteCallback,
                                                                 FLAC__StreamDecoderMetadataCallback, FLAC__StreamDecoderErrorCallback, void *)
{
    return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *dec, const char *filename, FLAC__StreamDecoderWriteCallback write_callback,
                                                             FLAC__StreamDecoderMetadataCallback metadata_callback,
                                                             FLAC__StreamDecoderErrorCallback error_callback, void *client_data)
{
    DecImpl *d = impl(dec);
    if (d->state != FLAC__STREAM_DECODER_UNINITIALIZED) return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
    if (!write_callback || !error_callback) return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
    FILE *f = filename ? fopen(filename, "rb") : stdin;
    if (!f) return FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE;
    const FLAC__StreamDecoderInitStatus rc = FLAC__stream_decoder_init_FILE(dec, f, write_callback, metadata_callback, error_callback, client_data);
    if (rc == FLAC__STREAM_DECODER_INIT_STATUS_OK) d->own_file = (f != stdin);
    else if (f != stdin) fclose(f);
    return rc;
}
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *, const char *, FLAC__StreamDecoderWriteCallback,
                                                                 FLAC__StreamDecoderMetadataCallback, FLAC__StreamDecoderErrorCallback, void *)
{
    return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 1;
    if (d->file) { if (d->own_file) fclose(d->file); d->file = nullptr; }
    // stream_decoder.h:1339-1348: false when MD5 checking is on, a STREAMINFO with a non-zero signature was read and
    // the signature of the delivered samples differs
    FLAC__bool md5_ok = 1;
    if (d->do_md5 && d->have_si) {
        static const uint8_t zero[16] = {0};
        uint8_t got[16];
        d->md5.final(got);
        if (memcmp(d->si.md5sum, zero, 16) != 0 && memcmp(d->si.md5sum, got, 16) != 0) md5_ok = 0;
    }
    d->md5_checking = 0;
    d->block_cb = nullptr;
    if (fg_tune("FLACGPU_API_PROF"))
        fprintf(stderr, "[flacgpu api prof] decoder totals: read callback + buffer %.3f ms, index %.3f, upload/decode/download %.3f, delivery %.3f, buffer compaction %.3f\n",
                d->prof_ms[0], d->prof_ms[1], d->prof_ms[2], d->prof_ms[3], d->prof_ms[4]);
    for (double &v : d->prof_ms) v = 0;
    // the large buffers go to the next decoder (a finished decoder may live on for a while, e.g. until a garbage collector runs)
    d->pcm.give_back();
    d->buf.give_back();
    if (d->ctx) { DevPool::give(d->d_stream, d->ctx->device); DevPool::give(d->d_pcm, d->ctx->device); DevPool::give(d->d_pcm16, d->ctx->device); }
    reset_stream(d);
    memset(d->respond, 0, sizeof d->respond); d->respond[FLAC__METADATA_TYPE_STREAMINFO] = true; d->app_ids.clear();
    d->state = FLAC__STREAM_DECODER_UNINITIALIZED;
    return md5_ok;
}
FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    drop_front(d, d->buf.size()); d->npre = 0;
    d->ix = Indexer(); d->frames.clear(); d->status.clear(); d->next_frame = 0; d->frames_delivered_bound = 0;
    d->win.restart(d->consumed_total); d->walker = fgref::Walker(); d->walk_pending = false; d->ix_origin = 0; d->walk_stuck_at = UINT64_MAX;
    d->eof = false;          // (the client may have moved its source: ask again)
    d->do_md5 = false;       // stream_decoder.h:1357-1359: a flush turns MD5 checking off
    // libFLAC's flush also forgets how far the stream has been decoded and the last frame's header (stream_decoder.h:1353-1366:
    // "the decoder's state is reset for the next frame"): nothing is filled with silence across a flush, and a STREAMINFO sample
    // count does not end a stream the client has rewound
    d->samples_decoded = 0; d->last_set = false;
    d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
    return 1;
}
FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    if (d->file) { if (fseek(d->file, 0, SEEK_SET) != 0) return 0; }
    reset_stream(d);
    d->state = FLAC__STREAM_DECODER_SEARCH_FOR_METADATA;
    return 1;
}

FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    if (d->state == FLAC__STREAM_DECODER_END_OF_STREAM) return 1;
    if (d->state == FLAC__STREAM_DECODER_ABORTED) return 0;
    if (!d->have_meta) {
        if (!ensure_metadata(d)) return d->state == FLAC__STREAM_DECODER_END_OF_STREAM;
        return 1;                                    // one call consumes the metadata, like libFLAC
    }
    if (!fill_queue(d)) return d->state == FLAC__STREAM_DECODER_END_OF_STREAM;
    if (d->round_blocks) return deliver_blocks(d, 1) ? 1 : 0;
    return deliver_one(d) ? 1 : 0;
}

FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    if (d->have_meta) return 1;
    if (!ensure_metadata(d)) return d->state == FLAC__STREAM_DECODER_END_OF_STREAM;
    return 1;
}

FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    for (;;) {
        if (d->state == FLAC__STREAM_DECODER_END_OF_STREAM) return 1;
        if (d->state == FLAC__STREAM_DECODER_ABORTED) return 0;
        if (!fill_queue(d)) return d->state == FLAC__STREAM_DECODER_END_OF_STREAM;
        ProfSpan span(&d->prof_ms[3]);
        if (d->round_blocks) { if (!deliver_blocks(d, ~(size_t)0)) return 0; }
        else while (d->next_frame < d->frames.size())
            if (!deliver_one(d)) return 0;
    }
}

FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *dec)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    if (!d->have_meta) return FLAC__stream_decoder_process_single(dec);
    if (!fill_queue(d)) return d->state == FLAC__STREAM_DECODER_END_OF_STREAM;
    d->samples_decoded += d->frames[d->next_frame].n;
    d->next_frame++;
    return 1;
}

// Seeking (stream_decoder.h:1412-1450).  pyFLAC declares the entry point and never calls it; libFLAC seeks when it was given a
// file or all four of the seek / tell / length / eof callbacks.  What a client of the reference binary sees (probed in the build
// container, tests/test_gpu_api.py TestSeek): the call itself delivers the frame that holds the target, from the target sample on
// and under the target's sample number, returns true and leaves the decoder ready for the frame behind it; a target at or behind
// STREAMINFO's sample count returns false and changes nothing; MD5 checking is off from then on.  Here the source is put back to
// the first frame and the frames in front of the target are decoded and dropped -- the GPU decodes a ten-minute stream in a few
// milliseconds, a search for the right place in the bytes would save little and add the resync cases of libFLAC's search.
FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *dec, FLAC__uint64 sample)
{
    DecImpl *d = impl(dec);
    if (d->state == FLAC__STREAM_DECODER_UNINITIALIZED) return 0;
    const bool seekable = d->file != nullptr || (d->seek_cb && d->tell_cb && d->length_cb && d->eof_cb);
    if (!seekable) { d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }      // (libFLAC without seek callbacks: the same)
    if (d->state == FLAC__STREAM_DECODER_ABORTED) return 0;
    if (!d->have_meta && !ensure_metadata(d)) return 0;
    if (d->have_si && d->si.total_samples && sample >= d->si.total_samples) return 0;
    if (d->file) { if (fseeko(d->file, (off_t)d->audio_start, SEEK_SET) != 0) { d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; } }
    else if (d->seek_cb(&d->pub, d->audio_start, d->client) != FLAC__STREAM_DECODER_SEEK_STATUS_OK) { d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }
    d->buf.clear(); d->npre = 0; d->consumed_total = d->audio_start;
    d->ix = Indexer(); d->frames.clear(); d->status.clear(); d->next_frame = 0; d->frames_delivered_bound = 0;
    d->win.restart(d->audio_start); d->walker = fgref::Walker(); d->walk_pending = false; d->ix_origin = 0; d->walk_stuck_at = UINT64_MAX;
    d->eof = false; d->do_md5 = false; d->samples_decoded = 0; d->last_set = false; d->fixed_blocksize = 0;
    d->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
    d->seeking = true; d->seek_target = sample; d->seek_done = false;
    while (d->seeking) {
        if (!fill_queue(d)) break;                                  // the stream ended (or the client aborted) in front of the target
        while (d->next_frame < d->frames.size() && d->seeking) {
            if (d->round_blocks) { if (!deliver_blocks(d, 1)) { d->seeking = false; return 0; } }
            else if (!deliver_one(d)) { d->seeking = false; return 0; }
        }
    }
    d->seeking = false;
    if (!d->seek_done) { d->state = FLAC__STREAM_DECODER_SEEK_ERROR; return 0; }
    return 1;
}

}  // extern "C"
