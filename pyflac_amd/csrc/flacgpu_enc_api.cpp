// flacgpu_enc_api.cpp -- the libFLAC stream-encoder entry points pyFLAC binds
// (reference: pyflac/builder/encoder.py:266-322), backed by the HIP frame encoder.
//
// Host responsibilities: input buffering with libFLAC's
// blocksize+1 look-ahead, the fLaC/STREAMINFO/VORBIS_COMMENT header, frame-size statistics, MD5 of the
// raw PCM, callback delivery in stream order.  Every frame byte is produced on the GPU; there is no CPU
// fallback: without a HIP device init_stream/init_file fail with ENCODER_ERROR.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <condition_variable>
#include <functional>

#include "fg_host.h"

extern "C" {
const char *FLAC__VERSION_STRING = "1.4.3";
const char *FLAC__VENDOR_STRING = "reference libFLAC 1.4.3 20230623";   // byte-exact VORBIS_COMMENT parity

const char *const FLAC__StreamEncoderStateString[] = {
    "FLAC__STREAM_ENCODER_OK", "FLAC__STREAM_ENCODER_UNINITIALIZED", "FLAC__STREAM_ENCODER_OGG_ERROR",
    "FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR", "FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA",
    "FLAC__STREAM_ENCODER_CLIENT_ERROR", "FLAC__STREAM_ENCODER_IO_ERROR", "FLAC__STREAM_ENCODER_FRAMING_ERROR",
    "FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR"};
const char *const FLAC__StreamEncoderInitStatusString[] = {
    "FLAC__STREAM_ENCODER_INIT_STATUS_OK", "FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR",
    "FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS",
    "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE",
    "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE",
    "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION",
    "FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE",
    "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA", "FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED"};
}

// ------------------------------------------------------------------ MD5 / CRC on the host
static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
// One 64-byte block (RFC 1321, section 3.4), rounds written out so that the message-word index and the rotation are
// immediates.  The serial chain a -> b -> c -> d bounds it at roughly 5 cycles per step on the host cores.
#define FG_ROL(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define FG_F(b, c, d) ((d) ^ ((b) & ((c) ^ (d))))
#define FG_G(b, c, d) ((c) ^ ((d) & ((b) ^ (c))))
#define FG_H(b, c, d) ((b) ^ (c) ^ (d))
#define FG_I(b, c, d) ((c) ^ ((b) | ~(d)))
#define FG_STEP(f, a, b, c, d, g, i, sft) do { (a) += f((b), (c), (d)) + w[g] + MD5_K[i]; (a) = FG_ROL((a), sft) + (b); } while (0)
static void md5_block(FgMd5 *m, const uint8_t *p)
{
    uint32_t w[16];
    memcpy(w, p, 64);   // little-endian host
    uint32_t a = m->a, b = m->b, c = m->c, d = m->d;
#define FG_R4(f, i, g0, g1, g2, g3, s0, s1, s2, s3) \
    FG_STEP(f, a, b, c, d, g0, i, s0); FG_STEP(f, d, a, b, c, g1, i + 1, s1); FG_STEP(f, c, d, a, b, g2, i + 2, s2); FG_STEP(f, b, c, d, a, g3, i + 3, s3)
    FG_R4(FG_F, 0, 0, 1, 2, 3, 7, 12, 17, 22);    FG_R4(FG_F, 4, 4, 5, 6, 7, 7, 12, 17, 22);
    FG_R4(FG_F, 8, 8, 9, 10, 11, 7, 12, 17, 22);  FG_R4(FG_F, 12, 12, 13, 14, 15, 7, 12, 17, 22);
    FG_R4(FG_G, 16, 1, 6, 11, 0, 5, 9, 14, 20);   FG_R4(FG_G, 20, 5, 10, 15, 4, 5, 9, 14, 20);
    FG_R4(FG_G, 24, 9, 14, 3, 8, 5, 9, 14, 20);   FG_R4(FG_G, 28, 13, 2, 7, 12, 5, 9, 14, 20);
    FG_R4(FG_H, 32, 5, 8, 11, 14, 4, 11, 16, 23); FG_R4(FG_H, 36, 1, 4, 7, 10, 4, 11, 16, 23);
    FG_R4(FG_H, 40, 13, 0, 3, 6, 4, 11, 16, 23);  FG_R4(FG_H, 44, 9, 12, 15, 2, 4, 11, 16, 23);
    FG_R4(FG_I, 48, 0, 7, 14, 5, 6, 10, 15, 21);  FG_R4(FG_I, 52, 12, 3, 10, 1, 6, 10, 15, 21);
    FG_R4(FG_I, 56, 8, 15, 6, 13, 6, 10, 15, 21); FG_R4(FG_I, 60, 4, 11, 2, 9, 6, 10, 15, 21);
#undef FG_R4
    m->a += a; m->b += b; m->c += c; m->d += d;
}
void FgMd5::init() { a = 0x67452301; b = 0xefcdab89; c = 0x98badcfe; d = 0x10325476; len = 0; fill = 0; }
void FgMd5::update(const uint8_t *p, size_t n)
{
    len += n;
    if (fill) {
        size_t k = std::min<size_t>(64 - fill, n);
        memcpy(buf + fill, p, k);
        fill += (uint32_t)k; p += k; n -= k;
        if (fill == 64) { md5_block(this, buf); fill = 0; }
    }
    while (n >= 64) { md5_block(this, p); p += 64; n -= 64; }
    if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
}
void FgMd5::final(uint8_t out[16])
{
    const uint64_t bits = len * 8;
    uint8_t pad[72];
    memset(pad, 0, sizeof pad);
    pad[0] = 0x80;
    const size_t padlen = (fill < 56) ? (56 - fill) : (120 - fill);
    update(pad, padlen);
    uint8_t l[8];
    for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
    update(l, 8);
    const uint32_t v[4] = {a, b, c, d};
    memcpy(out, v, 16);
}
void FgMd5::update_pcm(const int32_t *x, uint64_t nvalues, uint32_t bps)
{
    // little-endian, (bps+7)/8 bytes per sample, interleaved
    const uint32_t bytes = (bps + 7) / 8;
    alignas(8) uint8_t tmp[4096 + 8];
    size_t f = 0;
    if (bytes == 2) {
        uint64_t i = 0;
        while (i < nvalues) {
            const uint64_t k = std::min<uint64_t>(nvalues - i, 2048);
            uint16_t *t16 = (uint16_t *)tmp;                 // little-endian host: the low half of every value
            for (uint64_t j = 0; j < k; j++) t16[j] = (uint16_t)x[i + j];
            update(tmp, (size_t)k * 2);
            i += k;
        }
    }
    else {
        for (uint64_t i = 0; i < nvalues; i++) {
            const uint32_t v = (uint32_t)x[i];
            for (uint32_t b = 0; b < bytes; b++) tmp[f++] = (uint8_t)(v >> (8 * b));
            if (f >= 4096) { update(tmp, f); f = 0; }
        }
    }
    update(tmp, f);
}

static uint8_t g_crc8[256];
static uint16_t g_crc16[256];
static std::once_flag g_crc_once;
static void crc_tables()
{
    for (int i = 0; i < 256; i++) {
        uint8_t c = (uint8_t)i;
        for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
        g_crc8[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++) d = (uint16_t)((d & 0x8000) ? ((d << 1) ^ 0x8005) : (d << 1));
        g_crc16[i] = d;
    }
}
uint8_t fg_crc8(const uint8_t *p, size_t n)
{
    std::call_once(g_crc_once, crc_tables);
    uint8_t c = 0;
    while (n--) c = g_crc8[c ^ *p++];
    return c;
}
uint16_t fg_crc16(const uint8_t *p, size_t n)
{
    std::call_once(g_crc_once, crc_tables);
    uint16_t c = 0;
    while (n--) c = (uint16_t)((c << 8) ^ g_crc16[(c >> 8) ^ *p++]);
    return c;
}

// ------------------------------------------------------------------ encoder object
namespace {

// The MD5 of a process call runs on a helper thread beside the call's GPU work.  Round 5: ONE thread per encoder, started at the
// first call and woken through a condition variable -- creating and joining a std::thread inside every call was a third of the host
// time of a one-block call ("what's weak" 6).
struct Md5Worker {
    std::thread t;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
    void submit(std::function<void()> j)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            if (!t.joinable()) t = std::thread([this] { run(); });
            job = std::move(j);
            busy = true;
        }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || (bool)job; });
            if (!job) return;
            std::function<void()> j = std::move(job);
            job = nullptr;
            lk.unlock();
            j();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    ~Md5Worker()
    {
        if (t.joinable()) {
            { std::lock_guard<std::mutex> lk(m); quit = true; }
            cv.notify_all();
            t.join();
        }
    }
};

struct EncImpl {
    FLAC__StreamEncoder pub;   // must be first: the handle pyFLAC holds points here
    FLAC__StreamEncoderState state;
    // settings as the setters leave them
    FLAC__bool verify, streamable_subset, do_md5, limit_min_bitrate;
    uint32_t channels, bps, sample_rate, blocksize;
    FLAC__bool do_mid_side, loose_mid_side;
    uint32_t apod_parts;
    bool apod_supported;
    uint32_t max_lpc_order, qlp_precision;
    FLAC__bool prec_search, escape_coding, exhaustive;
    uint32_t min_po, max_po, rice_dist;
    uint64_t total_estimate;
    // run state
    flacgpu_settings s;
    flacgpu_ctx *ctx;
    FLAC__StreamEncoderWriteCallback write_cb;
    FLAC__StreamEncoderSeekCallback seek_cb;
    FLAC__StreamEncoderTellCallback tell_cb;
    FLAC__StreamEncoderMetadataCallback meta_cb;
    FLAC__StreamEncoderProgressCallback progress_cb;
    void *client;
    FILE *file;
    bool own_file;
    uint64_t bytes_written;
    std::vector<int32_t> pending;   // interleaved samples not yet encoded
    uint32_t launch_blocks = 1;     // flacgpu_stream_encoder_set_launch_blocks: complete blocks to wait for before a launch
    uint32_t frame_number;
    uint32_t last_ca = 0;          // loose mid-side: channel assignment the previous process call ended on
    std::vector<FLAC__StreamMetadata *> metadata;   // FLAC__stream_encoder_set_metadata: the caller's blocks (pointers only)
    uint64_t samples_done;
    uint32_t min_frame, max_frame;
    FgMd5 md5;
    void *h_pin = nullptr;          // pinned host copy of the encoded frames (+ offsets) of one call
    size_t h_pin_cap = 0;
    void *h_in = nullptr;           // pinned staging of the samples of a small call (the upload then is one asynchronous copy)
    size_t h_in_cap = 0;
    Md5Worker md5w;
    std::vector<uint64_t> hoffs;
    DevBuf d_pcm, d_out, d_offs;    // per-encoder device staging
    DevBuf d_in16;                  // 16-bit input as uploaded, before the device widens it into d_pcm
    DevBuf d_verify;                // verify: decoded PCM + the first-mismatch word
    // verify: where the round trip differed (FLAC__stream_encoder_get_verify_decoder_error_stats)
    uint64_t v_abs_sample; uint32_t v_frame, v_channel, v_sample; int32_t v_expected, v_got;
};

void set_defaults(EncImpl *e)
{
    e->verify = 0; e->streamable_subset = 1; e->do_md5 = 1; e->limit_min_bitrate = 0;
    e->metadata.clear();
    e->v_abs_sample = 0; e->v_frame = 0; e->v_channel = 0; e->v_sample = 0; e->v_expected = 0; e->v_got = 0;
    e->channels = 2; e->bps = 16; e->sample_rate = 44100; e->blocksize = 0;
    e->total_estimate = 0;
    e->prec_search = 0; e->escape_coding = 0; e->exhaustive = 0; e->rice_dist = 0; e->qlp_precision = 0;
    e->apod_supported = true;
    // compression level 5 (stream_encoder.h:850)
    e->do_mid_side = 1; e->loose_mid_side = 0; e->apod_parts = 0; e->max_lpc_order = 8; e->min_po = 0; e->max_po = 5;
    e->write_cb = nullptr; e->seek_cb = nullptr; e->tell_cb = nullptr; e->meta_cb = nullptr; e->progress_cb = nullptr;
    e->client = nullptr; e->file = nullptr; e->own_file = false;
}

inline EncImpl *impl(FLAC__StreamEncoder *e) { return reinterpret_cast<EncImpl *>(e); }
inline const EncImpl *impl(const FLAC__StreamEncoder *e) { return reinterpret_cast<const EncImpl *>(e); }

bool emit(EncImpl *e, const uint8_t *p, size_t n, uint32_t samples, uint32_t frame)
{
    if (e->file) {
        if (fwrite(p, 1, n, e->file) != n) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return false; }
    }
    else if (e->write_cb(&e->pub, p, n, samples, frame, e->client) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
        e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
        return false;
    }
    e->bytes_written += n;
    return true;
}

size_t build_header(const EncImpl *e, uint8_t *out, uint32_t minf, uint32_t maxf, uint64_t total, const uint8_t *md5)
{
    // fLaC + STREAMINFO (format.h:536-557) + VORBIS_COMMENT with the vendor string only
    const flacgpu_settings &s = e->s;
    uint8_t *p = out;
    memcpy(p, "fLaC", 4); p += 4;
    *p++ = 0x00; *p++ = 0; *p++ = 0; *p++ = 34;
    *p++ = (uint8_t)(s.blocksize >> 8); *p++ = (uint8_t)s.blocksize;
    *p++ = (uint8_t)(s.blocksize >> 8); *p++ = (uint8_t)s.blocksize;
    *p++ = (uint8_t)(minf >> 16); *p++ = (uint8_t)(minf >> 8); *p++ = (uint8_t)minf;
    *p++ = (uint8_t)(maxf >> 16); *p++ = (uint8_t)(maxf >> 8); *p++ = (uint8_t)maxf;
    const uint64_t v = ((uint64_t)s.sample_rate << 44) | ((uint64_t)(s.channels - 1) << 41) |
                       ((uint64_t)(s.bits_per_sample - 1) << 36) | (total & 0xFFFFFFFFFull);
    for (int i = 7; i >= 0; i--) *p++ = (uint8_t)(v >> (8 * i));
    if (md5) memcpy(p, md5, 16); else memset(p, 0, 16);
    p += 16;
    const uint32_t vl = (uint32_t)strlen(FLAC__VENDOR_STRING);
    *p++ = 0x84; *p++ = 0; *p++ = 0; *p++ = (uint8_t)(8 + vl);
    *p++ = (uint8_t)vl; *p++ = (uint8_t)(vl >> 8); *p++ = (uint8_t)(vl >> 16); *p++ = (uint8_t)(vl >> 24);
    memcpy(p, FLAC__VENDOR_STRING, vl); p += vl;
    memset(p, 0, 4); p += 4;
    return (size_t)(p - out);
}

// ---- metadata blocks supplied through FLAC__stream_encoder_set_metadata (stream_encoder.h:1132-1214, format.h:560-880)
// Serialise one block (4-byte header + body) the way libFLAC's stream_encoder_framing.c does.  Returns false for a block
// that cannot be written (unknown layout, inconsistent length).
static void put_be(std::vector<uint8_t> &o, uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_le32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }

static bool serialise_metadata(const FLAC__StreamMetadata *m, bool is_last, std::vector<uint8_t> &o)
{
    std::vector<uint8_t> b;
    switch (m->type) {
    case FLAC__METADATA_TYPE_PADDING:
        b.assign(m->length, 0);
        break;
    case FLAC__METADATA_TYPE_APPLICATION:
        if (m->length < 4) return false;
        b.insert(b.end(), m->data.application.id, m->data.application.id + 4);
        if (m->length > 4) {
            if (!m->data.application.data) return false;
            b.insert(b.end(), m->data.application.data, m->data.application.data + (m->length - 4));
        }
        break;
    case FLAC__METADATA_TYPE_SEEKTABLE:
        for (uint32_t i = 0; i < m->data.seek_table.num_points; i++) {
            const FLAC__StreamMetadata_SeekPoint &p = m->data.seek_table.points[i];
            put_be(b, p.sample_number, 8); put_be(b, p.stream_offset, 8); put_be(b, p.frame_samples, 2);
        }
        break;
    case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
        // the vendor string is always libFLAC's own (stream_encoder.h:1180-1186)
        const uint32_t vl = (uint32_t)strlen(FLAC__VENDOR_STRING);
        put_le32(b, vl);
        b.insert(b.end(), (const uint8_t *)FLAC__VENDOR_STRING, (const uint8_t *)FLAC__VENDOR_STRING + vl);
        put_le32(b, m->data.vorbis_comment.num_comments);
        for (uint32_t i = 0; i < m->data.vorbis_comment.num_comments; i++) {
            const FLAC__StreamMetadata_VorbisComment_Entry &c = m->data.vorbis_comment.comments[i];
            put_le32(b, c.length);
            if (c.length) { if (!c.entry) return false; b.insert(b.end(), c.entry, c.entry + c.length); }
        }
        break;
    }
    case FLAC__METADATA_TYPE_CUESHEET: {
        const FLAC__StreamMetadata_CueSheet &cs = m->data.cue_sheet;
        b.insert(b.end(), (const uint8_t *)cs.media_catalog_number, (const uint8_t *)cs.media_catalog_number + 128);
        put_be(b, cs.lead_in, 8);
        b.push_back(cs.is_cd ? 0x80 : 0x00);                       // 1 bit + 7 of the 2071 reserved bits
        b.insert(b.end(), 258, 0);
        b.push_back((uint8_t)cs.num_tracks);
        for (uint32_t t = 0; t < cs.num_tracks; t++) {
            const FLAC__StreamMetadata_CueSheet_Track &tr = cs.tracks[t];
            put_be(b, tr.offset, 8);
            b.push_back(tr.number);
            b.insert(b.end(), (const uint8_t *)tr.isrc, (const uint8_t *)tr.isrc + 12);
            b.push_back((uint8_t)((tr.type ? 0x80 : 0) | (tr.pre_emphasis ? 0x40 : 0)));      // 2 bits + 6 of the 110 reserved bits
            b.insert(b.end(), 13, 0);
            b.push_back(tr.num_indices);
            for (uint32_t k = 0; k < tr.num_indices; k++) {
                put_be(b, tr.indices[k].offset, 8);
                b.push_back(tr.indices[k].number);
                b.insert(b.end(), 3, 0);
            }
        }
        break;
    }
    case FLAC__METADATA_TYPE_PICTURE: {
        const FLAC__StreamMetadata_Picture &pc = m->data.picture;
        const size_t ml = pc.mime_type ? strlen(pc.mime_type) : 0, dl = pc.description ? strlen((const char *)pc.description) : 0;
        put_be(b, (uint32_t)pc.type, 4);
        put_be(b, ml, 4); b.insert(b.end(), (const uint8_t *)pc.mime_type, (const uint8_t *)pc.mime_type + ml);
        put_be(b, dl, 4); b.insert(b.end(), (const uint8_t *)pc.description, (const uint8_t *)pc.description + dl);
        put_be(b, pc.width, 4); put_be(b, pc.height, 4); put_be(b, pc.depth, 4); put_be(b, pc.colors, 4);
        put_be(b, pc.data_length, 4);
        if (pc.data_length) { if (!pc.data) return false; b.insert(b.end(), pc.data, pc.data + pc.data_length); }
        break;
    }
    case FLAC__METADATA_TYPE_STREAMINFO:
        return false;
    default:
        if (m->length) { if (!m->data.unknown.data) return false; b.insert(b.end(), m->data.unknown.data, m->data.unknown.data + m->length); }
        break;
    }
    if (b.size() >= (1u << 24)) return false;
    o.push_back((uint8_t)((is_last ? 0x80 : 0) | ((uint32_t)m->type & 0x7F)));
    put_be(o, b.size(), 3);
    o.insert(o.end(), b.begin(), b.end());
    return true;
}

// What init checks about the supplied blocks (stream_encoder.c init_stream_internal_): no STREAMINFO, at most one SEEKTABLE
// and one VORBIS_COMMENT, a legal seek table (ascending sample numbers, placeholders last)
// FLAC__format_cuesheet_is_legal / FLAC__format_picture_is_legal as libFLAC's encoder applies them at init (a CUESHEET is
// checked against the CD-DA subset when it says is_cd); pinned by tests/golden/legality_vectors.json, recorded from the
// reference binary.
static uint32_t utf8_len(const uint8_t *u)
{
    if ((u[0] & 0x80) == 0) return 1;
    if ((u[0] & 0xE0) == 0xC0 && (u[1] & 0xC0) == 0x80) return (u[0] & 0xFE) == 0xC0 ? 0 : 2;               // (overlong)
    if ((u[0] & 0xF0) == 0xE0 && (u[1] & 0xC0) == 0x80 && (u[2] & 0xC0) == 0x80) {
        if (u[0] == 0xE0 && (u[1] & 0xE0) == 0x80) return 0;                                                   // overlong
        if (u[0] == 0xED && (u[1] & 0xE0) == 0xA0) return 0;                                                   // U+D800 .. U+DFFF
        if (u[0] == 0xEF && u[1] == 0xBF && (u[2] & 0xFE) == 0xBE) return 0;                                   // U+FFFE, U+FFFF
        return 3;
    }
    if ((u[0] & 0xF8) == 0xF0 && (u[1] & 0xC0) == 0x80 && (u[2] & 0xC0) == 0x80 && (u[3] & 0xC0) == 0x80)
        return (u[0] == 0xF0 && (u[1] & 0xF0) == 0x80) ? 0 : 4;
    if ((u[0] & 0xFC) == 0xF8 && (u[1] & 0xC0) == 0x80 && (u[2] & 0xC0) == 0x80 && (u[3] & 0xC0) == 0x80 && (u[4] & 0xC0) == 0x80)
        return (u[0] == 0xF8 && (u[1] & 0xF8) == 0x80) ? 0 : 5;
    if ((u[0] & 0xFE) == 0xFC && (u[1] & 0xC0) == 0x80 && (u[2] & 0xC0) == 0x80 && (u[3] & 0xC0) == 0x80 && (u[4] & 0xC0) == 0x80 &&
        (u[5] & 0xC0) == 0x80)
        return (u[0] == 0xFC && (u[1] & 0xFC) == 0x80) ? 0 : 6;
    return 0;
}

static bool picture_is_legal(const FLAC__StreamMetadata_Picture &p)
{
    if (!p.mime_type || !p.description) return false;
    for (const char *c = p.mime_type; *c; c++) if ((uint8_t)*c < 0x20 || (uint8_t)*c > 0x7E) return false;
    for (const uint8_t *b = p.description; *b;) {
        const uint32_t n = utf8_len(b);
        if (n == 0) return false;
        b += n;
    }
    return true;
}

static bool cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet &cs)
{
    const bool cd = cs.is_cd != 0;
    if (cd && (cs.lead_in < 2 * 44100 || cs.lead_in % 588 != 0)) return false;
    if (cs.num_tracks == 0) return false;                                   // (the lead-out is a track)
    if (cd && cs.tracks[cs.num_tracks - 1].number != 170) return false;
    for (uint32_t i = 0; i < cs.num_tracks; i++) {
        const FLAC__StreamMetadata_CueSheet_Track &t = cs.tracks[i];
        if (t.number == 0) return false;
        if (cd && !((t.number >= 1 && t.number <= 99) || t.number == 170)) return false;
        if (cd && t.offset % 588 != 0) return false;
        if (i < cs.num_tracks - 1) {
            if (t.num_indices == 0) return false;
            if (t.indices[0].number > 1) return false;
        }
        for (uint32_t j = 0; j < t.num_indices; j++) {
            if (cd && t.indices[j].offset % 588 != 0) return false;
            if (j > 0 && t.indices[j].number != t.indices[j - 1].number + 1) return false;
        }
    }
    return true;
}

static bool metadata_acceptable(const std::vector<FLAC__StreamMetadata *> &v)
{
    bool seek = false, vc = false;
    for (const FLAC__StreamMetadata *m : v) {
        if (!m) return false;
        if (m->type == FLAC__METADATA_TYPE_STREAMINFO) return false;
        if (m->type == FLAC__METADATA_TYPE_SEEKTABLE) {
            if (seek) return false;
            seek = true;
            uint64_t prev = 0;
            bool got = false;
            for (uint32_t i = 0; i < m->data.seek_table.num_points; i++) {
                const uint64_t sn = m->data.seek_table.points[i].sample_number;
                if (got && sn != 0xFFFFFFFFFFFFFFFFull && sn <= prev) return false;
                if (got && prev == 0xFFFFFFFFFFFFFFFFull && sn != prev) return false;
                prev = sn; got = true;
            }
        }
        if (m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) { if (vc) return false; vc = true; }
        if (m->type == FLAC__METADATA_TYPE_CUESHEET && !cuesheet_is_legal(m->data.cue_sheet)) return false;
        if (m->type == FLAC__METADATA_TYPE_PICTURE && !picture_is_legal(m->data.picture)) return false;
    }
    return true;
}

FLAC__StreamEncoderInitStatus init_common(EncImpl *e)
{
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
    flacgpu_settings &s = e->s;
    memset(&s, 0, sizeof s);
    s.channels = e->channels; s.bits_per_sample = e->bps; s.sample_rate = e->sample_rate; s.blocksize = e->blocksize;
    s.do_mid_side = e->do_mid_side; s.loose_mid_side = e->loose_mid_side; s.max_lpc_order = e->max_lpc_order;
    s.qlp_coeff_precision = e->qlp_precision; s.min_partition_order = e->min_po; s.max_partition_order = e->max_po;
    s.apod_parts = e->apod_parts; s.streamable_subset = e->streamable_subset; s.limit_min_bitrate = e->limit_min_bitrate ? 1 : 0;
    const int rc = fg_resolve_settings(&s);
    if (rc != 0) return (FLAC__StreamEncoderInitStatus)rc;
    // settings libFLAC accepts but this GPU core does not implement are refused loudly (out of scope)
    if (e->prec_search || e->exhaustive || e->escape_coding || !e->apod_supported || s.max_partition_order > 8 || s.apod_parts > 3) {
        fg_set_error("setting not supported by the GPU encoder");
        return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
    }
    if (!metadata_acceptable(e->metadata)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
    e->ctx = fg_default_ctx();
    if (!e->ctx) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
    FgEncParams P;
    fg_fill_params(s, s.blocksize, false, false, &P);
    P.sig_stride = 0;   // worst case: samples read in place
    if (fg_enc_lds_bytes(&P) > 160 * 1024) {
        fg_set_error("settings need more LDS than the device has");
        return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
    }
    e->blocksize = s.blocksize; e->qlp_precision = s.qlp_coeff_precision; e->do_mid_side = s.do_mid_side;
    e->loose_mid_side = s.loose_mid_side; e->min_po = s.min_partition_order; e->max_po = s.max_partition_order;
    e->pending.clear();
    e->frame_number = 0; e->last_ca = 0; e->samples_done = 0; e->min_frame = 0; e->max_frame = 0; e->bytes_written = 0;
    e->md5.init();
    e->state = FLAC__STREAM_ENCODER_OK;
    // metadata writes: "fLaC", STREAMINFO, then either the default VORBIS_COMMENT followed by the caller's blocks, or -- when
    // the caller supplied a VORBIS_COMMENT -- the caller's blocks alone, in the order given (native FLAC keeps the order;
    // only Ogg FLAC moves the VORBIS_COMMENT to the front, stream_encoder.h:1484-1486 / the golden vectors).
    uint8_t hdr[128];
    const size_t hl = build_header(e, hdr, 0, 0, 0, nullptr);
    bool user_vc = false;
    for (const FLAC__StreamMetadata *m : e->metadata) user_vc |= m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT;
    bool ok = emit(e, hdr, 4, 0, 0) && emit(e, hdr + 4, 38, 0, 0);
    if (ok && !user_vc) {
        if (!e->metadata.empty()) hdr[42] &= 0x7F;          // the default VORBIS_COMMENT is no longer the last block
        ok = emit(e, hdr + 42, hl - 42, 0, 0);
    }
    std::vector<uint8_t> blk;
    for (size_t i = 0; ok && i < e->metadata.size(); i++) {
        blk.clear();
        ok = serialise_metadata(e->metadata[i], i + 1 == e->metadata.size(), blk) && emit(e, blk.data(), blk.size(), 0, 0);
    }
    if (!ok) {
        e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
        return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
    }
    return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
}

// Encode the complete blocks (with `flush_all`, also the remaining partial block) of what is buffered in e->pending followed
// by the caller's new samples (`in32` or `in16`, interleaved, `in_samples` inter-channel samples; both null = nothing new).
// The new samples go to the device straight from the caller's buffer -- 16-bit input as 16-bit, widened by a kernel -- and
// only what is left over (less than a block + 1 sample) is copied into e->pending.
bool encode_pending(EncImpl *e, bool flush_all, const int32_t *in32 = nullptr, const int16_t *in16 = nullptr, uint64_t in_samples = 0)
{
    const uint32_t C = e->s.channels, bs = e->s.blocksize;
    const uint64_t pend = e->pending.size() / C;
    const uint64_t have = pend + in_samples;
    auto keep_input = [&](uint64_t from) {          // append the caller's samples [from, in_samples) to e->pending
        const size_t base = e->pending.size(), nv = (size_t)(in_samples - from) * C;
        e->pending.resize(base + nv);
        int32_t *dst = e->pending.data() + base;
        if (in32) memcpy(dst, in32 + (size_t)from * C, nv * 4);
        else if (in16) { const int16_t *src = in16 + (size_t)from * C; for (size_t k = 0; k < nv; k++) dst[k] = src[k]; }
    };
    uint64_t take;
    if (flush_all) take = have;
    else take = have >= 1 ? ((have - 1) / bs) * bs : 0;   // libFLAC emits a frame once blocksize+1 samples are buffered
    // (extension, flacgpu_stream_encoder_set_launch_blocks: fewer, larger launches for callers that feed small pieces and can
    // take their frames in bursts -- every launch is six kernels and a wait, whatever it carries)
    if (take == 0 || (!flush_all && take < (uint64_t)e->launch_blocks * bs)) { keep_input(0); return true; }
    const uint64_t from_pend = std::min<uint64_t>(take, pend), from_in = take - from_pend;
    flacgpu_ctx *c = e->ctx;
    (void)hipSetDevice(c->device);
    const size_t pcm_bytes = (size_t)take * C * 4;
    flacgpu_stream_desc sd;
    sd.pcm_offset = 0; sd.nsamples = take; sd.first_frame = e->frame_number; sd.prev_channel_assignment = e->last_ca;
    uint32_t nblocks = 0;
    const uint64_t bound = flacgpu_encode_bound(&e->s, &sd, 1, &nblocks);
    if (!e->d_pcm.ensure(pcm_bytes) || !e->d_out.ensure(bound) || !e->d_offs.ensure(((size_t)nblocks + 1) * 8) ||
        (in16 && from_in && !e->d_in16.ensure((size_t)from_in * C * 2))) {
        e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return false;
    }
    // MD5 of the consumed PCM on a helper thread beside everything else this call does (upload, kernels, download, the
    // client's write callbacks); joined before the samples are dropped from `pending`, on every way out.  16-bit input at 16
    // bits per sample is hashed as it lies in memory (the MD5 is over little-endian samples, format.h:560).
    struct Joiner { Md5Worker *w = nullptr; ~Joiner() { if (w) w->wait(); } } md5j;
    if (e->do_md5) md5j.w = &e->md5w;
    if (e->do_md5) e->md5w.submit([=] {
        if (from_pend) e->md5.update_pcm(e->pending.data(), from_pend * C, e->s.bits_per_sample);
        if (from_in) {
            if (in32) e->md5.update_pcm(in32, from_in * C, e->s.bits_per_sample);
            else if (e->s.bits_per_sample > 8 && e->s.bits_per_sample <= 16) e->md5.update((const uint8_t *)in16, (size_t)from_in * C * 2);
            else {
                std::vector<int32_t> w(4096);
                for (uint64_t o = 0; o < from_in * C; o += 4096) {
                    const size_t k = (size_t)std::min<uint64_t>(4096, from_in * C - o);
                    for (size_t q = 0; q < k; q++) w[q] = in16[o + q];
                    e->md5.update_pcm(w.data(), k, e->s.bits_per_sample);
                }
            }
        }
    });
    // Small calls (the streaming use of the class: a block or a few per call, pyflac/encoder.py:86-119): the samples are gathered in
    // pinned memory and go up with ONE asynchronous copy queued in front of the encode kernels -- no wait of its own --, and the
    // frames and their index are written by the kernels straight into pinned host memory, which the call's end-of-call signal makes
    // visible: no download, no second wait.  (Round 4: three stream synchronisations and up to five copies a call.)
    const size_t in_bytes = (size_t)from_pend * C * 4 + (size_t)from_in * C * (in16 ? 2 : 4);
    const bool small = !e->verify && nblocks <= 16 && in_bytes <= ((size_t)1 << 20);
    bool direct_out = false;
    if (small) {
        if (in_bytes + 64 > e->h_in_cap) {
            if (e->h_in) (void)hipHostFree(e->h_in);
            e->h_in = nullptr; e->h_in_cap = 0;
            const size_t want = std::max(in_bytes + 64, (size_t)256 << 10);
            if (hipHostMalloc(&e->h_in, want, hipHostMallocDefault) == hipSuccess) e->h_in_cap = want;
        }
        const size_t need = (((size_t)bound + 63) & ~(size_t)63) + ((size_t)nblocks + 1) * 8;
        if (need > e->h_pin_cap) {
            if (e->h_pin) (void)hipHostFree(e->h_pin);
            e->h_pin = nullptr; e->h_pin_cap = 0;
            const size_t want = std::max(need, (size_t)1 << 20) * 3 / 2;
            if (hipHostMalloc(&e->h_pin, want, hipHostMallocDefault) == hipSuccess) e->h_pin_cap = want;
        }
        direct_out = e->h_in != nullptr && e->h_pin != nullptr;
    }
    if (direct_out) {
        uint8_t *hp = (uint8_t *)e->h_in;
        const size_t pb = (size_t)from_pend * C * 4;
        if (from_pend) memcpy(hp, e->pending.data(), pb);
        if (from_in) memcpy(hp + pb, in32 ? (const void *)in32 : (const void *)in16, in_bytes - pb);
        int32_t *d_pcm = (int32_t *)e->d_pcm.p;
        bool up = true;
        if (from_pend) up = hipMemcpyAsync(d_pcm, hp, pb, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        if (up && from_in && in32) up = hipMemcpyAsync(d_pcm + (size_t)from_pend * C, hp + pb, in_bytes - pb, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        if (up && from_in && in16)
            up = hipMemcpyAsync(e->d_in16.p, hp + pb, in_bytes - pb, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                 fg_launch_widen16((const int16_t *)e->d_in16.p, d_pcm + (size_t)from_pend * C, (uint64_t)from_in * C, c->stream) == 0;
        if (!up) { e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return false; }
        fg_set_input_on_stream(true);          // (the encode call must order every stream it uses behind what is queued on the main one)
    }
    else {
        bool up = true;
        int32_t *d_pcm = (int32_t *)e->d_pcm.p;
        if (from_pend) up = hipMemcpyAsync(d_pcm, e->pending.data(), (size_t)from_pend * C * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        if (up && from_in && in32) up = hipMemcpyAsync(d_pcm + (size_t)from_pend * C, in32, (size_t)from_in * C * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        if (up && from_in && in16)
            up = hipMemcpyAsync(e->d_in16.p, in16, (size_t)from_in * C * 2, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                 fg_launch_widen16((const int16_t *)e->d_in16.p, d_pcm + (size_t)from_pend * C, (uint64_t)from_in * C, c->stream) == 0;
        if (!up || hipStreamSynchronize(c->stream) != hipSuccess) { e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; return false; }
    }
    flacgpu_encode_stats st;
    uint8_t *const pin_frames = (uint8_t *)e->h_pin;
    uint64_t *const pin_offs = direct_out ? (uint64_t *)(pin_frames + (((size_t)bound + 63) & ~(size_t)63)) : nullptr;
    const int rc = direct_out ? flacgpu_encode_streams(c, &e->s, e->d_pcm.p, 0, &sd, 1, pin_frames, bound, pin_offs, &st)
                              : flacgpu_encode_streams(c, &e->s, e->d_pcm.p, 0, &sd, 1, e->d_out.p, e->d_out.cap, e->d_offs.p, &st);
    fg_set_input_on_stream(false);
    bool ok = rc == 0;
    if (ok) e->last_ca = st.last_channel_assignment;
    if (ok && st.error_flags) {
        ok = false;
        e->state = (st.error_flags & FG_ERR_RANGE) ? FLAC__STREAM_ENCODER_CLIENT_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
    }
    else if (!ok) e->state = FLAC__STREAM_ENCODER_FRAMING_ERROR;
    const uint8_t *hout = nullptr;
    if (ok && direct_out) { hout = pin_frames; e->hoffs.assign(pin_offs, pin_offs + st.nblocks + 1); }
    else if (ok) {
        // frames and their offsets land in pinned memory (per encoder; grown on demand) and are handed out from there
        const size_t need = (((size_t)st.total_bytes + 63) & ~(size_t)63) + ((size_t)st.nblocks + 1) * 8;
        if (need > e->h_pin_cap) {
            if (e->h_pin) (void)hipHostFree(e->h_pin);
            e->h_pin = nullptr; e->h_pin_cap = 0;
            const size_t want = std::max(need, (size_t)1 << 20) * 3 / 2;
            if (hipHostMalloc(&e->h_pin, want, hipHostMallocDefault) == hipSuccess) e->h_pin_cap = want;
        }
        if (!e->h_pin) { ok = false; e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; }
        else {
            uint8_t *hp = (uint8_t *)e->h_pin;
            uint64_t *ho = (uint64_t *)(hp + (((size_t)st.total_bytes + 63) & ~(size_t)63));
            if (hipMemcpyAsync(hp, e->d_out.p, st.total_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(ho, e->d_offs.p, ((size_t)st.nblocks + 1) * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) {
                ok = false; e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR;
            }
            else { hout = hp; e->hoffs.assign(ho, ho + st.nblocks + 1); }
        }
    }
    // verify (stream_encoder.h: FLAC__stream_encoder_set_verify): decode the fresh frames on the GPU and compare with the
    // PCM that went in, before anything is handed to the write callback
    if (ok && e->verify && st.nblocks) {
        flacgpu_decode_stats dst;
        std::vector<uint32_t> fstat((size_t)st.nblocks * 2);
        const uint64_t nvals = take * C;
        if (!e->d_verify.ensure((size_t)nvals * 4 + 64)) { ok = false; e->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR; }
        else if (flacgpu_decode_frames(c, e->d_out.p, st.total_bytes, (const uint64_t *)e->hoffs.data(), st.nblocks, C, e->s.bits_per_sample,
                                       e->d_verify.p, take, fstat.data(), &dst) != 0 || dst.error_frames != 0 || dst.total_samples != take) {
            ok = false; e->state = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR;
        }
        else {
            unsigned long long *d_first = (unsigned long long *)((char *)e->d_verify.p + (((size_t)nvals * 4 + 15) & ~(size_t)15));
            unsigned long long first = 0;
            // FLACGPU_VERIFY_SELFTEST: disturb the reference copy so that the mismatch path can be exercised by a test
            if (fg_sel("FLACGPU_VERIFY_SELFTEST")) {
                const int32_t first0 = from_pend ? e->pending[0] : (in32 ? in32[0] : (int32_t)in16[0]);
                const int32_t poison = first0 ^ 0x55;
                (void)hipMemcpy(e->d_pcm.p, &poison, 4, hipMemcpyHostToDevice);
            }
            if (fg_launch_compare((const int32_t *)e->d_verify.p, (const int32_t *)e->d_pcm.p, nvals, d_first, c->stream) != 0 ||
                hipMemcpyAsync(&first, d_first, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) { ok = false; e->state = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR; }
            else if (first != ~0ull) {
                int32_t got = 0, want = 0;
                (void)hipMemcpy(&got, (const int32_t *)e->d_verify.p + first, 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&want, (const int32_t *)e->d_pcm.p + first, 4, hipMemcpyDeviceToHost);
                const uint64_t fr = first / C;
                e->v_abs_sample = e->samples_done + fr; e->v_frame = e->frame_number + (uint32_t)(fr / bs);
                e->v_channel = (uint32_t)(first % C); e->v_sample = (uint32_t)(fr % bs);
                e->v_expected = want; e->v_got = got;
                ok = false; e->state = FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA;
            }
        }
    }
    if (!ok) return false;
    uint64_t pos = 0;
    for (uint32_t b = 0; b < st.nblocks; b++) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(bs, take - pos);
        const uint32_t fb = (uint32_t)(e->hoffs[b + 1] - e->hoffs[b]);
        if (!emit(e, hout + e->hoffs[b], fb, n, e->frame_number)) return false;
        if (e->min_frame == 0 || fb < e->min_frame) e->min_frame = fb;
        if (fb > e->max_frame) e->max_frame = fb;
        e->frame_number++;
        e->samples_done += n;
        pos += n;
        if (e->progress_cb) {
            const uint32_t est = e->total_estimate ? (uint32_t)((e->total_estimate + bs - 1) / bs) : 0;
            e->progress_cb(&e->pub, e->bytes_written, e->samples_done, e->frame_number, est, e->client);
        }
    }
    if (md5j.w) { md5j.w->wait(); md5j.w = nullptr; }
    e->pending.erase(e->pending.begin(), e->pending.begin() + (size_t)from_pend * C);
    keep_input(from_in);
    return true;
}

}  // namespace

extern "C" {

FLAC__StreamEncoder *FLAC__stream_encoder_new(void)
{
    EncImpl *e = new EncImpl();
    e->pub.protected_ = nullptr; e->pub.private_ = nullptr;
    e->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
    e->ctx = nullptr;
    set_defaults(e);
    return &e->pub;
}

void FLAC__stream_encoder_delete(FLAC__StreamEncoder *enc)
{
    if (!enc) return;
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) {
        // tolerate un-finished instances (GC finaliser, any thread): drop buffered data silently
        if (e->file && e->own_file) fclose(e->file);
    }
    if (e->ctx) (void)hipSetDevice(e->ctx->device);
    e->d_pcm.release(); e->d_out.release(); e->d_offs.release(); e->d_verify.release(); e->d_in16.release();
    if (e->h_pin) (void)hipHostFree(e->h_pin);
    if (e->h_in) (void)hipHostFree(e->h_in);
    delete e;
}

#define SETTER(name, field, type)                                                            \
    FLAC__bool FLAC__stream_encoder_set_##name(FLAC__StreamEncoder *enc, type value)          \
    {                                                                                         \
        EncImpl *e = impl(enc);                                                               \
        if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;                         \
        e->field = value;                                                                     \
        return 1;                                                                             \
    }
// The blocks to write behind STREAMINFO (stream_encoder.h:1132-1214).  Only the pointers are kept: the blocks must stay alive
// until finish.  A SEEKTABLE is written as given (what libFLAC does for a client without a seek callback); the seek points
// of a template are not filled in.
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *enc, FLAC__StreamMetadata **metadata, uint32_t num_blocks)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
    if (!metadata) num_blocks = 0;
    e->metadata.assign(metadata, metadata + num_blocks);
    return 1;
}

SETTER(verify, verify, FLAC__bool)
SETTER(channels, channels, uint32_t)
SETTER(bits_per_sample, bps, uint32_t)
SETTER(sample_rate, sample_rate, uint32_t)
SETTER(blocksize, blocksize, uint32_t)
SETTER(do_mid_side_stereo, do_mid_side, FLAC__bool)
SETTER(loose_mid_side_stereo, loose_mid_side, FLAC__bool)
SETTER(max_lpc_order, max_lpc_order, uint32_t)
SETTER(qlp_coeff_precision, qlp_precision, uint32_t)
SETTER(do_qlp_coeff_prec_search, prec_search, FLAC__bool)
SETTER(do_exhaustive_model_search, exhaustive, FLAC__bool)
SETTER(min_residual_partition_order, min_po, uint32_t)
SETTER(max_residual_partition_order, max_po, uint32_t)
SETTER(rice_parameter_search_dist, rice_dist, uint32_t)
SETTER(streamable_subset, streamable_subset, FLAC__bool)
SETTER(limit_min_bitrate, limit_min_bitrate, FLAC__bool)
SETTER(do_md5, do_md5, FLAC__bool)

FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *enc, FLAC__uint64 value)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
    e->total_estimate = value < ((1ull << 36) - 1) ? value : ((1ull << 36) - 1);
    return 1;
}

FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *enc, uint32_t value)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
    flacgpu_settings s;
    flacgpu_settings_from_level(&s, value, 2, 16, 44100, 4096, 0);   // only the preset columns are used
    static const uint32_t parts[9] = {0, 0, 0, 0, 0, 0, 2, 2, 3};
    static const uint32_t ms[9] = {0, 1, 1, 0, 1, 1, 1, 1, 1}, loose[9] = {0, 1, 0, 0, 1, 0, 0, 0, 0};
    const uint32_t lv = value > 8 ? 8 : value;
    e->do_mid_side = ms[lv]; e->loose_mid_side = loose[lv]; e->apod_parts = parts[lv]; e->apod_supported = true;
    e->max_lpc_order = s.max_lpc_order; e->qlp_precision = 0; e->prec_search = 0; e->escape_coding = 0; e->exhaustive = 0;
    e->min_po = 0;
    static const uint32_t maxpo[9] = {3, 3, 3, 4, 4, 5, 6, 6, 6};
    e->max_po = maxpo[lv];
    e->rice_dist = 0;
    return 1;
}

FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *enc, const char *spec)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
    // the level presets only use tukey(5e-1) and subdivide_tukey(N); anything else is refused at init
    if (!spec) return 0;
    if (!strcmp(spec, "tukey(5e-1)") || !strcmp(spec, "tukey(0.5)")) { e->apod_parts = 0; e->apod_supported = true; }
    else if (!strncmp(spec, "subdivide_tukey(", 16)) {
        const int n = atoi(spec + 16);
        e->apod_parts = (n >= 2) ? (uint32_t)n : 0;
        e->apod_supported = n >= 1 && strchr(spec, '/') == nullptr && strchr(spec, ';') == nullptr;
    }
    else e->apod_supported = false;
    return 1;
}

FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *enc) { return impl(enc)->state; }
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *enc)
{
    return FLAC__StreamEncoderStateString[impl(enc)->state];
}
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *enc, FLAC__uint64 *absolute_sample, uint32_t *frame_number,
                                                         uint32_t *channel, uint32_t *sample, FLAC__int32 *expected, FLAC__int32 *got)
{
    const EncImpl *e = impl(enc);
    if (absolute_sample) *absolute_sample = e->v_abs_sample;
    if (frame_number) *frame_number = e->v_frame;
    if (channel) *channel = e->v_channel;
    if (sample) *sample = e->v_sample;
    if (expected) *expected = e->v_expected;
    if (got) *got = e->v_got;
}
#define GETTER(name, field, type) \
    type FLAC__stream_encoder_get_##name(const FLAC__StreamEncoder *enc) { return (type)impl(enc)->field; }
GETTER(verify, verify, FLAC__bool)
GETTER(streamable_subset, streamable_subset, FLAC__bool)
GETTER(channels, channels, uint32_t)
GETTER(bits_per_sample, bps, uint32_t)
GETTER(sample_rate, sample_rate, uint32_t)
GETTER(blocksize, blocksize, uint32_t)
GETTER(do_mid_side_stereo, do_mid_side, FLAC__bool)
GETTER(loose_mid_side_stereo, loose_mid_side, FLAC__bool)
GETTER(max_lpc_order, max_lpc_order, uint32_t)
GETTER(qlp_coeff_precision, qlp_precision, uint32_t)
GETTER(do_qlp_coeff_prec_search, prec_search, FLAC__bool)
GETTER(do_escape_coding, escape_coding, FLAC__bool)
GETTER(do_exhaustive_model_search, exhaustive, FLAC__bool)
GETTER(min_residual_partition_order, min_po, uint32_t)
GETTER(max_residual_partition_order, max_po, uint32_t)
GETTER(rice_parameter_search_dist, rice_dist, uint32_t)
GETTER(total_samples_estimate, total_estimate, FLAC__uint64)
GETTER(limit_min_bitrate, limit_min_bitrate, FLAC__bool)

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *enc, FLAC__StreamEncoderWriteCallback write_callback,
                                                               FLAC__StreamEncoderSeekCallback seek_callback,
                                                               FLAC__StreamEncoderTellCallback tell_callback,
                                                               FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
    if (!write_callback || (seek_callback && !tell_callback)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS;
    e->write_cb = write_callback; e->seek_cb = seek_callback; e->tell_cb = tell_callback; e->meta_cb = metadata_callback;
    e->progress_cb = nullptr; e->client = client_data; e->file = nullptr; e->own_file = false;
    return init_common(e);
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *, FLAC__StreamEncoderReadCallback, FLAC__StreamEncoderWriteCallback,
                                                                   FLAC__StreamEncoderSeekCallback, FLAC__StreamEncoderTellCallback,
                                                                   FLAC__StreamEncoderMetadataCallback, void *)
{
    return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;   // as the reference build: FLAC_API_SUPPORTS_OGG_FLAC == 0
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *enc, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback,
                                                             void *client_data)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
    if (!file) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
    e->write_cb = nullptr; e->seek_cb = nullptr; e->tell_cb = nullptr; e->meta_cb = nullptr;
    e->progress_cb = progress_callback; e->client = client_data; e->file = file; e->own_file = false;
    FLAC__StreamEncoderInitStatus rc = init_common(e);
    if (rc != FLAC__STREAM_ENCODER_INIT_STATUS_OK) e->file = nullptr;
    return rc;
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *, FILE *, FLAC__StreamEncoderProgressCallback, void *)
{
    return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *enc, const char *filename,
                                                             FLAC__StreamEncoderProgressCallback progress_callback, void *client_data)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
    FILE *f = filename ? fopen(filename, "w+b") : stdout;
    if (!f) { e->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
    FLAC__StreamEncoderInitStatus rc = FLAC__stream_encoder_init_FILE(enc, f, progress_callback, client_data);
    if (rc == FLAC__STREAM_ENCODER_INIT_STATUS_OK) e->own_file = (f != stdout);
    else if (f != stdout) fclose(f);
    return rc;
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *, const char *, FLAC__StreamEncoderProgressCallback, void *)
{
    return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;
}

FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *enc, const FLAC__int32 buffer[], uint32_t samples)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_OK) return 0;
    return encode_pending(e, false, buffer, nullptr, samples) ? 1 : 0;
}

// Extension (include/flacgpu.h): the same with 16-bit interleaved input.  pyFLAC widens int16 arrays to int32 in numpy before
// the call (pyflac/encoder.py:112); taking them as they are saves that pass and halves the bytes that cross PCIe (the device
// widens them).
FLAC__bool flacgpu_stream_encoder_set_launch_blocks(FLAC__StreamEncoder *enc, uint32_t blocks)
{
    EncImpl *e = impl(enc);
    if (!e) return 0;
    e->launch_blocks = blocks < 1 ? 1 : blocks;
    return 1;
}

FLAC__bool flacgpu_stream_encoder_process_interleaved_i16(FLAC__StreamEncoder *enc, const int16_t *buffer, uint32_t samples)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_OK) return 0;
    return encode_pending(e, false, nullptr, buffer, samples) ? 1 : 0;
}

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *enc, const FLAC__int32 *const buffer[], uint32_t samples)
{
    EncImpl *e = impl(enc);
    if (e->state != FLAC__STREAM_ENCODER_OK) return 0;
    const uint32_t C = e->s.channels;
    const size_t base = e->pending.size();
    e->pending.resize(base + (size_t)samples * C);
    for (uint32_t i = 0; i < samples; i++)
        for (uint32_t c = 0; c < C; c++) e->pending[base + (size_t)i * C + c] = buffer[c][i];
    return encode_pending(e, false) ? 1 : 0;
}

FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *enc)
{
    EncImpl *e = impl(enc);
    if (e->state == FLAC__STREAM_ENCODER_UNINITIALIZED) return 1;
    bool error = false;
    if (e->state == FLAC__STREAM_ENCODER_OK) {
        if (!e->pending.empty() && !encode_pending(e, true)) error = true;
    }
    if (e->state == FLAC__STREAM_ENCODER_OK || !error) {
        uint8_t md5[16];
        memset(md5, 0, sizeof md5);
        if (e->do_md5) e->md5.final(md5);
        uint8_t hdr[128];
        build_header(e, hdr, e->min_frame, e->max_frame, e->samples_done, md5);
        if (e->file) {
            // rewrite STREAMINFO in place (file and stream outputs differ only in these bytes)
            if (fseek(e->file, 8, SEEK_SET) == 0) {
                if (fwrite(hdr + 8, 1, 34, e->file) != 34) error = true;
                fseek(e->file, 0, SEEK_END);
            }
        }
        else if (e->seek_cb && e->state == FLAC__STREAM_ENCODER_OK) {
            // libFLAC's update_metadata_: MD5 (offset 26), total samples (offset 21, 5 bytes, keeping the
            // bits-per-sample nibble), min/max frame size (offset 12), each a seek + write
            struct { uint64_t off; size_t len; } pieces[3] = {{26, 16}, {21, 5}, {12, 6}};
            for (int i = 0; i < 3 && !error; i++) {
                if (e->seek_cb(&e->pub, pieces[i].off, e->client) != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) { error = true; break; }
                if (e->write_cb(&e->pub, hdr + pieces[i].off, pieces[i].len, 0, 0, e->client) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) error = true;
            }
            if (error) e->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
        }
        if (e->meta_cb && e->state == FLAC__STREAM_ENCODER_OK) {
            FLAC__StreamMetadata m;
            memset(&m, 0, sizeof m);
            m.type = FLAC__METADATA_TYPE_STREAMINFO; m.is_last = 0; m.length = 34;
            FLAC__StreamMetadata_StreamInfo &si = m.data.stream_info;
            si.min_blocksize = si.max_blocksize = e->s.blocksize;
            si.min_framesize = e->min_frame; si.max_framesize = e->max_frame;
            si.sample_rate = e->s.sample_rate; si.channels = e->s.channels; si.bits_per_sample = e->s.bits_per_sample;
            si.total_samples = e->samples_done;
            memcpy(si.md5sum, md5, 16);
            e->meta_cb(&e->pub, &m, e->client);
        }
    }
    if (e->file) {
        if (e->own_file) fclose(e->file); else fflush(e->file);
        e->file = nullptr;
    }
    const bool failed = error || (e->state != FLAC__STREAM_ENCODER_OK);
    e->pending.clear(); e->pending.shrink_to_fit();
    set_defaults(e);                         // finish resets every setting (stream_encoder.h:225-227)
    e->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
    return failed ? 0 : 1;
}

}  // extern "C"
