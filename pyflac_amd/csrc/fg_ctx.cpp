// fg_ctx.cpp -- device context, settings resolution, window tables, MD5/CRC and the batch encode entry
// points of libflacgpu (Part 2 of include/flacgpu.h).
#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fg_host.h"
#include "fg_window_tables.inc"

static thread_local std::string g_err;
static thread_local bool g_input_on_stream = false;
void fg_set_input_on_stream(bool on) { g_input_on_stream = on; }
void fg_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *flacgpu_last_error(void) { return g_err.c_str(); }

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fg_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return false;                                                                    \
        }                                                                                    \
    } while (0)

bool DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return true;
    size_t want = std::max(bytes, cap + cap / 2);
    want = (want + 4095) & ~(size_t)4095;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) { fg_set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); return false; }
    if (p) (void)hipFree(p);
    p = np; cap = want;
    return true;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}

bool flacgpu_ctx::ensure_pinned(size_t bytes)
{
    if (bytes <= h_pin_cap) return true;
    size_t want = std::max(bytes, h_pin_cap * 2);
    void *np = nullptr;
    if (hipHostMalloc(&np, want, hipHostMallocDefault) != hipSuccess) { fg_set_error("hipHostMalloc failed"); return false; }
    if (h_pin) (void)hipHostFree(h_pin);
    h_pin = np; h_pin_cap = want;
    return true;
}

bool flacgpu_ctx::ensure_pinned_res(size_t bytes)
{
    if (bytes <= h_res_cap) return true;
    size_t want = std::max(bytes, h_res_cap * 2);
    void *np = nullptr;
    if (hipHostMalloc(&np, want, hipHostMallocDefault) != hipSuccess) { fg_set_error("hipHostMalloc failed"); return false; }
    if (h_res) (void)hipHostFree(h_res);
    h_res = np; h_res_cap = want;
    return true;
}

// ------------------------------------------------------------------ window tables
// Generated on the host with the C library's cosf, exactly as libFLAC does; tests/golden/window_hashes.json
// pins the tables so that a libm difference on another host is detected rather than silently encoded.
void fg_tukey_window(float *w, int32_t L, float p)
{
    for (int32_t n = 0; n < L; n++) w[n] = 1.0f;
    if (p <= 0.0f || p >= 1.0f) return;
    const int32_t Np = (int32_t)(p / 2.0f * L) - 1;
    if (Np > 0) {
        for (int32_t n = 0; n <= Np; n++) {
            w[n] = (float)(0.5f - 0.5f * cosf((float)(M_PI * n / Np)));
            w[L - Np - 1 + n] = (float)(0.5f - 0.5f * cosf((float)(M_PI * (n + Np) / Np)));
        }
    }
}

uint32_t flacgpu_ctx::window_offset(uint32_t n, uint32_t parts)
{
    for (const WindowEntry &e : win_index)
        if (e.n == n && e.parts == parts) return e.off;
    WindowEntry e;
    e.n = n; e.parts = parts; e.off = (uint32_t)h_windows.size();
    h_windows.resize(h_windows.size() + ((n + 3) & ~3u));
    const float p = parts >= 2 ? 0.5f / (float)(int32_t)parts : 0.5f;
    fg_tukey_window(h_windows.data() + e.off, (int32_t)n, p);
    // Self-check (hard part 1): the tapers come from this host's cosf, as they do in libFLAC, so a C library
    // whose cosf rounds differently would change encoder output silently.  The tapers of the preset shapes at the default
    // block size are committed (fg_window_tables.inc, the build container's glibc): a host that disagrees gets the committed
    // values and a note in flacgpu_last_error() / flacgpu_window_note().
    if (n == 4096 && parts <= 3) {
        const uint32_t *ref = parts >= 3 ? FG_TUKEY4096_P16 : parts == 2 ? FG_TUKEY4096_P25 : FG_TUKEY4096_P50;
        const uint32_t np = parts >= 3 ? FG_TUKEY4096_P16_NP : parts == 2 ? FG_TUKEY4096_P25_NP : FG_TUKEY4096_P50_NP;
        float *w = h_windows.data() + e.off;
        bool differs = fg_sel("FLACGPU_WINDOW_SELFTEST") != nullptr;      // (test hook: pretend the host's cosf disagrees)
        if (differs) for (uint32_t i = 0; i <= np; i++) { w[i] = 0.25f; w[n - np - 1 + i] = 0.25f; }
        for (uint32_t i = 0; i <= np && !differs; i++) {
            uint32_t a, b;
            memcpy(&a, &w[i], 4); memcpy(&b, &w[n - np - 1 + i], 4);
            if (a != ref[i] || b != ref[np + 1 + i]) differs = true;
        }
        if (differs) {
            for (uint32_t i = 0; i <= np; i++) { memcpy(&w[i], &ref[i], 4); memcpy(&w[n - np - 1 + i], &ref[np + 1 + i], 4); }
            window_note = "tukey window: this host's cosf differs from the committed table (glibc 2.35); the committed table is used";
            fg_set_error(window_note);
        }
    }
    win_index.push_back(e);
    windows_dirty = true;
    return e.off;
}

bool flacgpu_ctx::sync_windows()
{
    if (!windows_dirty) return true;
    HIPCHK(hipStreamSynchronize(stream));
    if (!windows.ensure(h_windows.size() * sizeof(float))) return false;
    HIPCHK(hipMemcpyAsync(windows.p, h_windows.data(), h_windows.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    windows_dirty = false;
    return true;
}

// ------------------------------------------------------------------ context
hipError_t fg_stream_wait(hipStream_t stream)
{
    static const long spin_us = fg_tune("FLACGPU_SPIN_US") ? atol(fg_tune("FLACGPU_SPIN_US")) : 3000;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(stream);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    return hipStreamSynchronize(stream);
}

bool flacgpu_ctx::wait_signal(unsigned long long seq)
{
    static const long spin_us = fg_tune("FLACGPU_SPIN_US") ? atol(fg_tune("FLACGPU_SPIN_US")) : 3000;
    volatile unsigned long long *flag = h_sig;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; it++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
            if ((it & 63) == 63 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return false;
    return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
}

extern "C" int flacgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static bool ctx_init(flacgpu_ctx *c, int device)
{
    c->device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&c->evp[i], hipEventDisableTiming));
    for (int i = 0; i < 3; i++) { HIPCHK(hipStreamCreateWithFlags(&c->gstream[i], hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&c->gev_join[i], hipEventDisableTiming)); }
    HIPCHK(hipEventCreateWithFlags(&c->gev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; i++) HIPCHK(hipEventCreateWithFlags(&c->gev_eval[i], hipEventDisableTiming));
    for (int i = 0; i < 3; i++) HIPCHK(hipEventCreateWithFlags(&c->evx[i], hipEventDisableTiming));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c->ev[i]));
    for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&c->evs[i]));
    if (hipHostMalloc((void **)&c->h_sig, 128, hipHostMallocDefault) != hipSuccess) { fg_set_error("hipHostMalloc failed"); return false; }
    memset(c->h_sig, 0, 128);
    if (!c->stamp.ensure(64)) return false;
    HIPCHK(hipMemset(c->stamp.p, 0, 64));
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_khz = (double)khz;
    }
    if (!c->crctab.ensure(FG_CRC_TABLE_WORDS * sizeof(uint16_t))) return false;      // (2048 entries of round 1-2 tables, then the direct packing path's two sets)
    // (made on the host, once per process, and copied: round 5's table kernel -- four waves walking powers of x bit by bit -- took
    // 5.5 ms of every context's creation)
    HIPCHK(hipMemcpy(c->crctab.p, fg_crc_tables_host(), FG_CRC_TABLE_WORDS * sizeof(uint16_t), hipMemcpyHostToDevice));
    // The encoder pipeline's autocorrelation runs its fp64 chains on the matrix core and is bit-exact only while that instruction
    // sums in v_fma_f64's order (flac_enc_pipe.hip fg_mfma_selfcheck): checked here, once per context.  A device that does it
    // differently keeps the byte-exact output -- every block then takes the generic kernel, whose chains are v_fma_f64 -- and says so.
    c->mfma_bad = fg_mfma_selfcheck(c->stream);
    if (c->mfma_bad != 0) {
        c->selfcheck_note = c->mfma_bad < 0 ? "matrix-core self-check could not run; the encoder uses its generic kernel"
                                            : "v_mfma_f64_4x4x4_4b_f64 does not sum like a v_fma_f64 chain on this device; the encoder uses its generic kernel";
        fg_set_error(c->selfcheck_note);
    }
    return true;
}

extern "C" flacgpu_ctx *flacgpu_ctx_create(int device)
{
    if (flacgpu_device_count() <= 0) { fg_set_error("no HIP device available: libflacgpu has no CPU fallback"); return nullptr; }
    flacgpu_ctx *c = new flacgpu_ctx();
    if (!ctx_init(c, device)) { delete c; return nullptr; }
    return c;
}

extern "C" void flacgpu_ctx_destroy(flacgpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->descs, &c->slots, &c->results, &c->dbg, &c->crctab, &c->windows, &c->offsets, &c->scratch_pcm,
                      &c->scratch_out, &c->dec_frames, &c->dec_results, &c->dec_scratch, &c->dec_subs, &c->dec_poff, &c->dec_hrec, &c->dec_prof, &c->dec_redo, &c->dec_info, &c->dec_off, &c->dec_rparams, &c->dec_warm, &c->dec_ranges, &c->pipe,
                      &c->mc_tmp, &c->mc_offs, &c->mc_map, &c->mc_sizes, &c->mc_res, &c->mc_foffs};
    for (DevBuf *b : bufs) b->release();
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_sig) (void)hipHostFree(c->h_sig);
    c->stamp.release();
    for (int i = 0; i < 4; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 8; i++) if (c->evs[i]) (void)hipEventDestroy(c->evs[i]);
    for (int i = 0; i < 3; i++) if (c->evx[i]) (void)hipEventDestroy(c->evx[i]);
    for (int i = 0; i < 2; i++) if (c->evp[i]) (void)hipEventDestroy(c->evp[i]);
    for (int i = 0; i < 3; i++) { if (c->gev_join[i]) (void)hipEventDestroy(c->gev_join[i]); if (c->gstream[i]) (void)hipStreamDestroy(c->gstream[i]); }
    if (c->gev_fork) (void)hipEventDestroy(c->gev_fork);
    for (int i = 0; i < 3; i++) if (c->gev_eval[i]) (void)hipEventDestroy(c->gev_eval[i]);
    c->lb.release();
    c->md5_jobs.release();
    if (c->md5_stream) (void)hipStreamDestroy(c->md5_stream);
    for (int i = 0; i < 2; i++) if (c->md5_ev[i]) (void)hipEventDestroy(c->md5_ev[i]);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static std::mutex g_default_mu;
static flacgpu_ctx *g_default = nullptr;
flacgpu_ctx *fg_default_ctx()
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        int dev = 0;
        const char *env = getenv("FLACGPU_DEVICE");
        if (env) dev = atoi(env);
        g_default = flacgpu_ctx_create(dev);
    }
    return g_default;
}

// ------------------------------------------------------------------ settings (stream_encoder.h:845-853)
extern "C" int flacgpu_settings_from_level(flacgpu_settings *s, uint32_t level, uint32_t channels, uint32_t bps,
                                           uint32_t sample_rate, uint32_t blocksize, int subset)
{
    static const struct { uint32_t ms, loose, parts, order, minpo, maxpo; } L[9] = {
        {0, 0, 0, 0, 0, 3}, {1, 1, 0, 0, 0, 3}, {1, 0, 0, 0, 0, 3}, {0, 0, 0, 6, 0, 4}, {1, 1, 0, 8, 0, 4},
        {1, 0, 0, 8, 0, 5}, {1, 0, 2, 8, 0, 6}, {1, 0, 2, 12, 0, 6}, {1, 0, 3, 12, 0, 6}};
    if (level > 8) level = 8;
    memset(s, 0, sizeof *s);
    s->channels = channels; s->bits_per_sample = bps; s->sample_rate = sample_rate; s->blocksize = blocksize;
    s->do_mid_side = L[level].ms; s->loose_mid_side = L[level].loose; s->apod_parts = L[level].parts;
    s->max_lpc_order = L[level].order; s->min_partition_order = L[level].minpo; s->max_partition_order = L[level].maxpo;
    s->streamable_subset = subset ? 1 : 0;
    s->qlp_coeff_precision = 0;
    return fg_resolve_settings(s);
}

// The checks and defaults init_stream applies, in libFLAC's order; returns a
// FLAC__StreamEncoderInitStatus value.
int fg_resolve_settings(flacgpu_settings *s)
{
    if (s->channels == 0 || s->channels > 8) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS;
    if (s->channels != 2) { s->do_mid_side = 0; s->loose_mid_side = 0; }
    else if (!s->do_mid_side) s->loose_mid_side = 0;
    const uint32_t bps = s->bits_per_sample;
    if (bps < 4 || bps > 32) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE;
    if (s->sample_rate > 1048575u) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE;
    if (s->blocksize == 0) s->blocksize = s->max_lpc_order == 0 ? 1152 : 4096;
    if (s->blocksize < 16 || s->blocksize > 65535) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE;
    if (s->max_lpc_order > 32) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER;
    if (s->blocksize < s->max_lpc_order) return FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER;
    if (s->qlp_coeff_precision == 0) {
        const uint32_t bs = s->blocksize;
        uint32_t q;
        if (bps < 16) { q = 2 + bps / 2; if (q < 5) q = 5; }
        else if (bps == 16) q = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
        else q = bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
        s->qlp_coeff_precision = q;
    }
    else if (s->qlp_coeff_precision < 5 || s->qlp_coeff_precision > 15)
        return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION;
    if (s->streamable_subset) {
        const uint32_t bs = s->blocksize, sr = s->sample_rate;
        if (bs > 16384 || (sr <= 48000 && bs > 4608)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
        if (sr >= 65536 && !(sr % 1000 == 0 || sr % 10 == 0)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
        if (bps != 8 && bps != 12 && bps != 16 && bps != 20 && bps != 24 && bps != 32) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
        if (s->max_partition_order > 8) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
        if (sr <= 48000 && (bs > 4608 || s->max_lpc_order > 12)) return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
    }
    if (s->max_partition_order >= 16) s->max_partition_order = 15;
    if (s->min_partition_order >= s->max_partition_order) s->min_partition_order = s->max_partition_order;
    return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
}

// ------------------------------------------------------------------ launch parameters
uint32_t fg_slot_bytes(const flacgpu_settings &s, uint32_t max_n)
{
    // a frame never exceeds its verbatim size by more than n/2 bits per subframe (+ parameters)
    uint64_t per = ((uint64_t)max_n * (s.bits_per_sample + 2)) / 8 + 512;
    uint64_t slot = 64 + (uint64_t)s.channels * per;
    return (uint32_t)((slot + 255) & ~(uint64_t)255);
}

void fg_fill_params(const flacgpu_settings &s, uint32_t max_n, bool pcm_i16, bool debug, FgEncParams *P)
{
    memset(P, 0, sizeof *P);
    P->channels = s.channels; P->bps = s.bits_per_sample; P->sample_rate = s.sample_rate; P->blocksize = s.blocksize;
    P->do_mid_side = (s.channels == 2 && s.do_mid_side) ? 1 : 0;
    P->max_lpc_order = s.max_lpc_order; P->qlp_precision = s.qlp_coeff_precision;
    P->min_po = s.min_partition_order; P->max_po = s.max_partition_order;
    P->apod_parts = s.apod_parts;
    P->limit_min_bitrate = s.limit_min_bitrate ? 1 : 0;
    P->rice_limit = s.bits_per_sample > 16 ? 31 : 15;
    P->slot_bytes = fg_slot_bytes(s, max_n);
    P->sig_stride = (max_n + 7) & ~7u;
    uint32_t nvec = 1;
    if (s.apod_parts >= 2) for (uint32_t b = 2; b <= s.apod_parts; b++) nvec += (b == 2) ? 2 : 2 * b;
    P->nvec = nvec;
    P->pcm_i16 = pcm_i16 ? 1 : 0;
    P->debug = debug ? 1 : 0;
    if (fg_tune("FLACGPU_STOP")) P->debug = 100 + atoi(fg_tune("FLACGPU_STOP"));
    const uint32_t mo = s.max_lpc_order ? s.max_lpc_order : 1;
    uint32_t dbuf = 4 * (FG_DH + FG_DK) * 8;
    const uint32_t lev = FG_MAX_CAND * nvec * mo * 12 + 64;
    if (lev > dbuf) dbuf = lev;
    P->lds_dbuf_bytes = (dbuf + 15) & ~15u;
}

// ------------------------------------------------------------------ batch encode
extern "C" uint64_t flacgpu_encode_bound(const flacgpu_settings *s, const flacgpu_stream_desc *streams, uint32_t nstreams,
                                         uint32_t *nblocks)
{
    uint64_t nb = 0;
    for (uint32_t i = 0; i < nstreams; i++) nb += (streams[i].nsamples + s->blocksize - 1) / s->blocksize;
    if (nblocks) *nblocks = (uint32_t)nb;
    return nb * fg_slot_bytes(*s, s->blocksize);
}

extern "C" void flacgpu_set_debug(flacgpu_ctx *ctx, int on) { ctx->debug = on != 0; }
extern "C" const char *flacgpu_window_note(flacgpu_ctx *ctx) { return ctx->window_note.c_str(); }

// ---- the CRC-16 tables of the kernels (polynomial x^16 + x^15 + x^2 + 1, format.h:447: FLAC__crc16), built on the host
// Layout (uint16 entries): [0,256) the CRC of byte i; [256,512) (i x^8) x^2048 and [512,768) i x^2048 -- the contribution of a
// byte 2048 bits further up --; [768,832) x^(32 k); [1024,1792) byte i followed by 1, 2, 3 zero bytes (slicing by four).  Then, for the
// direct packing path (fg_pipe_pack_kernel<DIRECT>: a pass of NT threads over 16-byte granules), a set for NT = 256 at 2048 and one
// for NT = 128 at 2048 + 5632: [0,256) (i x^8) x^(128 NT), [256,512) i x^(128 NT), [512,1536) byte i followed by 3, 2, 1, 0 zero bytes,
// [1536 + rem NT + t) x^(128 (NT - 1 - t) + 8 rem) for rem = 0..15.
// Products and powers in GF(2)[x] / P: a power of x by squaring, so the whole table is some 14 000 products of sixteen steps.
static inline uint32_t fg_gf16_mul(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000u) ? (((r << 1) ^ 0x8005u) & 0xFFFFu) : ((r << 1) & 0xFFFFu);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
static uint32_t fg_gf16_xpow(uint32_t n)
{
    uint32_t r = 1, base = 2;           // (the polynomial x)
    for (; n; n >>= 1) { if (n & 1u) r = fg_gf16_mul(r, base); base = fg_gf16_mul(base, base); }
    return r;
}
const uint16_t *fg_crc_tables_host()
{
    static uint16_t tab[FG_CRC_TABLE_WORDS];
    static std::once_flag once;
    std::call_once(once, [] {
        memset(tab, 0, sizeof tab);
        const uint32_t x8 = fg_gf16_xpow(8), x16 = fg_gf16_xpow(16), x24 = fg_gf16_xpow(24), x2048 = fg_gf16_xpow(2048);
        uint32_t crc8[256];
        for (uint32_t i = 0; i < 256; i++) {
            const uint32_t c = fg_gf16_mul(i << 8, x8);        // byte i, its eight bits shifted out: i x^16 mod P
            crc8[i] = c;
            tab[i] = (uint16_t)c;
            tab[256 + i] = (uint16_t)fg_gf16_mul(i << 8, x2048);
            tab[512 + i] = (uint16_t)fg_gf16_mul(i, x2048);
            if (i < 64) tab[768 + i] = (uint16_t)fg_gf16_xpow(32 * i);
            tab[1024 + i] = (uint16_t)fg_gf16_mul(c, x8);
            tab[1280 + i] = (uint16_t)fg_gf16_mul(c, x16);
            tab[1536 + i] = (uint16_t)fg_gf16_mul(c, x24);
        }
        for (int set = 0; set < 2; set++) {
            const uint32_t nt = set == 0 ? 256u : 128u;
            uint16_t *x = tab + 2048 + 5632 * set;
            const uint32_t xs = fg_gf16_xpow(128 * nt);
            for (uint32_t i = 0; i < 256; i++) {
                const uint32_t c = crc8[i];
                x[i] = (uint16_t)fg_gf16_mul(i << 8, xs);
                x[256 + i] = (uint16_t)fg_gf16_mul(i, xs);
                x[512 + i] = (uint16_t)fg_gf16_mul(c, x24);
                x[768 + i] = (uint16_t)fg_gf16_mul(c, x16);
                x[1024 + i] = (uint16_t)fg_gf16_mul(c, x8);
                x[1280 + i] = (uint16_t)c;
                if (i < nt)
                    for (uint32_t rem = 0; rem < 16; rem++) x[1536 + rem * nt + i] = (uint16_t)fg_gf16_xpow(128 * (nt - 1 - i) + 8 * rem);
            }
        }
    });
    return tab;
}
// (for the CPU test that holds this table against the bit-by-bit definition)
extern "C" void flacgpu_debug_crc_tables(uint16_t *out) { memcpy(out, fg_crc_tables_host(), FG_CRC_TABLE_WORDS * sizeof(uint16_t)); }
#include "fg_build_id.inc"
extern "C" const char *flacgpu_build_id(void) { return FG_BUILD_ID; }
extern "C" const char *flacgpu_kernel_id(void) { return FG_KERNEL_ID; }
extern "C" const char *flacgpu_host_id(void) { return FG_HOST_ID; }
extern "C" unsigned int flacgpu_build_flags(void)
{
    unsigned int f = 0;
#ifdef FG_TUNING
    f |= 1u;
#endif
#ifdef FG_TESTHOOKS
    f |= 4u;
#endif
    return f;
}
extern "C" int flacgpu_selfcheck(flacgpu_ctx *ctx, const char **note)
{
    if (note) *note = ctx->selfcheck_note.c_str();
    return ctx->mfma_bad;
}
extern "C" void flacgpu_force_selfcheck_result(flacgpu_ctx *ctx, int mfma_bad)
{
    ctx->mfma_bad = mfma_bad;
    ctx->selfcheck_note = mfma_bad ? "matrix-core self-check overridden (flacgpu_force_selfcheck_result); the encoder uses its generic kernel" : "";
    ctx->desc_key.clear();       // (the kernel choice is part of what a cached block list stands for)
}
extern "C" void flacgpu_set_stage_timing(flacgpu_ctx *ctx, int level) { ctx->stage_timing = level < 0 ? 0 : level > 3 ? 3 : level; }
extern "C" void flacgpu_set_log_guard(flacgpu_ctx *ctx, double thr) { ctx->log_guard_thr = thr; }
extern "C" void flacgpu_set_direct(flacgpu_ctx *ctx, int on) { ctx->direct = on < 0 ? 0 : on > 2 ? 2 : on; }

extern "C" int flacgpu_copy_debug(flacgpu_ctx *c, void *dst, uint32_t first, uint32_t n)
{
    if (!c->dbg.p || first + n > c->last_nblocks) return -1;
    (void)hipSetDevice(c->device);
    return hipMemcpy(dst, (char *)c->dbg.p + (size_t)first * sizeof(FgDebugRec), (size_t)n * sizeof(FgDebugRec), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

extern "C" int flacgpu_copy_block_results(flacgpu_ctx *c, void *dst, uint32_t n)
{
    if (!c->results.p || n > c->last_nblocks) return -1;
    (void)hipSetDevice(c->device);
    return hipMemcpy(dst, c->results.p, (size_t)n * sizeof(FgBlockResult), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

// `view` > 0: the streams are ONE-channel views of streams of `view` interleaved channels -- entry j stands for channel j % view of
// the caller's stream j / view (its pcm_offset in inter-channel samples as usual): FgBlockDesc.pcm_off becomes an element offset
// and FgBlockDesc.reserved the element stride (encode_multichannel below).
static bool encode_streams_impl(flacgpu_ctx *c, const flacgpu_settings *s, const void *d_pcm, int pcm_is_i16,
                                const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_out, uint64_t out_cap,
                                void *d_offsets, flacgpu_encode_stats *st, uint32_t view = 0)
{
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (s->blocksize < 16 || s->blocksize > 65535) { fg_set_error("invalid blocksize"); return false; }
    if (s->max_partition_order > 8) { fg_set_error("max_residual_partition_order > 8 is not supported by the GPU path"); return false; }
    if (s->apod_parts > 3) { fg_set_error("subdivide_tukey parts > 3 is not supported by the GPU path"); return false; }
    if (s->channels < 1 || s->channels > 8 || s->bits_per_sample < 4 || s->bits_per_sample > 32 || s->qlp_coeff_precision < 5 ||
        s->qlp_coeff_precision > 15 || s->max_lpc_order > 32) {
        fg_set_error("invalid encoder settings"); return false;
    }
    // block list.  A repeated call with the same settings and stream list (and no per-call frame decisions: loose
    // mid-side rewrites the list) reuses the ordered list of the previous call, already on the device.
    std::vector<unsigned char> key;
    {
        const size_t sb = sizeof *s, tb = (size_t)nstreams * sizeof *streams;
        key.resize(sb + tb + 1);
        memcpy(key.data(), s, sb);
        if (tb) memcpy(key.data() + sb, streams, tb);
        // (the kernel selection switches -- tuning aids read per call -- decide which blocks count as pipeline blocks: part of the key)
        key[sb + tb] = (unsigned char)((pcm_is_i16 ? 1 : 0) | (fg_sel("FLACGPU_NO_FAST") ? 2 : 0) | (view << 4) | (c->direct ? 4 : 0) | (c->direct >= 2 ? 128 : 0) |
                                       ((fg_sel("FLACGPU_WS") && atoi(fg_sel("FLACGPU_WS")) == 1) ? 8 : 0));
    }
    const bool reuse = !(s->do_mid_side && s->loose_mid_side) && !c->debug && c->dev_descs_ptr == c->descs.p && c->dev_descs_ptr != nullptr &&
                       !c->dev_descs.empty() && key == c->desc_key;
    std::vector<FgBlockDesc> built;
    uint64_t nb = 0;
    for (uint32_t i = 0; i < nstreams; i++) nb += (streams[i].nsamples + s->blocksize - 1) / s->blocksize;
    if (!reuse) built.reserve(nb);
    uint32_t win_n = 0, win_off = 0;                     // the window table is looked up when the block length changes
    for (uint32_t i = 0; i < nstreams && !reuse; i++) {
        uint64_t pos = 0;
        uint32_t fn = streams[i].first_frame;
        while (pos < streams[i].nsamples) {
            FgBlockDesc d;
            const uint64_t left = streams[i].nsamples - pos;
            d.pcm_off = view ? (streams[i].pcm_offset + pos) * view + (i % view) : streams[i].pcm_offset + pos;
            d.n = (uint32_t)std::min<uint64_t>(left, s->blocksize);
            d.frame_number = fn++;
            if (s->max_lpc_order && d.n != win_n) { win_n = d.n; win_off = c->window_offset(d.n, s->apod_parts); }
            d.win_off = s->max_lpc_order ? win_off : 0;
            d.forced_ca = 0xFF;
            d.out_slot = (uint32_t)built.size();
            d.reserved = view;
            built.push_back(d);
            pos += d.n;
        }
    }
    std::vector<FgBlockDesc> &descs = reuse ? c->dev_descs : built;
    const uint32_t nblocks = (uint32_t)descs.size();
    memset(st, 0, sizeof *st);
    st->nblocks = nblocks;
    st->lpc_order_min_margin = INFINITY;        // (no LPC order guess looked at yet)
    if (nblocks == 0) return true;
    if (s->max_lpc_order == 0 && c->h_windows.empty()) c->window_offset(16, 0);
    if (!c->sync_windows()) return false;
    FgEncParams P;
    fg_fill_params(*s, s->blocksize, pcm_is_i16 != 0, c->debug, &P);
    if (fg_enc_lds_bytes(&P) > 64 * 1024) {
        // large blocks: the generic kernel reads the PCM in place instead of staging it in LDS
        if (pcm_is_i16) { fg_set_error("int16 ingest needs blocks that fit LDS"); return false; }
        P.sig_stride = 0;
        if (fg_enc_lds_bytes(&P) > 160 * 1024) { fg_set_error("settings need more LDS than the device has"); return false; }
    }
    // ---- which kernels: the de-fused pipeline (flac_enc_pipe_impl.h) where it applies, round 1's single kernel with
    // FLACGPU_PIPE=0, the generic kernel for everything else
    const bool cfg_fast = !fg_sel("FLACGPU_NO_FAST") && c->mfma_bad == 0 && P.sig_stride != 0 && s->channels <= 2 && s->max_lpc_order <= 12 &&
                          // (32-bit streams: blocks whose channels share eight wasted bits, flac_enc_pipe_impl.h pipe_preshift; the rest is handed
                          // to the generic kernel block by block, which the decision probe of loose mid-side does not expect)
                          (s->bits_per_sample <= 24 || (s->bits_per_sample == 32 && !(s->do_mid_side && s->loose_mid_side)));
    const bool use_pipe = cfg_fast && fg_pipe_supported(&P);
    if (view && !use_pipe && cfg_fast) { fg_set_error("one-channel views need the pipeline or the generic kernel"); return false; }
    const bool ws1_only = fg_sel("FLACGPU_WS") && atoi(fg_sel("FLACGPU_WS")) == 1;   // tuning aid: never two packing waves per subframe
    const uint32_t nw = 4;                            // chunk slots per frame (channels x packing waves per subframe, <= 4)
    uint32_t chunk_cap_words = 0, fbw_words = s->bits_per_sample <= 16 ? 800 : 1280;   // 16-bit stereo: 5 workgroups per CU
    // (FLACGPU_FBW, test-hooks builds: a smaller window -- a lane whose codes do not fit it hands its block to the generic kernel, which
    // is how the tests reach that path now that every content class stays in the pipeline; the bytes are the same)
    if (fg_sel("FLACGPU_FBW")) { fbw_words = (uint32_t)atoi(fg_sel("FLACGPU_FBW")); if (fbw_words < 24) fbw_words = 24; }
    if (use_pipe) {
        // a chunk holds at most a whole subframe (all of a subframe's bits may sit in one half) plus the frame header
        const uint64_t per = ((uint64_t)s->blocksize * (s->bits_per_sample + 2)) / 8 + 512 + 64;
        chunk_cap_words = (uint32_t)((per + 255) / 256 * 64);      // whole 256-byte rows: chunks start on a row boundary
        const uint32_t need = s->channels * 2 * chunk_cap_words * 4;
        if (need > P.slot_bytes) P.slot_bytes = (need + 255) & ~255u;
    }
    if (!c->descs.ensure((size_t)nblocks * sizeof(FgBlockDesc))) return false;
    if (!c->slots.ensure((size_t)nblocks * P.slot_bytes)) return false;
    if (!c->results.ensure((size_t)nblocks * sizeof(FgBlockResult))) return false;
    if (!c->offsets.ensure(fg_scan_words(nblocks) * 8)) return false;
    FgPipeLaunch PL;
    memset(&PL, 0, sizeof PL);
    bool guard_was_clean = false;
    if (use_pipe) {
        if (!c->pipe.ensure(fg_pipe_scratch_bytes(&P, nblocks))) return false;
        fg_pipe_carve(&P, nblocks, c->pipe.p, &PL.B);
        // (were the counters the kernels add to left reset by the last call's signal kernel?  Whatever this call launches first
        // -- a loose mid-side probe, the launch proper -- uses that up: FgPipeLaunch.guard_clean)
        guard_was_clean = PL.B.guard && c->guard_clean_ptr == PL.B.guard;
        c->guard_clean_ptr = nullptr;            // (set again by a call that ends through the signal kernel)
        // near-tie guard of the LPC order guess: count, smallest margin (as the bits of a positive double); the
        // autocorrelation kernel resets both
        PL.guard_thr = c->log_guard_thr;
    }
    FgDebugRec *dbg = nullptr;
    if (c->debug) {
        if (!c->dbg.ensure((size_t)nblocks * sizeof(FgDebugRec))) return false;
        HIPCHK(hipMemsetAsync(c->dbg.p, 0, (size_t)nblocks * sizeof(FgDebugRec), c->stream));
        dbg = (FgDebugRec *)c->dbg.p;
    }
    // can the pipeline / the single-wave kernel take this block?  (lane = segment: enough samples per lane for the
    // predictor history, partitions no finer than a lane)
    auto block_fast = [&](const FgBlockDesc &d) -> bool {
        if (!cfg_fast) return false;
        // (the pipeline has a lane geometry for ragged blocks -- the tail of nearly every real stream --; round 1's single kernel,
        // FLACGPU_PIPE=0, wants whole lanes of 16 samples)
        if (use_pipe) return fg_pipe_block_ok(d.n, s->max_partition_order) != 0;
        if (d.n < 64 * 16 || (d.n % 64) != 0) return false;
        uint32_t pm = 0, b = d.n;
        while (!(b & 1)) { pm++; b >>= 1; }
        if (pm > s->max_partition_order) pm = s->max_partition_order;
        return pm <= 6;
    };
    auto block_ws = [&](const FgBlockDesc &d) -> uint32_t { const uint32_t w = fg_pipe_block_ws(d.n); return (w == 2 && ws1_only) ? 1u : w; };
    PL.pcm = d_pcm; PL.descs = (const FgBlockDesc *)c->descs.p; PL.windows = (const float *)c->windows.p; PL.P = P;
    PL.slots = (uint8_t *)c->slots.p; PL.results = (FgBlockResult *)c->results.p; PL.dbg = dbg;
    PL.chunk_cap_words = chunk_cap_words; PL.fbw_words = fbw_words; PL.acc64 = s->bits_per_sample > 16 ? 1 : 0;
    PL.stream = (void *)c->stream; PL.stream2 = (void *)c->stream3; PL.ev_fork = (void *)c->evp[0]; PL.ev_join = (void *)c->evp[1];
    for (int i = 0; i < 3; i++) { PL.gstream[i] = (void *)c->gstream[i]; PL.gev_join[i] = (void *)c->gev_join[i]; }
    PL.gev_fork = (void *)c->gev_fork;
    // the block list goes to the device once per distinct layout (repeated calls with the same streams skip the copy)
    auto upload_descs = [&](const std::vector<FgBlockDesc> &v, bool cache) -> bool {
        const size_t bytes = v.size() * sizeof(FgBlockDesc);
        if (cache && c->dev_descs_ptr == c->descs.p && c->dev_descs.size() == v.size() && memcmp(c->dev_descs.data(), v.data(), bytes) == 0) return true;
        if (hipMemcpyAsync(c->descs.p, v.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { fg_set_error("H2D of the block list failed"); return false; }
        // the pageable copy is staged before the call returns, so the vector may be reused; remember what is there
        if (cache) { c->dev_descs = v; c->dev_descs_ptr = c->descs.p; }
        else { c->dev_descs.clear(); c->dev_descs_ptr = nullptr; }
        return true;
    };
    // loose mid-side (levels 1, 4; stream_encoder.h:826-838): libFLAC decides between independent and mid-side coding on
    // every `period`-th frame OF THE STREAM (frame number % period == 0) and the frames in between copy that decision, so
    // the phase follows the absolute frame number and a continuation call starts with the assignment the previous call
    // ended on.  The decision frames are independent of each other: a probe pass analyses them, the host spreads the result.
    if (P.do_mid_side && s->loose_mid_side) {
        c->desc_key.clear();
        uint32_t period = (uint32_t)((double)s->sample_rate * 0.4 / (double)s->blocksize + 0.5);
        if (period == 0) period = 1;
        std::vector<FgBlockDesc> dec, decp, decr, decg;      // decision frames: all; pipeline (regular, then ragged geometry) first, generic after
        std::vector<uint32_t> decidx;
        auto is_rag = [&](const FgBlockDesc &d) { return (d.n % 64) != 0 || d.n / 64 < 16; };
        for (uint32_t b = 0; b < nblocks; b++)
            if (descs[b].frame_number % period == 0) {
                decidx.push_back(b);
                if (block_fast(descs[b]) && use_pipe) (is_rag(descs[b]) ? decr : decp).push_back(descs[b]);
                else decg.push_back(descs[b]);
            }
        const uint32_t nrag_probe = (uint32_t)decr.size();
        decp.insert(decp.end(), decr.begin(), decr.end());
        dec = decp; dec.insert(dec.end(), decg.begin(), decg.end());
        for (size_t k = 0; k < dec.size(); k++) dec[k].reserved = dec[k].out_slot;      // remember the block, probe into slot k
        for (size_t k = 0; k < dec.size(); k++) dec[k].out_slot = (uint32_t)k;
        std::vector<FgBlockResult> r(dec.size());
        if (!dec.empty()) {
            std::vector<FgBlockDesc> up = dec;
            for (FgBlockDesc &d : up) d.reserved = 0;
            if (!upload_descs(up, false)) return false;
            if (!decp.empty()) {
                guard_was_clean = false;                 // (the probe adds to the counters; it starts with the begin kernel itself)
                PL.nblocks = (uint32_t)decp.size(); PL.nblocks_rag = nrag_probe; PL.stages = 1; PL.dbg = nullptr;
                if (fg_launch_encode_pipe(&PL) != 0) { fg_set_error("encode pipeline launch failed"); return false; }
                PL.dbg = dbg; PL.nblocks_rag = 0;
            }
            if (!decg.empty() &&
                fg_launch_encode(d_pcm, (const FgBlockDesc *)c->descs.p + decp.size(), (const float *)c->windows.p, &P, (uint32_t)decg.size(),
                                 (uint8_t *)c->slots.p, (FgBlockResult *)c->results.p, nullptr, (const uint16_t *)c->crctab.p, c->stream) != 0) {
                fg_set_error("encode kernel launch failed"); return false;
            }
            HIPCHK(hipMemcpyAsync(r.data(), c->results.p, dec.size() * sizeof(FgBlockResult), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        std::vector<uint32_t> decision(nblocks, 0xFFFFFFFFu);
        for (size_t k = 0; k < dec.size(); k++) {
            const FgBlockResult &rr = r[k];
            decision[dec[k].reserved] = (rr.best_bits[2] + rr.best_bits[3]) < (rr.best_bits[0] + rr.best_bits[1]) ? 3u : 0u;
        }
        size_t bi = 0;
        for (uint32_t i = 0; i < nstreams; i++) {
            const uint64_t cnt = (streams[i].nsamples + s->blocksize - 1) / s->blocksize;
            uint32_t last = streams[i].prev_channel_assignment == 3 ? 3u : 0u;
            for (uint64_t k = 0; k < cnt; k++) {
                const bool decides = decision[bi + k] != 0xFFFFFFFFu;
                if (decides) last = decision[bi + k];
                descs[bi + k].forced_ca = (decides && last == 3) ? (3u | 0x80u) : last;
            }
            bi += cnt;
            st->last_channel_assignment = last;
        }
    }
    // blocks the specialised kernels cover go first, the rest to the generic kernel (same bytes either way)
    uint32_t nfast = 0;
    // (direct packing, FgPackDirect: does a block that keeps the chunk form / takes the generic kernel lie in front of a block of two
    // packing waves per subframe in the output?  Its size is then waited for: FgPipeLaunch.side_first)
    uint32_t side_first = 0, slow_first = 0;
    if (reuse) { nfast = c->desc_nfast; PL.nblocks_ws2 = c->desc_nws2; PL.nblocks_rag = c->desc_nrag; side_first = c->desc_side_first; slow_first = c->desc_slow_first; }
    else {
        std::vector<FgBlockDesc> ordered;
        ordered.reserve(nblocks);
        std::vector<FgBlockDesc> slow, ws1, rag;
        for (const FgBlockDesc &d : descs) {
            if (!block_fast(d)) slow.push_back(d);
            else if (use_pipe && ((d.n % 64) != 0 || d.n / 64 < 16)) rag.push_back(d);      // ragged lane geometry (tails, odd sizes; is_rag above)
            else if (use_pipe && block_ws(d) != 2) ws1.push_back(d);
            else ordered.push_back(d);
        }
        if (use_pipe && (!ordered.empty() || !ws1.empty())) {
            // (direct: the blocks of the regular lane geometry; every class keeps the output order)
            uint32_t last_direct = 0;
            if (!ordered.empty()) last_direct = ordered.back().out_slot;
            if (!ws1.empty() && ws1.back().out_slot > last_direct) last_direct = ws1.back().out_slot;
            if (!rag.empty() && rag.front().out_slot < last_direct) side_first = 1;
            if (!slow.empty() && slow.front().out_slot < last_direct) slow_first = 1;
        }
        PL.nblocks_ws2 = use_pipe ? (uint32_t)ordered.size() : 0;
        ordered.insert(ordered.end(), ws1.begin(), ws1.end());
        PL.nblocks_rag = (uint32_t)rag.size();
        ordered.insert(ordered.end(), rag.begin(), rag.end());
        nfast = (uint32_t)ordered.size();
        ordered.insert(ordered.end(), slow.begin(), slow.end());
        descs.swap(ordered);
        c->desc_key.clear();
        if (!upload_descs(descs, true)) return false;
        if (!(P.do_mid_side && s->loose_mid_side)) {
            c->desc_key = key; c->desc_nfast = nfast; c->desc_nws2 = PL.nblocks_ws2; c->desc_nrag = PL.nblocks_rag;
            c->desc_side_first = side_first; c->desc_slow_first = slow_first;
        }
    }
    // ---- direct packing (round 5): the blocks packed by two waves per subframe assemble their frames in LDS and write them at their
    // final place; no sizes scan, no chunks through HBM, no assembly kernel for them.  <= 16 bit and blocks of up to 4608 samples (the
    // frame buffer beside the staged samples leaves four workgroups a CU), not for one-channel views (the splice wants the chunks).
    // Round 6: 17..24-bit input too (the kernel's 64-bit forms; the chunk form's four windows took as much LDS as the one frame buffer
    // does -- two workgroups a CU either way), and 32-bit input in blocks of up to 4096 samples (a frame of 33-bit verbatim samples is
    // 34 KB: 74 KB a workgroup, still two a CU); FLACGPU_DIRECT24=0 in a test-hooks build keeps the chunk form for all of them.
    const bool direct24_off = fg_sel("FLACGPU_DIRECT24") && atoi(fg_sel("FLACGPU_DIRECT24")) == 0;
    bool direct = use_pipe && c->direct != 0 && d_out != nullptr && view == 0 && s->blocksize <= 4608 &&
                  (s->bits_per_sample <= 16 || ((s->bits_per_sample <= 24 || s->blocksize <= 4096) && !direct24_off)) &&
                  nfast > PL.nblocks_rag && !slow_first && !ws1_only;
    uint32_t direct_fcap = 0;
    bool lb_cleared = false;           // (queued on the main stream: the other streams of the launch wait for it)
    if (direct) {
        const uint64_t fbits = 16 * 8 + (uint64_t)s->channels * (8 + 32 + (uint64_t)s->blocksize * (s->bits_per_sample + 1));
        direct_fcap = (uint32_t)(((fbits + 31) / 32 + 2 + 3) & ~3ull);
        if (fg_tune("FLACGPU_FCAP")) direct_fcap = (uint32_t)atoi(fg_tune("FLACGPU_FCAP")) & ~3u;      // (occupancy experiments: larger frames fall back)
        const void *before = c->lb.p;
        if (!c->lb.ensure((size_t)nblocks * 8)) return false;
        // (a word counts only with this call's epoch: a fresh buffer, or the epoch counter back at its start, is cleared)
        c->lb_epoch = (c->lb_epoch + 1) & 0xFFFFFu;
        if (c->lb.p != before || c->lb_epoch == 0) {
            HIPCHK(hipMemsetAsync(c->lb.p, 0, c->lb.cap, c->stream));
            if (c->lb_epoch == 0) c->lb_epoch = 1;
            lb_cleared = true;
        }
    }
    // How the call ends and what is timed.  Level 0 (default): no events -- every event record costs a few microseconds of
    // idle GPU between two kernels -- a stamp kernel in front, a signal kernel at the end (fg_signal_kernel: totals and
    // stamps into pinned memory, the host polls the sequence number).  Level 1: HIP events around the call and the
    // encode kernels.  Level 2: also between the pipeline's stages.
    // Level 3: as level 0, plus ONE event in front of the call's first kernel and one behind its last (in front of the signal
    // kernel): total_gpu_ms is then the HIP-event time of exactly the kernels the default call runs.
    const bool lean = c->stage_timing == 0 || c->stage_timing == 3;
    const bool ev2 = c->stage_timing == 3;
    const bool timing = c->stage_timing == 2 && use_pipe;
    int nev = 0;
    auto mark = [&]() { if (timing && nev < 8) (void)hipEventRecord(c->evs[nev++], c->stream); };
    PL.B.stamp = nullptr;
    if (ev2) HIPCHK(hipEventRecord(c->ev[0], c->stream));
    // (several groups: no begin kernel when the last call's signal kernel left the counters reset, no fork event when nothing this
    // call put on the main stream concerns the other groups -- FgPipeLaunch.guard_clean / no_fork)
    static const bool quick_off = fg_sel("FLACGPU_QUICK_START") && atoi(fg_sel("FLACGPU_QUICK_START")) == 0;
    PL.guard_clean = (lean && use_pipe && !quick_off && guard_was_clean) ? 1u : 0u;
    PL.no_fork = (PL.guard_clean && reuse && !c->debug && !lb_cleared && !g_input_on_stream) ? 1u : 0u;
    if (lean) {
        // (the pipeline's first kernel takes the stamp itself)
        if (nfast && use_pipe) PL.B.stamp = (unsigned long long *)c->stamp.p;
        else if (fg_launch_stamp((unsigned long long *)c->stamp.p, c->stream) != 0) { fg_set_error("stamp kernel launch failed"); return false; }
    }
    else HIPCHK(hipEventRecord(c->ev[0], c->stream));
    const bool side = nfast > 0 && nblocks > nfast;     // overlap the few generic blocks with the fast launch
    if (side) {
        HIPCHK(hipEventRecord(c->evx[0], c->stream));
        HIPCHK(hipStreamWaitEvent(c->stream2, c->evx[0], 0));
    }
    bool piped = false;
    auto set_direct = [&](bool on) {
        PL.nblocks_direct = on ? nfast - PL.nblocks_rag : 0;        // (the blocks of the regular lane geometry, whether one or two waves pack a subframe)
        PL.side_first = on ? side_first : 0;
        PL.fused = (on && c->direct >= 2) ? 1u : 0u;
        PL.side_stream = (void *)c->stream3; PL.side_ev = (void *)c->evp[0];
        for (int i = 0; i < 3; i++) PL.gev_eval[i] = (void *)c->gev_eval[i];
        PL.D.lb = (unsigned long long *)c->lb.p; PL.D.offsets = (unsigned long long *)c->offsets.p;
        PL.D.user_offsets = (unsigned long long *)d_offsets; PL.D.dst = (uint8_t *)d_out; PL.D.dst_cap = out_cap;
        PL.D.crcx = (const uint16_t *)c->crctab.p + 2048; PL.D.epoch = c->lb_epoch; PL.D.nblocks = nblocks; PL.D.fcap_words = direct_fcap;
        PL.D.reserved = fg_tune("FLACGPU_DIRECT_X") ? (uint32_t)atoi(fg_tune("FLACGPU_DIRECT_X")) : 0u;
    };
    set_direct(direct);
    // (FLACGPU_AUTOC1, test-hooks builds: 0 = the wave-a-block kernel for every launch, 2 = the workgroup-a-block kernel for every launch,
    // 3 = that kernel with its windows one behind the other also where they would run side by side)
    PL.no_autoc1 = fg_sel("FLACGPU_AUTOC1") ? (atoi(fg_sel("FLACGPU_AUTOC1")) == 0 ? 1u : (atoi(fg_sel("FLACGPU_AUTOC1")) == 2 ? 2u : (atoi(fg_sel("FLACGPU_AUTOC1")) == 3 ? 3u : 0u))) : 0u;
    if (nfast && use_pipe) {
        PL.nblocks = nfast;
        if (timing) {
            // one launch per stage group so that the events land between the kernels
            mark();
            PL.stages = 1;
            // (analysis = autocorrelation + Levinson-Durbin + evaluation; the split inside is in the rocprof trace)
            if (fg_launch_encode_pipe(&PL) != 0) { fg_set_error("encode pipeline launch failed"); return false; }
            mark();
            PL.stages = 2;
            if (fg_launch_encode_pipe(&PL) != 0) { fg_set_error("encode pipeline launch failed"); return false; }
            mark();
        }
        else {
            PL.stages = 3;
            // (groups: flac_enc_pipe_impl.h / pipe_shape.inc; a launch of a few hundred blocks does not fill the chip once)
            static const int groups_env = fg_sel("FLACGPU_GROUPS") ? atoi(fg_sel("FLACGPU_GROUPS")) : 0;
            // (direct packing: the packing kernels of two groups run one behind the other anyway -- pipe_shape.inc --, and one chain of
            // whole-launch kernels then is the faster form: 0.431-0.444 ms against 0.448-0.458 on the headline stream)
            PL.ngroups = groups_env > 0 ? (uint32_t)groups_env : ((nfast >= 4096 && !direct) ? 2u : 1u);
            static const bool keep_off = fg_sel("FLACGPU_KEEP") && atoi(fg_sel("FLACGPU_KEEP")) == 0;
            PL.no_keep = keep_off ? 1u : 0u;
            if (c->debug) PL.ngroups = 1;
            if (fg_launch_encode_pipe(&PL) != 0) { fg_set_error("encode pipeline launch failed"); return false; }
            PL.ngroups = 1;
        }
        piped = true;
    }
    else if (nfast) nfast = 0;       // (not reached: without the pipeline no block counts as fast)
    if (nblocks > nfast) {
        hipStream_t ss = (side && nfast) ? c->stream2 : c->stream;
        if (fg_launch_encode(d_pcm, (const FgBlockDesc *)c->descs.p + nfast, (const float *)c->windows.p, &P, nblocks - nfast, (uint8_t *)c->slots.p,
                             (FgBlockResult *)c->results.p, dbg, (const uint16_t *)c->crctab.p, ss) != 0) {
            fg_set_error("encode kernel launch failed"); return false;
        }
        if (ss == c->stream2) {
            HIPCHK(hipEventRecord(c->evx[1], c->stream2));
            HIPCHK(hipStreamWaitEvent(c->stream, c->evx[1], 0));
        }
    }
    if (!lean) HIPCHK(hipEventRecord(c->ev[1], c->stream));
    // sizes -> offsets -> contiguous output, all queued behind the encode kernels; the host looks at the totals once, at the
    // end.  (Frames that would not fit `out_cap` are skipped; blocks the specialised kernels handed back show up as
    // FG_ERR_REDO in the flags and are redone below, which repeats the scan and the assembly: rare.)
    if (!c->ensure_pinned_res(64)) return false;
    unsigned long long *tail = (unsigned long long *)c->h_res;   // total bytes, OR of error flags
    auto finish_pass = [&](bool first) -> bool {
        if (direct && first) {
            // direct packing: the frames of the direct blocks are in place.  The blocks that kept the chunk form (and the frames of the
            // generic kernel, whose sizes are published here) find their places through the look-back words and are assembled now;
            // the first wave of that kernel also leaves the error flags beside the total.
            const uint32_t nd = PL.nblocks_direct;
            if (nblocks > nfast && fg_launch_pipe_publish((const FgBlockDesc *)c->descs.p, nfast, nblocks - nfast, (const FgBlockResult *)c->results.p,
                                                          PL.B.chunk_bits, &PL.D, c->stream) != 0) { fg_set_error("publish kernel launch failed"); return false; }
            mark();
            const bool rest = nd < nblocks;          // (no block in the chunk form: the signal kernel fetches flags and counters itself)
            if (rest && fg_launch_pipe_assemble((const FgBlockDesc *)c->descs.p, nblocks, (const uint8_t *)c->slots.p, P.slot_bytes, chunk_cap_words, nw,
                                                PL.B.chunk_bits, (FgBlockResult *)c->results.p, (unsigned long long *)c->offsets.p, (uint8_t *)d_out, out_cap,
                                                (const uint16_t *)c->crctab.p, (unsigned long long *)d_offsets, PL.B.guard, c->stream, nd, &PL.D) != 0) {
                fg_set_error("frame assembly kernel launch failed"); return false;
            }
            mark();
            const unsigned long long *tl = (const unsigned long long *)c->offsets.p + nblocks;
            if (lean) {
                const unsigned long long seq = ++c->sig_seq;
                if (ev2 && hipEventRecord(c->ev[2], c->stream) != hipSuccess) return false;
                if (!rest) {
                    if (fg_launch_signal_direct(tl, PL.B.guard, (const unsigned long long *)c->stamp.p, c->h_sig, seq, c->stream,
                                                !quick_off ? PL.B.guard : nullptr) != 0) return false;
                }
                else if (fg_launch_signal(tl, 2, PL.B.guard, 2, (const unsigned long long *)c->stamp.p, c->h_sig, seq, c->stream,
                                          !quick_off ? PL.B.guard : nullptr) != 0) return false;
                if (!c->wait_signal(seq)) return false;
                c->guard_clean_ptr = !quick_off ? PL.B.guard : nullptr;
                for (int k = 0; k < 4; k++) tail[k] = c->h_sig[2 + k];
                return true;
            }
            tail[0] = tail[1] = 0;
            if (hipMemcpyAsync(tail, tl, rest ? 16 : 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
            if (!rest && hipMemcpyAsync(tail + 1, PL.B.guard + 2, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
            if (hipMemcpyAsync(tail + 2, PL.B.guard, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
            if (hipEventRecord(c->ev[2], c->stream) != hipSuccess) return false;
            return fg_stream_wait(c->stream) == hipSuccess;
        }
        // (frame sizes of the pipeline's blocks: from the chunk bit counts, inside the scan)
        if (fg_launch_scan((FgBlockResult *)c->results.p, piped ? PL.B.chunk_bits : nullptr, nblocks, (unsigned long long *)c->offsets.p,
                           (piped && first && nfast == nblocks && d_out) ? 1 : 0, piped ? PL.B.guard + 2 : nullptr, c->stream) != 0) {
            fg_set_error("scan kernel launch failed"); return false;
        }
        if (first) mark();
        const bool asm_here = d_out && piped;       // the assembly kernel also hands out the frame index and the guard counters
        if (d_out) {
            const int rc = piped ? fg_launch_pipe_assemble((const FgBlockDesc *)c->descs.p, nblocks, (const uint8_t *)c->slots.p, P.slot_bytes,
                                                           chunk_cap_words, nw, PL.B.chunk_bits, (FgBlockResult *)c->results.p,
                                                           (unsigned long long *)c->offsets.p, (uint8_t *)d_out, out_cap,
                                                           (const uint16_t *)c->crctab.p, (unsigned long long *)d_offsets, PL.B.guard, c->stream, 0, nullptr)
                                 : fg_launch_copy((const uint8_t *)c->slots.p, P.slot_bytes, (const FgBlockResult *)c->results.p, nblocks,
                                                  (const unsigned long long *)c->offsets.p, (uint8_t *)d_out, c->stream, out_cap);
            if (rc != 0) { fg_set_error("frame assembly kernel launch failed"); return false; }
        }
        if (first) mark();
        if (d_offsets && !asm_here && hipMemcpyAsync(d_offsets, c->offsets.p, ((size_t)nblocks + 1) * 8, hipMemcpyDefault, c->stream) != hipSuccess) return false;
        if (lean) {
            // totals and error flags (offsets[nblocks .. +1]), guard counters (offsets[nblocks + 2 .. +3] when the assembly kernel
            // parked them there, the guard words themselves otherwise)
            const unsigned long long seq = ++c->sig_seq;
            const unsigned long long *tl = (const unsigned long long *)c->offsets.p + nblocks;
            const bool sep = use_pipe && !asm_here;
            if (ev2 && hipEventRecord(c->ev[2], c->stream) != hipSuccess) return false;
            if (fg_launch_signal(tl, asm_here ? 4 : 2, sep ? PL.B.guard : nullptr, sep ? 2 : 0, (const unsigned long long *)c->stamp.p,
                                 c->h_sig, seq, c->stream, (use_pipe && !quick_off) ? PL.B.guard : nullptr) != 0) return false;
            if (!c->wait_signal(seq)) return false;
            c->guard_clean_ptr = (use_pipe && !quick_off) ? PL.B.guard : nullptr;
            for (int k = 0; k < 4; k++) tail[k] = c->h_sig[2 + k];
            return true;
        }
        tail[0] = tail[1] = 0;
        if (hipMemcpyAsync(tail, (char *)c->offsets.p + (size_t)nblocks * 8, asm_here ? 32 : 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
        if (use_pipe && !asm_here && hipMemcpyAsync(tail + 2, PL.B.guard, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
        if (hipEventRecord(c->ev[2], c->stream) != hipSuccess) return false;
        return fg_stream_wait(c->stream) == hipSuccess;
    };
    if (!finish_pass(true)) { if (!*flacgpu_last_error()) fg_set_error("encode pass failed"); return false; }
    // (what the near-tie guard counted belongs to the analysis: a repeated packing or a redo pass ends through a signal kernel that
    // finds the counters reset)
    const unsigned long long guard_count = tail[2], guard_margin = tail[3];
    st->direct_path = direct ? 1u : 0u;
    const uint32_t first_flags = direct ? ((uint32_t)tail[1] & ~(FG_ERR_REDO | FG_ERR_CHAIN)) : 0u;      // (range errors: the repeated pass does not see them again)
    if (direct && ((uint32_t)tail[1] & (FG_ERR_CHAIN | FG_ERR_REDO))) {
        // a frame the direct path could not place (FG_ERR_CHAIN): pack again in the chunk form -- the decisions of the analysis are
        // all there --, then sizes, scan and assembly as in rounds 2-4
        direct = false;
        st->direct_path = 2;
        set_direct(false);
        PL.stages = 2; PL.ngroups = 1; PL.nblocks = nfast;
        if (fg_launch_encode_pipe(&PL) != 0) { fg_set_error("encode pipeline launch failed"); return false; }
        if (!finish_pass(true)) { if (!*flacgpu_last_error()) fg_set_error("encode pass failed"); return false; }
    }
    if ((uint32_t)tail[1] & FG_ERR_REDO) {
        // blocks the specialised kernels declined (absurd code lengths; wasted bits in round 1's kernel): encode them with the
        // generic kernel, then scan and assemble again
        std::vector<FgBlockResult> r(nblocks);
        HIPCHK(hipMemcpyAsync(r.data(), c->results.p, (size_t)nblocks * sizeof(FgBlockResult), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        std::vector<FgBlockDesc> redo;
        for (uint32_t i = 0; i < nfast; i++) if (r[descs[i].out_slot].err & FG_ERR_REDO) redo.push_back(descs[i]);
        st->redo_blocks = (uint32_t)redo.size();
        if (!redo.empty()) {
            // the redo list goes behind the block list the assembly kernel still needs
            if (!c->descs.ensure(((size_t)nblocks + redo.size()) * sizeof(FgBlockDesc))) return false;
            if (c->dev_descs_ptr != c->descs.p && !upload_descs(descs, true)) return false;      // (the buffer moved)
            PL.descs = (const FgBlockDesc *)c->descs.p;
            FgBlockDesc *d_redo = (FgBlockDesc *)c->descs.p + nblocks;
            HIPCHK(hipMemcpyAsync(d_redo, redo.data(), redo.size() * sizeof(FgBlockDesc), hipMemcpyHostToDevice, c->stream));
            if (fg_launch_encode(d_pcm, d_redo, (const float *)c->windows.p, &P, (uint32_t)redo.size(), (uint8_t *)c->slots.p,
                                 (FgBlockResult *)c->results.p, dbg, (const uint16_t *)c->crctab.p, c->stream) != 0) {
                fg_set_error("encode kernel launch failed"); return false;
            }
            if (!finish_pass(false)) { if (!*flacgpu_last_error()) fg_set_error("encode pass failed"); return false; }
        }
    }
    st->total_bytes = tail[0];
    st->error_flags = ((uint32_t)tail[1] & ~(FG_ERR_REDO | FG_ERR_CHAIN)) | first_flags;
    if (piped) {        // (the autocorrelation kernel resets the counters: they mean something only when the pipeline ran)
        st->log_guard_subframes = (uint32_t)guard_count;
        double mm; memcpy(&mm, &guard_margin, 8);
        st->lpc_order_min_margin = mm;
    }
    c->last_nblocks = nblocks;
    c->last_chunk_bits = piped ? PL.B.chunk_bits : nullptr;
    if (d_out && tail[0] > out_cap) { fg_set_error("output buffer too small"); return false; }
    if (lean && !ev2) st->total_gpu_ms = (float)((double)(c->h_sig[11] - c->h_sig[10]) / c->wall_khz);    // (encode_kernel_ms: levels 1, 2)
    else if (ev2) {
        // (the host saw the signal kernel's word in pinned memory; the runtime may not have looked at the event in front of that
        // kernel yet -- hipEventElapsedTime then answers "not ready", one call in a dozen)
        HIPCHK(hipEventSynchronize(c->ev[2]));
        HIPCHK(hipEventElapsedTime(&st->total_gpu_ms, c->ev[0], c->ev[2]));
    }
    else {
        HIPCHK(hipEventElapsedTime(&st->encode_kernel_ms, c->ev[0], c->ev[1]));
        HIPCHK(hipEventElapsedTime(&st->total_gpu_ms, c->ev[0], c->ev[2]));
    }
    if (timing) {
        // analysis, packing, sizes + scan, assembly
        for (int k = 0; k + 1 < nev; k++) (void)hipEventElapsedTime(&st->stage_ms[k], c->evs[k], c->evs[k + 1]);
    }
    return true;
}

// Streams of three to eight channels.  libFLAC codes their channels independently, so the pipeline's one-channel shape encodes
// every channel as a strided view of the interleaved PCM -- C complete one-channel frames per block, in a scratch stream -- and
// flac_enc_merge.hip splices them into the C-channel frames (one header, the C subframes, one CRC-16).
static bool encode_multichannel(flacgpu_ctx *c, const flacgpu_settings *s, const void *d_pcm, int pcm_is_i16,
                                const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_out, uint64_t out_cap,
                                void *d_offsets, flacgpu_encode_stats *st)
{
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const uint32_t C = s->channels;
    flacgpu_settings s1 = *s;
    s1.channels = 1; s1.do_mid_side = 0; s1.loose_mid_side = 0;
    std::vector<flacgpu_stream_desc> views((size_t)nstreams * C);
    std::vector<uint32_t> map;                       // [frame] first view frame, then [frame] distance between the channels' frames
    uint64_t nframes = 0;
    for (uint32_t i = 0; i < nstreams; i++) nframes += (streams[i].nsamples + s->blocksize - 1) / s->blocksize;
    if (nframes > 0x7FFFFFFFull / C) { fg_set_error("too many frames"); return false; }
    map.resize((size_t)nframes * 2);
    uint64_t fb = 0;
    for (uint32_t i = 0; i < nstreams; i++) {
        for (uint32_t ch = 0; ch < C; ch++) { views[(size_t)i * C + ch] = streams[i]; views[(size_t)i * C + ch].prev_channel_assignment = 0; }
        const uint64_t nb = (streams[i].nsamples + s->blocksize - 1) / s->blocksize;
        for (uint64_t b = 0; b < nb; b++) { map[fb + b] = (uint32_t)(C * fb + b); map[nframes + fb + b] = (uint32_t)nb; }
        fb += nb;
    }
    memset(st, 0, sizeof *st);
    st->lpc_order_min_margin = INFINITY;
    if (nframes == 0) return true;
    uint32_t nviewblocks = 0;
    const uint64_t bound = flacgpu_encode_bound(&s1, views.data(), (uint32_t)views.size(), &nviewblocks);
    if (!c->mc_tmp.ensure((size_t)bound + 64) || !c->mc_offs.ensure(((size_t)nviewblocks + 4) * 8)) return false;
    flacgpu_encode_stats st1;
    if (!encode_streams_impl(c, &s1, d_pcm, pcm_is_i16, views.data(), (uint32_t)views.size(), c->mc_tmp.p, bound, c->mc_offs.p, &st1, C)) return false;
    const uint32_t nf = (uint32_t)nframes;
    if (!c->mc_map.ensure((size_t)nf * 8) || !c->mc_sizes.ensure((size_t)nf * 8) || !c->mc_res.ensure((size_t)nf * sizeof(FgBlockResult)) ||
        !c->mc_foffs.ensure(((size_t)nf + 4) * 8) || !c->ensure_pinned_res(64)) return false;
    HIPCHK(hipEventRecord(c->evs[6], c->stream));
    HIPCHK(hipMemcpyAsync(c->mc_map.p, map.data(), (size_t)nf * 8, hipMemcpyHostToDevice, c->stream));
    if (fg_launch_merge((const FgBlockResult *)c->results.p, c->last_chunk_bits, (const unsigned long long *)c->mc_offs.p, (const uint8_t *)c->mc_tmp.p,
                        (const uint32_t *)c->mc_map.p, (const uint32_t *)c->mc_map.p + nf, C, nf, (uint32_t *)c->mc_sizes.p, (uint32_t *)c->mc_sizes.p + nf,
                        (unsigned long long *)c->mc_foffs.p, (uint8_t *)d_out, out_cap, (FgBlockResult *)c->mc_res.p, (unsigned long long *)d_offsets,
                        c->stream) != 0) {
        fg_set_error("frame merge kernel launch failed"); return false;
    }
    unsigned long long *tail = (unsigned long long *)c->h_res;
    HIPCHK(hipMemcpyAsync(tail, (const char *)c->mc_foffs.p + (size_t)nf * 8, 16, hipMemcpyDeviceToHost, c->stream));
    // (flacgpu_copy_block_results: one record per frame of the caller's streams)
    HIPCHK(hipMemcpyAsync(c->results.p, c->mc_res.p, (size_t)nf * sizeof(FgBlockResult), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipEventRecord(c->evs[7], c->stream));
    if (fg_stream_wait(c->stream) != hipSuccess) { fg_set_error("frame merge failed"); return false; }
    float merge_ms = 0.0f;
    (void)hipEventElapsedTime(&merge_ms, c->evs[6], c->evs[7]);
    *st = st1;
    st->nblocks = nf;
    st->total_bytes = tail[0];
    st->error_flags = ((uint32_t)tail[1] & ~FG_ERR_REDO) | st1.error_flags | (((uint32_t)tail[1] & FG_ERR_REDO) ? FG_ERR_INTERNAL : 0u);
    st->total_gpu_ms = st1.total_gpu_ms + merge_ms;
    st->last_channel_assignment = 0;
    c->last_nblocks = nf;
    c->desc_key.clear();                              // (the results buffer no longer matches the cached block list's slots)
    if (d_out && tail[0] > out_cap) { fg_set_error("output buffer too small"); return false; }
    return true;
}

extern "C" int flacgpu_encode_streams(flacgpu_ctx *c, const flacgpu_settings *s, const void *d_pcm, int pcm_is_i16,
                                      const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_out, uint64_t out_cap,
                                      void *d_offsets, flacgpu_encode_stats *st)
{
    flacgpu_encode_stats local;
    if (!st) st = &local;
    if (!c) { fg_set_error("null context"); return -1; }
    // more than two channels: one-channel views through the pipeline where its shape applies (limit_min_bitrate looks across the
    // channels of a frame, large blocks and wide samples stay with the generic kernel)
    if (s->channels > 2 && s->channels <= 8 && (s->bits_per_sample <= 24 || s->bits_per_sample == 32) && s->max_lpc_order <= 12 && !s->limit_min_bitrate && !c->debug &&
        s->blocksize >= 16 && s->blocksize <= 16384 && d_out && !fg_sel("FLACGPU_NO_FAST") && c->mfma_bad == 0 && !(fg_sel("FLACGPU_MC") && atoi(fg_sel("FLACGPU_MC")) == 0))
        return encode_multichannel(c, s, d_pcm, pcm_is_i16, streams, nstreams, d_out, out_cap, d_offsets, st) ? 0 : -1;
    return encode_streams_impl(c, s, d_pcm, pcm_is_i16, streams, nstreams, d_out, out_cap, d_offsets, st) ? 0 : -1;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, kernel): one process may drive several devices
// (batch.MultiContext: one thread per device), so what has been configured is kept per device, behind a mutex.
#include <map>
// STREAMINFO MD5 of device-resident streams (the batch entry point leaves STREAMINFO's md5sum to the caller: the hash is a serial
// chain per stream, format.h:543).  d_md5: 16 bytes per stream, device memory.  One GPU thread per stream on a stream of its own,
// so a call from another host thread runs beside an encode launch of the same context's data; *gpu_ms gets its duration.
extern "C" int flacgpu_md5_streams(flacgpu_ctx *c, const void *d_pcm, int pcm_is_i16, uint32_t channels, uint32_t bits_per_sample,
                                   const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_md5, float *gpu_ms)
{
    if (!c || !d_pcm || !d_md5 || channels < 1 || channels > 8 || bits_per_sample < 4 || bits_per_sample > 32) { fg_set_error("flacgpu_md5_streams: bad arguments"); return -1; }
    if (nstreams == 0) return 0;
    std::lock_guard<std::mutex> lk(c->md5_mu);
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    std::vector<unsigned long long> jobs(2 * (size_t)nstreams);
    for (uint32_t i = 0; i < nstreams; i++) { jobs[2 * i] = streams[i].pcm_offset; jobs[2 * i + 1] = streams[i].nsamples; }
    if (!c->md5_stream && hipStreamCreateWithFlags(&c->md5_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!c->md5_ev[0]) { if (hipEventCreate(&c->md5_ev[0]) != hipSuccess || hipEventCreate(&c->md5_ev[1]) != hipSuccess) return -1; }
    if (!c->md5_jobs.ensure(jobs.size() * 8)) return -1;
    if (hipMemcpyAsync(c->md5_jobs.p, jobs.data(), jobs.size() * 8, hipMemcpyHostToDevice, c->md5_stream) != hipSuccess) return -1;
    (void)hipEventRecord(c->md5_ev[0], c->md5_stream);
    if (fg_launch_md5_streams(d_pcm, pcm_is_i16 ? 1u : 0u, channels, bits_per_sample, c->md5_jobs.p, nstreams, (uint32_t *)d_md5, c->md5_stream) != 0) {
        fg_set_error("MD5 kernel launch failed"); return -1;
    }
    (void)hipEventRecord(c->md5_ev[1], c->md5_stream);
    if (hipStreamSynchronize(c->md5_stream) != hipSuccess) { fg_set_error("MD5 kernel failed"); return -1; }
    if (gpu_ms) (void)hipEventElapsedTime(gpu_ms, c->md5_ev[0], c->md5_ev[1]);
    return 0;
}

extern "C" int fg_func_set_lds(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(mu);
    size_t &cur = done[std::make_pair(dev, fn)];
    if (bytes > cur) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        cur = bytes;
    }
    return 0;
}
