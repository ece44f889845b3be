/*
 * flac_oracle.c -- CPU restatement of libFLAC 1.4.3's encode/decode hot path.
 * TEST INFRASTRUCTURE ONLY (see flac_oracle.h for the pinning story).
 *
 * Section markers cite SURVEY.md Appendix A/B (the validated spec of the third-party
 * dependency) and the vendored headers under /root/reference/pyflac/include/FLAC/.
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  No fast-math: the LPC
 * analysis is floating-point-order sensitive (hard part 1).
 */
#include "flac_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const char VENDOR[] = "reference libFLAC 1.4.3 20230623"; /* SURVEY A.2 */

/* ------------------------------------------------------------------ bit helpers */
static inline uint32_t ilog2_32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
static inline uint32_t ilog2_64(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }

/* ------------------------------------------------------------------ CRC (format.h:446-450,468-472) */
static uint8_t crc8_tab[256];
static uint16_t crc16_tab[256];
static int crc_ready = 0;
static void crc_init(void)
{
    if (crc_ready) return;
    for (int i = 0; i < 256; i++) {
        uint8_t c = (uint8_t)i;
        for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
        crc8_tab[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++) d = (uint16_t)((d & 0x8000) ? ((d << 1) ^ 0x8005) : (d << 1));
        crc16_tab[i] = d;
    }
    crc_ready = 1;
}
uint8_t flo_crc8(const uint8_t *p, size_t n)
{
    crc_init();
    uint8_t c = 0;
    while (n--) c = crc8_tab[c ^ *p++];
    return c;
}
uint16_t flo_crc16(const uint8_t *p, size_t n)
{
    crc_init();
    uint16_t c = 0;
    while (n--) c = (uint16_t)((c << 8) ^ crc16_tab[(c >> 8) ^ *p++]);
    return c;
}

/* ------------------------------------------------------------------ MD5 */
typedef struct { uint32_t a, b, c, d; uint64_t len; uint8_t buf[64]; uint32_t fill; } md5_t;
static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t MD5_S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
static void md5_block(md5_t *m, const uint8_t *p)
{
    uint32_t w[16];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
               ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = m->a, b = m->b, c = m->c, d = m->d;
    for (int i = 0; i < 64; i++) {
        uint32_t f, g;
        if (i < 16) { f = (b & c) | (~b & d); g = (uint32_t)i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5u * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3u * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7u * i) & 15; }
        uint32_t t = a + f + MD5_K[i] + w[g];
        a = d; d = c; c = b;
        b = b + ((t << MD5_S[i]) | (t >> (32 - MD5_S[i])));
    }
    m->a += a; m->b += b; m->c += c; m->d += d;
}
static void md5_init(md5_t *m)
{
    m->a = 0x67452301; m->b = 0xefcdab89; m->c = 0x98badcfe; m->d = 0x10325476;
    m->len = 0; m->fill = 0;
}
static void md5_update(md5_t *m, const uint8_t *p, size_t n)
{
    m->len += n;
    while (n) {
        size_t k = 64 - m->fill;
        if (k > n) k = n;
        memcpy(m->buf + m->fill, p, k);
        m->fill += (uint32_t)k; p += k; n -= k;
        if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
    }
}
static void md5_final(md5_t *m, uint8_t out[16])
{
    uint64_t bits = m->len * 8;
    uint8_t pad = 0x80;
    md5_update(m, &pad, 1);
    uint8_t z = 0;
    while (m->fill != 56) md5_update(m, &z, 1);
    uint8_t l[8];
    for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
    md5_update(m, l, 8);
    uint32_t v[4] = {m->a, m->b, m->c, m->d};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
}
void flo_md5_pcm(const int32_t *x, uint64_t nsamples, uint32_t channels, uint32_t bps, uint8_t digest[16])
{
    /* little-endian, (bps+7)/8 bytes per sample, interleaved */
    md5_t m;
    md5_init(&m);
    uint32_t bytes = (bps + 7) / 8;
    uint8_t tmp[4096];
    size_t fill = 0;
    uint64_t total = nsamples * channels;
    for (uint64_t i = 0; i < total; i++) {
        uint32_t v = (uint32_t)x[i];
        for (uint32_t b = 0; b < bytes; b++) tmp[fill++] = (uint8_t)(v >> (8 * b));
        if (fill + 4 > sizeof tmp) { md5_update(&m, tmp, fill); fill = 0; }
    }
    md5_update(&m, tmp, fill);
    md5_final(&m, digest);
}

/* ------------------------------------------------------------------ settings */
int flo_config_from_level(flo_config *c, uint32_t level, uint32_t channels, uint32_t bps,
                          uint32_t sample_rate, uint32_t blocksize, uint32_t subset)
{
    /* stream_encoder.h:845-853 */
    static const struct { int ms, loose; int apod; uint32_t order, minpo, maxpo; } L[9] = {
        {0, 0, 0, 0, 0, 3}, {1, 1, 0, 0, 0, 3}, {1, 0, 0, 0, 0, 3}, {0, 0, 0, 6, 0, 4}, {1, 1, 0, 8, 0, 4},
        {1, 0, 0, 8, 0, 5}, {1, 0, 2, 8, 0, 6}, {1, 0, 2, 12, 0, 6}, {1, 0, 3, 12, 0, 6}};
    if (level > 8) level = 8;
    memset(c, 0, sizeof *c);
    c->channels = channels; c->bps = bps; c->sample_rate = sample_rate; c->blocksize = blocksize;
    c->do_mid_side = (uint32_t)L[level].ms; c->loose_mid_side = (uint32_t)L[level].loose;
    c->max_lpc_order = L[level].order;
    c->min_partition_order = L[level].minpo; c->max_partition_order = L[level].maxpo;
    c->apod_type = L[level].apod ? 1 : 0; c->apod_p = 0.5f; c->apod_parts = (uint32_t)L[level].apod;
    c->streamable_subset = subset; c->do_md5 = 1;
    if (channels == 0 || channels > 8) return 4;
    if (channels != 2) { c->do_mid_side = 0; c->loose_mid_side = 0; }
    else if (!c->do_mid_side) c->loose_mid_side = 0;
    if (bps < 4 || bps > 32) return 5;
    if (sample_rate > 1048575u) return 6;
    if (c->blocksize == 0) c->blocksize = c->max_lpc_order == 0 ? 1152 : 4096;
    if (c->blocksize < 16 || c->blocksize > 65535) return 7;
    if (c->max_lpc_order > 32) return 8;
    if (c->blocksize < c->max_lpc_order) return 10;
    {
        uint32_t bs = c->blocksize, q;
        if (bps < 16) { q = 2 + bps / 2; if (q < 5) q = 5; }
        else if (bps == 16) q = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
        else q = bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
        c->qlp_coeff_precision = q;
    }
    if (subset) {
        uint32_t bs = c->blocksize;
        if (bs > 16384 || (sample_rate <= 48000 && bs > 4608)) return 11;
        if (sample_rate >= 65536 && !(sample_rate % 1000 == 0 || sample_rate % 10 == 0)) return 11;
        if (bps != 8 && bps != 12 && bps != 16 && bps != 20 && bps != 24 && bps != 32) return 11;
        if (c->max_partition_order > 8) return 11;
        if (sample_rate <= 48000 && (bs > 4608 || c->max_lpc_order > 12)) return 11;
    }
    if (c->max_partition_order >= 16) c->max_partition_order = 15;
    if (c->min_partition_order >= c->max_partition_order) c->min_partition_order = c->max_partition_order;
    return 0;
}

/* ------------------------------------------------------------------ bit writer */
typedef struct { uint8_t *p; size_t pos; uint64_t acc; uint32_t n; } bw_t;
static inline void bw_flush(bw_t *b)
{
    while (b->n >= 8) { b->p[b->pos++] = (uint8_t)(b->acc >> (b->n - 8)); b->n -= 8; }
}
static inline void bw_bits(bw_t *b, uint32_t v, uint32_t n)
{
    if (n == 0) return;
    if (n < 32) v &= (1u << n) - 1;
    b->acc = (b->acc << n) | v;
    b->n += n;
    bw_flush(b);
}
static inline void bw_bits64(bw_t *b, uint64_t v, uint32_t n)
{
    if (n > 32) { bw_bits(b, (uint32_t)(v >> 32), n - 32); bw_bits(b, (uint32_t)v, 32); }
    else bw_bits(b, (uint32_t)v, n);
}
static inline void bw_zeros(bw_t *b, uint32_t n)
{
    while (n >= 32) { bw_bits(b, 0, 32); n -= 32; }
    bw_bits(b, 0, n);
}
static inline void bw_unary(bw_t *b, uint32_t v) { bw_zeros(b, v); bw_bits(b, 1, 1); }
static inline void bw_rice(bw_t *b, int32_t val, uint32_t k)
{
    uint32_t u = ((uint32_t)val << 1) ^ (uint32_t)(val >> 31);
    bw_zeros(b, u >> k);
    bw_bits(b, (1u << k) | (k ? (u & ((1u << k) - 1)) : 0), k + 1);
}
static void bw_utf8(bw_t *b, uint32_t v)
{
    if (v < 0x80) bw_bits(b, v, 8);
    else if (v < 0x800) { bw_bits(b, 0xC0 | (v >> 6), 8); bw_bits(b, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x10000) { bw_bits(b, 0xE0 | (v >> 12), 8); bw_bits(b, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(b, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x200000) { bw_bits(b, 0xF0 | (v >> 18), 8); bw_bits(b, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(b, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x4000000) { bw_bits(b, 0xF8 | (v >> 24), 8); bw_bits(b, 0x80 | ((v >> 18) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(b, 0x80 | (v & 0x3F), 8); }
    else { bw_bits(b, 0xFC | (v >> 30), 8); bw_bits(b, 0x80 | ((v >> 24) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 18) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(b, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(b, 0x80 | (v & 0x3F), 8); }
}

/* ------------------------------------------------------------------ window */
static void window_tukey(float *w, int32_t L, float p)
{
    for (int32_t n = 0; n < L; n++) w[n] = 1.0f;
    if (p <= 0.0f) return;
    if (p >= 1.0f) { /* hann */
        const int32_t N = L - 1;
        for (int32_t n = 0; n < L; n++) w[n] = (float)(0.5f - 0.5f * cosf(2.0f * (float)M_PI * n / N));
        return;
    }
    const int32_t Np = (int32_t)(p / 2.0f * L) - 1;
    if (Np > 0) {
        for (int32_t n = 0; n <= Np; n++) {
            w[n] = (float)(0.5f - 0.5f * cosf((float)(M_PI * n / Np)));
            w[L - Np - 1 + n] = (float)(0.5f - 0.5f * cosf((float)(M_PI * (n + Np) / Np)));
        }
    }
}
void flo_window(const flo_config *c, uint32_t n, float *w)
{
    float p = c->apod_type == 1 ? c->apod_p / (float)(int32_t)c->apod_parts : c->apod_p;
    window_tukey(w, (int32_t)n, p);
}

/* ------------------------------------------------------------------ LPC analysis */
static void autocorr(const float *d, uint32_t len, uint32_t lag, double *a)
{
    /* strictly ascending-i sequential double sums per lag */
    for (uint32_t l = 0; l < lag; l++) a[l] = 0.0;
    for (uint32_t i = 0; i < len; i++) {
        const double di = d[i];
        uint32_t m = i + 1 < lag ? i + 1 : lag;
        for (uint32_t l = 0; l < m; l++) a[l] += di * (double)d[i - l];
    }
}
static void levinson(const double *autoc, uint32_t *max_order, float lp[][FLO_MAX_LPC_ORDER], double *error)
{
    double r, err, lpc[FLO_MAX_LPC_ORDER];
    uint32_t i, j;
    err = autoc[0];
    for (i = 0; i < *max_order; i++) {
        r = -autoc[i + 1];
        for (j = 0; j < i; j++) r -= lpc[j] * autoc[i - j];
        r /= err;
        lpc[i] = r;
        for (j = 0; j < (i >> 1); j++) {
            double tmp = lpc[j];
            lpc[j] += r * lpc[i - 1 - j];
            lpc[i - 1 - j] += r * tmp;
        }
        if (i & 1) lpc[j] += lpc[j] * r;
        err *= (1.0 - r * r);
        for (j = 0; j <= i; j++) lp[i][j] = (float)(-lpc[j]);
        error[i] = err;
        if (err == 0.0) { *max_order = i + 1; return; }
    }
}
static double ebps_scale(double e, double scale)
{
    if (e > 0.0) {
        double bps = (double)0.5 * log(scale * e) / M_LN2;
        return bps >= 0.0 ? bps : 0.0;
    }
    else if (e < 0.0) return 1e32;
    return 0.0;
}
static uint32_t best_order(const double *err, uint32_t max_order, uint32_t n, uint32_t overhead)
{
    double scale = 0.5 / (double)n, best = (double)(uint32_t)(-1);
    uint32_t bi = 0;
    for (uint32_t i = 0, o = 1; i < max_order; i++, o++) {
        double bits = ebps_scale(err[i], scale) * (double)(n - o) + (double)(o * overhead);
        if (bits < best) { bi = i; best = bits; }
    }
    return bi + 1;
}
static int quantize(const float *lp, uint32_t order, uint32_t precision, int32_t *q, int *shift)
{
    double cmax = 0.0;
    precision--;
    int32_t qmax = 1 << precision, qmin = -qmax;
    qmax--;
    for (uint32_t i = 0; i < order; i++) { double d = fabs((double)lp[i]); if (d > cmax) cmax = d; }
    if (cmax <= 0.0) return 2;
    {
        int log2cmax;
        (void)frexp(cmax, &log2cmax);
        log2cmax--;
        *shift = (int)precision - log2cmax - 1;
        if (*shift > 15) *shift = 15;
        else if (*shift < -16) return 1;
    }
    if (*shift >= 0) {
        double error = 0.0;
        for (uint32_t i = 0; i < order; i++) {
            error += lp[i] * (1 << *shift);
            int32_t v = (int32_t)lround(error);
            if (v > qmax) v = qmax; else if (v < qmin) v = qmin;
            error -= v;
            q[i] = v;
        }
    }
    else {
        const int nshift = -(*shift);
        double error = 0.0;
        for (uint32_t i = 0; i < order; i++) {
            error += lp[i] / (1 << nshift);
            int32_t v = (int32_t)lround(error);
            if (v > qmax) v = qmax; else if (v < qmin) v = qmin;
            error -= v;
            q[i] = v;
        }
        *shift = 0;
    }
    return 0;
}

/* ------------------------------------------------------------------ Rice partition search */
typedef struct {
    int type;                 /* 0 CONSTANT 1 VERBATIM 2 FIXED 3 LPC */
    uint32_t order, precision;
    int shift;
    int32_t qlp[FLO_MAX_LPC_ORDER];
    int rice_method;
    uint32_t porder;
    uint32_t *params;         /* 1<<15 */
    int32_t *residual;        /* blocksize */
} sub_t;

typedef struct {
    uint32_t n;               /* capacity */
    sub_t ws[FLO_MAX_CHANNELS][2];
    int best[FLO_MAX_CHANNELS];
    uint32_t best_bits[FLO_MAX_CHANNELS];
    int64_t *sig[FLO_MAX_CHANNELS];   /* int64: the side channel of a 32-bit stream needs 33 bits (integer_signal_33bit_side) */
    uint32_t wasted[FLO_MAX_CHANNELS], sbps[FLO_MAX_CHANNELS];
    uint64_t *sums;           /* partition sums for all orders: 2<<15 */
    uint32_t *tmp_params;     /* 1<<15 */
    float *window, *windowed;
    uint32_t window_n, window_type, window_parts;
    float window_p;
} enc_ws;

static uint32_t max_po_from_blocksize(uint32_t bs)
{
    uint32_t o = 0;
    while (!(bs & 1)) { o++; bs >>= 1; }
    return o < 15 ? o : 15;
}

static uint32_t count_rice_bits(uint32_t k, uint32_t n, uint64_t sum)
{
    uint64_t v = (uint64_t)4 + (uint64_t)(1 + k) * n + (k ? (sum >> (k - 1)) : (sum << 1)) - (n >> 1);
    return (uint32_t)(v < 0xFFFFFFFFull ? v : 0xFFFFFFFFull);
}

static int set_partitioned_rice(const uint64_t *sums, uint32_t residual_samples, uint32_t pred_order,
                                uint32_t limit, uint32_t po, uint32_t *params, uint32_t *bits)
{
    uint32_t bits_ = 6;
    const uint32_t partitions = 1u << po;
    uint32_t base = (residual_samples + pred_order) >> po;
    uint32_t div_base = 0x40000 / base;
    for (uint32_t p = 0; p < partitions; p++) {
        uint32_t n = base, div;
        if (p > 0) div = div_base;
        else {
            if (n <= pred_order) return 0;
            n -= pred_order;
            div = 0x40000 / n;
        }
        uint64_t mean = sums[p];
        uint32_t k;
        if (mean < 2 || (((mean - 1) * div) >> 18) == 0) k = 0;
        else k = ilog2_64(((mean - 1) * div) >> 18) + 1;
        if (k >= limit) k = limit - 1;
        uint32_t pb = count_rice_bits(k, n, mean);
        params[p] = k;
        if (pb < 0xFFFFFFFFu - bits_) bits_ += pb; else bits_ = 0xFFFFFFFFu;
    }
    *bits = bits_;
    return 1;
}

static uint32_t find_best_partition_order(enc_ws *w, const int32_t *res, uint32_t residual_samples,
                                          uint32_t pred_order, uint32_t limit, uint32_t min_po,
                                          uint32_t max_po, uint32_t bps, sub_t *s)
{
    const uint32_t blocksize = residual_samples + pred_order;
    while (max_po > 0 && (blocksize >> max_po) <= pred_order) max_po--;
    if (min_po > max_po) min_po = max_po;
    /* sums at max order, then pairwise merge downward */
    {
        const uint32_t dps = blocksize >> max_po;
        uint32_t parts = 1u << max_po;
        const uint32_t threshold = 32 - ilog2_32(dps);
        uint32_t rs = 0, end = (uint32_t)(-(int)pred_order);
        if (bps + 4 < threshold) {
            for (uint32_t p = 0; p < parts; p++) {
                uint32_t a = 0;
                end += dps;
                for (; rs < end; rs++) a += (uint32_t)abs(res[rs]);
                w->sums[p] = a;
            }
        }
        else {
            for (uint32_t p = 0; p < parts; p++) {
                uint64_t a = 0;
                end += dps;
                for (; rs < end; rs++) a += (uint64_t)llabs((long long)res[rs]);
                w->sums[p] = a;
            }
        }
        uint32_t from = 0, to = parts;
        for (int po = (int)max_po - 1; po >= (int)min_po; po--) {
            parts >>= 1;
            for (uint32_t i = 0; i < parts; i++) { w->sums[to++] = w->sums[from] + w->sums[from + 1]; from += 2; }
        }
    }
    uint32_t best_bits = 0, best_po = 0, sum = 0;
    for (int po = (int)max_po; po >= (int)min_po; po--) {
        uint32_t bits;
        if (!set_partitioned_rice(w->sums + sum, residual_samples, pred_order, limit, (uint32_t)po, w->tmp_params, &bits))
            break;
        sum += 1u << po;
        if (best_bits == 0 || bits < best_bits) {
            best_bits = bits; best_po = (uint32_t)po;
            memcpy(s->params, w->tmp_params, sizeof(uint32_t) << po);
        }
    }
    s->porder = best_po;
    s->rice_method = 0;
    for (uint32_t p = 0; p < (1u << best_po); p++) if (s->params[p] >= 15) { s->rice_method = 1; break; }
    return best_bits;
}

/* ------------------------------------------------------------------ fixed predictor */
/* The reference binary (x86-64, AVX2 dispatch) computes the error sums of the "wide" and "limit_residual" variants with
 * FLAC__fixed_compute_best_predictor_wide_intrin_avx2 / _limit_residual_intrin_avx2: four lanes walk data_len/4 samples
 * each.  The lanes' histories are taken at j*(data_len/4), but the lanes START at data_len/4, (2*data_len)/4 and
 * (3*data_len)/4; the data_len%4 samples behind the lanes are ignored by the _wide routine and added by a scalar loop in
 * the _limit_residual routine.  When data_len is a multiple of four this
 * is the exact sum; otherwise lanes 2 and 3 start one or two samples late against their history and the sums differ
 * from the plain C loop (which the same library uses on CPUs without AVX2).  Recovered from the disassembly of the
 * reference binary and pinned by tests/tools/fuzz_oracle_vs_ref.py; sums are what pyFLAC users on x86-64 get. */
static void avx2_lane_sums(const int64_t *d, uint32_t len, uint64_t t[5], int over[5])
{
    const uint32_t q = len / 4;
    const uint32_t start[4] = {0, len >> 2, len >> 1, (3 * len) >> 2};
    if (len < 4) return;
    for (int j = 0; j < 4; j++) {
        const int64_t *h = d + (size_t)j * q;
        int64_t p0 = h[-1], p1 = h[-1] - h[-2], p2 = p1 - (h[-2] - h[-3]), p3 = p2 - (h[-2] - 2 * h[-3] + h[-4]);
        for (uint32_t i = 0; i < q; i++) {
            const int64_t e0 = d[start[j] + i], e1 = e0 - p0, e2 = e1 - p1, e3 = e2 - p2, e4 = e3 - p3;
            const int64_t e[5] = {e0, e1, e2, e3, e4};
            for (int k = 0; k < 5; k++) {
                const uint64_t a = (uint64_t)(e[k] < 0 ? -e[k] : e[k]);
                t[k] += a;
                if (a > 0x7FFFFFFF) over[k] = 1;
            }
            p3 = e3; p2 = e2; p1 = e1; p0 = e0;
        }
    }
}

static uint32_t fixed_best_predictor(const int64_t *x, uint32_t n, uint32_t sbps, float rbps[5], uint64_t tot[5])
{
    /* x points at the start of the block */
    uint32_t order;
    if (sbps < 28) {
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const int64_t *d = x + 4;
        const uint32_t len = n - 4;
        /* stream_encoder.c process_subframe_: 32-bit accumulators while sbps + ilog2((blocksize-4)*17) < 32, else _wide */
        if (sbps + ilog2_32(len * 17) < 32) {
            for (int i = 0; i < (int)len; i++) {
                int32_t e0 = (int32_t)d[i], e1 = (int32_t)(d[i] - d[i - 1]), e2 = (int32_t)(d[i] - 2 * d[i - 1] + d[i - 2]);
                int32_t e3 = (int32_t)(d[i] - 3 * d[i - 1] + 3 * d[i - 2] - d[i - 3]);
                int32_t e4 = (int32_t)(d[i] - 4 * d[i - 1] + 6 * d[i - 2] - 4 * d[i - 3] + d[i - 4]);
                t0 += (uint32_t)abs(e0); t1 += (uint32_t)abs(e1); t2 += (uint32_t)abs(e2);
                t3 += (uint32_t)abs(e3); t4 += (uint32_t)abs(e4);
            }
        }
        else {
            uint64_t t[5] = {0, 0, 0, 0, 0};
            int over[5] = {0, 0, 0, 0, 0};
            avx2_lane_sums(d, len, t, over);
            t0 = t[0]; t1 = t[1]; t2 = t[2]; t3 = t[3]; t4 = t[4];
        }
        tot[0] = t0; tot[1] = t1; tot[2] = t2; tot[3] = t3; tot[4] = t4;
#define MIN2(a, b) ((a) < (b) ? (a) : (b))
        if (t0 <= MIN2(MIN2(MIN2(t1, t2), t3), t4)) order = 0;
        else if (t1 <= MIN2(MIN2(t2, t3), t4)) order = 1;
        else if (t2 <= MIN2(t3, t4)) order = 2;
        else if (t3 <= t4) order = 3;
        else order = 4;
        for (int k = 0; k < 5; k++)
            rbps[k] = (float)((tot[k] > 0) ? log(M_LN2 * (double)tot[k] / (double)len) / M_LN2 : 0.0);
    }
    else {
        /* _limit_residual variants: the sums include the four warm-up positions (i = -4..-1), and an order whose
         * residual would not fit int32 is disqualified.  sbps <= 32: the AVX2 lanes above; sbps == 33 (side channel
         * of a 32-bit stream): FLAC__fixed_compute_best_predictor_limit_residual_33bit, plain C, every sample. */
        uint64_t t[5] = {0, 0, 0, 0, 0}, smallest = UINT64_MAX;
        int over[5] = {0, 0, 0, 0, 0};
        const int64_t *d = x + 4;
        const uint32_t len = n - 4;
        for (int i = -4; i < (int)len; i++) {
            uint64_t e[5];
            /* sbps <= 32: the scalar parts of the AVX2 routine are the warm-up positions and the len%4 samples at the
             * end (which the shifted lanes may already have counted once) */
            if (sbps <= 32 && i >= 0 && i < (int)(len & ~3u)) continue;
            e[0] = (uint64_t)llabs((long long)d[i]);
            e[1] = (i > -4) ? (uint64_t)llabs((long long)d[i] - d[i - 1]) : 0;
            e[2] = (i > -3) ? (uint64_t)llabs((long long)d[i] - 2 * (long long)d[i - 1] + d[i - 2]) : 0;
            e[3] = (i > -2) ? (uint64_t)llabs((long long)d[i] - 3 * (long long)d[i - 1] + 3 * (long long)d[i - 2] - d[i - 3]) : 0;
            e[4] = (i > -1) ? (uint64_t)llabs((long long)d[i] - 4 * (long long)d[i - 1] + 6 * (long long)d[i - 2] - 4 * (long long)d[i - 3] + d[i - 4]) : 0;
            for (int k = 0; k < 5; k++) { t[k] += e[k]; if (e[k] > 0x7FFFFFFF) over[k] = 1; }
        }
        if (sbps <= 32) avx2_lane_sums(d, len, t, over);
        /* Observed on the reference binary (true 32-bit probes, tests/test_oracle_vs_reference.py):
         * ties go to the lowest order, a non-zero constant signal is NOT flagged constant and an
         * all-zero one is; i.e. the per-order estimate is derived from total_error_0 for every order. */
        order = 0;
        for (int k = 4; k >= 0; k--) {
            tot[k] = t[k];
            if (!over[k] && t[k] <= smallest) {
                order = (uint32_t)k; smallest = t[k];
                rbps[k] = (float)((t[0] > 0) ? log(M_LN2 * (double)t[0] / (double)len) / M_LN2 : 0.0);
            }
            else rbps[k] = 34.0f;
        }
    }
    return order;
}

static void fixed_residual(const int64_t *x, uint32_t n, uint32_t order, int32_t *r)
{
    /* x points at sample `order`; n residual samples.  The chosen order's residual fits int32 (guarded above for
     * sbps >= 28), so the truncation equals libFLAC's int32 / _wide / _wide_33bit variants. */
    switch (order) {
    case 0: for (int i = 0; i < (int)n; i++) r[i] = (int32_t)x[i]; break;
    case 1: for (int i = 0; i < (int)n; i++) r[i] = (int32_t)(x[i] - x[i - 1]); break;
    case 2: for (int i = 0; i < (int)n; i++) r[i] = (int32_t)(x[i] - 2 * x[i - 1] + x[i - 2]); break;
    case 3: for (int i = 0; i < (int)n; i++) r[i] = (int32_t)(x[i] - 3 * x[i - 1] + 3 * x[i - 2] - x[i - 3]); break;
    default: for (int i = 0; i < (int)n; i++) r[i] = (int32_t)(x[i] - 4 * x[i - 1] + 6 * x[i - 2] - 4 * x[i - 3] + x[i - 4]); break;
    }
}

/* returns 0 if a residual does not fit int32 (the _limit_residual guard) */
static int lpc_residual(const int64_t *x, uint32_t n, const int32_t *q, uint32_t order, int shift, int32_t *r)
{
    for (int i = 0; i < (int)n; i++) {
        int64_t sum = 0;
        for (uint32_t j = 0; j < order; j++) sum += (int64_t)q[j] * x[i - 1 - (int)j];
        int64_t v = x[i] - (sum >> shift);
        if (v <= INT32_MIN || v > INT32_MAX) return 0;
        r[i] = (int32_t)v;
    }
    return 1;
}

/* ------------------------------------------------------------------ per candidate subframe */
/* wide: get_wasted_bits_wide_ (the 33-bit side channel of a 32-bit stream): an all-zero signal reports one wasted bit,
 * which moves it to the 32-bit subframe path. */
static uint32_t get_wasted_bits(int64_t *s, uint32_t n, int wide)
{
    uint32_t i, shift;
    int64_t x = 0;
    for (i = 0; i < n && !(x & 1); i++) x |= s[i];
    if (x == 0) shift = wide ? 1 : 0;
    else for (shift = 0; !(x & 1); shift++) x >>= 1;
    if (shift > 0) for (i = 0; i < n; i++) s[i] >>= shift;
    return shift;
}

/* forbid_constant: limit_min_bitrate asks for at least one bit per sample, so a frame may not consist of CONSTANT
 * subframes only: the last channel of the frame (R after a constant L, S after a constant M) is evaluated with the
 * constant candidate disabled (observed on the reference binary: a silent stereo block becomes CONSTANT + FIXED order 0). */
static void process_subframe(const flo_config *c, enc_ws *w, uint32_t ch, uint32_t n, uint32_t min_po,
                             uint32_t max_po, flo_subframe_info *inf, int forbid_constant)
{
    const int64_t *x = w->sig[ch];
    const uint32_t sbps = w->sbps[ch], wasted = w->wasted[ch];
    const uint32_t limit = c->bps > 16 ? 31 : 15; /* rice parameter limit, SURVEY A.1 */
    int best = 0;
    uint32_t best_bits, cand;
    sub_t *S = w->ws[ch];

    S[best].type = 1;
    {
        uint64_t vb = (uint64_t)8 + wasted + (uint64_t)n * sbps;
        best_bits = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
    }
    if (inf) { inf->wasted = wasted; inf->sbps = sbps; }

    if (n > 4) {
        float rbps[5];
        uint64_t tot[5];
        uint32_t guess = fixed_best_predictor(x, n, sbps, rbps, tot);
        int constant = 0;
        if (inf) { memcpy(inf->fixed_tot, tot, sizeof tot); inf->fixed_guess = guess; }
        if (rbps[1] == 0.0f) {
            constant = 1;
            for (uint32_t i = 1; i < n; i++) if (x[0] != x[i]) { constant = 0; break; }
        }
        if (forbid_constant) constant = 0;
        if (constant) {
            S[!best].type = 0;
            cand = 8 + wasted + sbps;
            if (cand < best_bits) { best = !best; best_bits = cand; }
        }
        else {
            /* fixed */
            uint32_t fo = guess;
            if (fo >= n) fo = n - 1;
            if (!(rbps[fo] >= (float)sbps)) {
                sub_t *s = &S[!best];
                fixed_residual(x + fo, n - fo, fo, s->residual);
                uint32_t rb = find_best_partition_order(w, s->residual, n - fo, fo, limit, min_po, max_po, sbps, s);
                s->type = 2; s->order = fo;
                cand = 8 + wasted + fo * sbps;
                if (rb < 0xFFFFFFFFu - cand) cand += rb; else cand = 0xFFFFFFFFu;
                if (inf) inf->fixed_bits = cand;
                if (cand < best_bits) { best = !best; best_bits = cand; }
            }
            /* lpc */
            if (c->max_lpc_order > 0) {
                uint32_t max_lpc = c->max_lpc_order >= n ? n - 1 : c->max_lpc_order;
                if (max_lpc > 0) {
                    uint32_t a_b = 1, a_c = 0, done = 0, vec = 0;
                    double autoc[FLO_MAX_LPC_ORDER + 1], root[FLO_MAX_LPC_ORDER + 1], lerr[FLO_MAX_LPC_ORDER];
                    float lpl[FLO_MAX_LPC_ORDER][FLO_MAX_LPC_ORDER];
                    memset(autoc, 0, sizeof autoc); memset(root, 0, sizeof root);
                    while (!done) {
                        uint32_t mo = max_lpc;
                        int ok = 1;
                        if (a_b == 1) {
                            for (uint32_t i = 0; i < n; i++) w->windowed[i] = (float)x[i] * w->window[i];
                            autocorr(w->windowed, n, mo + 1, autoc);
                            if (c->apod_type == 1) { memcpy(root, autoc, mo * sizeof(double)); a_b++; }
                            else done = 1;
                        }
                        else {
                            if (n / a_b <= 32) ok = 0;
                            else if (!(a_c % 2)) {
                                const uint32_t part = n / a_b / 2, sh = (a_c / 2 * n) / a_b;
                                if (part + sh < n) {
                                    uint32_t i, j;
                                    for (i = 0; i < part; i++) w->windowed[i] = (float)x[sh + i] * w->window[i];
                                    if (n - part - sh < i) i = n - part - sh;
                                    for (j = n - part; j < n; i++, j++) w->windowed[i] = (float)x[sh + i] * w->window[j];
                                    if (i < n) w->windowed[i] = 0.0f;
                                }
                                autocorr(w->windowed, n / a_b, mo + 1, autoc);
                            }
                            else {
                                for (uint32_t i = 0; i < mo; i++) autoc[i] = root[i] - autoc[i];
                            }
                            /* set_next_subdivide_tukey */
                            if (a_b == 2) { if (a_c == 0) a_c = 2; else { a_c = 0; a_b++; } }
                            else if (a_c < 2 * a_b - 1) a_c++;
                            else { a_c = 0; a_b++; }
                            if (a_b > c->apod_parts) done = 1;
                        }
                        if (inf && vec < FLO_MAX_APOD_VECTORS) {
                            memcpy(inf->autoc[vec], autoc, sizeof autoc);
                            inf->lpc_guess[vec] = 0; inf->lpc_bits[vec] = 0;
                        }
                        const uint32_t vi = vec++;
                        if (!ok) { vec--; continue; }
                        if (autoc[0] == 0.0) continue;
                        levinson(autoc, &mo, lpl, lerr);
                        uint32_t lo = best_order(lerr, mo, n, sbps + c->qlp_coeff_precision);
                        if (inf && vi < FLO_MAX_APOD_VECTORS) inf->lpc_guess[vi] = lo;
                        if (ebps_scale(lerr[lo - 1], 0.5 / (double)(n - lo)) >= (double)sbps) continue;
                        {
                            sub_t *s = &S[!best];
                            uint32_t prec = c->qlp_coeff_precision;
                            int shift;
                            if (sbps <= 17) { uint32_t lim = 32 - sbps - ilog2_32(lo); if (lim < prec) prec = lim; }
                            if (quantize(lpl[lo - 1], lo, prec, s->qlp, &shift) != 0) continue;
                            if (!lpc_residual(x + lo, n - lo, s->qlp, lo, shift, s->residual)) continue;
                            uint32_t rb = find_best_partition_order(w, s->residual, n - lo, lo, limit, min_po, max_po, sbps, s);
                            s->type = 3; s->order = lo; s->precision = prec; s->shift = shift;
                            cand = 8 + wasted + 4 + 5 + lo * (prec + sbps);
                            if (rb < 0xFFFFFFFFu - cand) cand += rb; else cand = 0xFFFFFFFFu;
                            if (inf && vi < FLO_MAX_APOD_VECTORS) inf->lpc_bits[vi] = cand;
                            if (cand > 0 && cand < best_bits) { best = !best; best_bits = cand; }
                        }
                    }
                    if (inf) inf->n_vectors = vec < FLO_MAX_APOD_VECTORS ? vec : FLO_MAX_APOD_VECTORS;
                }
            }
        }
    }
    w->best[ch] = best;
    w->best_bits[ch] = best_bits;
    if (inf) {
        const sub_t *s = &S[best];
        inf->type = (uint32_t)s->type; inf->bits = best_bits;
        inf->order = (s->type >= 2) ? s->order : 0;
        inf->precision = s->type == 3 ? s->precision : 0;
        inf->shift = s->type == 3 ? s->shift : 0;
        memset(inf->qlp, 0, sizeof inf->qlp);
        if (s->type == 3) memcpy(inf->qlp, s->qlp, sizeof(int32_t) * s->order);
        inf->rice_method = s->type >= 2 ? (uint32_t)s->rice_method : 0;
        inf->porder = s->type >= 2 ? s->porder : 0;
        memset(inf->rice_params, 0, sizeof inf->rice_params);
        if (s->type >= 2) {
            uint32_t np = 1u << s->porder;
            if (np > FLO_INFO_MAX_PARTS) np = FLO_INFO_MAX_PARTS;
            memcpy(inf->rice_params, s->params, np * sizeof(uint32_t));
        }
    }
}

/* ------------------------------------------------------------------ bitstream */
static void write_subframe(bw_t *b, const sub_t *s, const int64_t *x, uint32_t n, uint32_t sbps, uint32_t wasted)
{
    uint32_t hdr;
    switch (s->type) {
    case 0: hdr = 0x00; break;
    case 1: hdr = 0x02; break;
    case 2: hdr = 0x10 | (s->order << 1); break;
    default: hdr = 0x40 | ((s->order - 1) << 1); break;
    }
    bw_bits(b, hdr | (wasted ? 1 : 0), 8);
    if (wasted) bw_unary(b, wasted - 1);
    if (s->type == 0) { bw_bits64(b, (uint64_t)x[0], sbps); return; }
    if (s->type == 1) { for (uint32_t i = 0; i < n; i++) bw_bits64(b, (uint64_t)x[i], sbps); return; }
    for (uint32_t i = 0; i < s->order; i++) bw_bits64(b, (uint64_t)x[i], sbps);
    if (s->type == 3) {
        bw_bits(b, s->precision - 1, 4);
        bw_bits(b, (uint32_t)s->shift, 5);
        for (uint32_t i = 0; i < s->order; i++) bw_bits(b, (uint32_t)s->qlp[i], s->precision);
    }
    bw_bits(b, (uint32_t)s->rice_method, 2);
    bw_bits(b, s->porder, 4);
    {
        const uint32_t plen = s->rice_method ? 5 : 4;
        const uint32_t base = n >> s->porder;
        uint32_t k = 0;
        for (uint32_t p = 0; p < (1u << s->porder); p++) {
            uint32_t cnt = base - (p == 0 ? s->order : 0);
            bw_bits(b, s->params[p], plen);
            for (uint32_t i = 0; i < cnt; i++) bw_rice(b, s->residual[k++], s->params[p]);
        }
    }
}

static void write_frame_header(bw_t *b, const flo_config *c, uint32_t n, uint32_t ca, uint32_t frame_number)
{
    size_t start = b->pos;
    uint32_t u, bs_hint = 0, sr_hint = 0;
    bw_bits(b, 0x3FFE, 14); bw_bits(b, 0, 1); bw_bits(b, 0, 1);
    switch (n) {
    case 192: u = 1; break; case 576: u = 2; break; case 1152: u = 3; break; case 2304: u = 4; break;
    case 4608: u = 5; break; case 256: u = 8; break; case 512: u = 9; break; case 1024: u = 10; break;
    case 2048: u = 11; break; case 4096: u = 12; break; case 8192: u = 13; break; case 16384: u = 14; break;
    case 32768: u = 15; break;
    default: bs_hint = u = (n <= 0x100) ? 6 : 7; break;
    }
    bw_bits(b, u, 4);
    switch (c->sample_rate) {
    case 88200: u = 1; break; case 176400: u = 2; break; case 192000: u = 3; break; case 8000: u = 4; break;
    case 16000: u = 5; break; case 22050: u = 6; break; case 24000: u = 7; break; case 32000: u = 8; break;
    case 44100: u = 9; break; case 48000: u = 10; break; case 96000: u = 11; break;
    default:
        if (c->sample_rate <= 255000 && c->sample_rate % 1000 == 0) sr_hint = u = 12;
        else if (c->sample_rate <= 655350 && c->sample_rate % 10 == 0) sr_hint = u = 14;
        else if (c->sample_rate <= 0xffff) sr_hint = u = 13;
        else u = 0;
        break;
    }
    bw_bits(b, u, 4);
    switch (ca) { case 0: u = c->channels - 1; break; case 1: u = 8; break; case 2: u = 9; break; default: u = 10; break; }
    bw_bits(b, u, 4);
    switch (c->bps) { case 8: u = 1; break; case 12: u = 2; break; case 16: u = 4; break; case 20: u = 5; break;
                      case 24: u = 6; break; case 32: u = 7; break; default: u = 0; break; }
    bw_bits(b, u, 3);
    bw_bits(b, 0, 1);
    bw_utf8(b, frame_number);
    if (bs_hint) bw_bits(b, n - 1, bs_hint == 6 ? 8 : 16);
    switch (sr_hint) {
    case 12: bw_bits(b, c->sample_rate / 1000, 8); break;
    case 13: bw_bits(b, c->sample_rate, 16); break;
    case 14: bw_bits(b, c->sample_rate / 10, 16); break;
    default: break;
    }
    bw_bits(b, flo_crc8(b->p + start, b->pos - start), 8);
}

/* ------------------------------------------------------------------ workspace */
static enc_ws *g_ws = NULL;
static enc_ws *get_ws(const flo_config *c, uint32_t n)
{
    enc_ws *w = g_ws;
    if (w && w->n >= n) goto win;
    if (w) {
        for (int ch = 0; ch < FLO_MAX_CHANNELS; ch++) {
            free(w->sig[ch]);
            for (int k = 0; k < 2; k++) { free(w->ws[ch][k].params); free(w->ws[ch][k].residual); }
        }
        free(w->sums); free(w->tmp_params); free(w->window); free(w->windowed); free(w);
    }
    w = (enc_ws *)calloc(1, sizeof *w);
    w->n = n < 4096 ? 4096 : n;
    for (int ch = 0; ch < FLO_MAX_CHANNELS; ch++) {
        w->sig[ch] = (int64_t *)malloc(sizeof(int64_t) * (w->n + 8));
        for (int k = 0; k < 2; k++) {
            w->ws[ch][k].params = (uint32_t *)malloc(sizeof(uint32_t) << 15);
            w->ws[ch][k].residual = (int32_t *)malloc(sizeof(int32_t) * (w->n + 8));
        }
    }
    w->sums = (uint64_t *)malloc(sizeof(uint64_t) * (2u << 15));
    w->tmp_params = (uint32_t *)malloc(sizeof(uint32_t) << 15);
    w->window = (float *)malloc(sizeof(float) * (w->n + 8));
    w->windowed = (float *)malloc(sizeof(float) * (w->n + 8));
    w->window_n = 0;
    g_ws = w;
win:
    if (c->max_lpc_order > 0 && (w->window_n != n || w->window_type != c->apod_type ||
                                 w->window_parts != c->apod_parts || w->window_p != c->apod_p)) {
        flo_window(c, n, w->window);
        w->window_n = n; w->window_type = c->apod_type; w->window_parts = c->apod_parts; w->window_p = c->apod_p;
    }
    return w;
}

/* ------------------------------------------------------------------ one frame */
size_t flo_encode_frame(const flo_config *c, const int32_t *in, uint32_t n, uint32_t frame_number,
                        flo_loose_state *loose, uint8_t *out, flo_frame_info *info)
{
    enc_ws *w = get_ws(c, n);
    const uint32_t C = c->channels;
    int do_indep = 1, do_ms = 0;
    uint32_t ca = 0;
    /* window depends on n only (cached); force recompute when apodization differs between calls */
    if (c->do_mid_side) {
        if (c->loose_mid_side && loose) {
            if (loose->count == 0) { do_indep = 1; do_ms = 1; }
            else { do_indep = (loose->last_ca == 0); do_ms = !do_indep; }
        }
        else { do_indep = 1; do_ms = 1; }
    }
    if (info) { memset(info, 0, sizeof *info); info->blocksize = n; }

    for (uint32_t ch = 0; ch < C; ch++)
        for (uint32_t i = 0; i < n; i++) w->sig[ch][i] = in[(size_t)i * C + ch];
    if (do_ms) {
        for (uint32_t i = 0; i < n; i++) {
            w->sig[3][i] = w->sig[0][i] - w->sig[1][i];
            w->sig[2][i] = (w->sig[0][i] + w->sig[1][i]) >> 1;
        }
    }
    uint32_t max_po = max_po_from_blocksize(n), min_po = c->min_partition_order;
    if (c->max_partition_order < max_po) max_po = c->max_partition_order;
    if (min_po > max_po) min_po = max_po;

    if (do_indep)
        for (uint32_t ch = 0; ch < C; ch++) {
            uint32_t ws = get_wasted_bits(w->sig[ch], n, 0);
            if (ws > c->bps) ws = c->bps;
            w->wasted[ch] = ws; w->sbps[ch] = c->bps - ws;
        }
    if (do_ms)
        for (uint32_t k = 0; k < 2; k++) {
            uint32_t ws = get_wasted_bits(w->sig[2 + k], n, k == 1 && c->bps == 32);
            if (ws > c->bps + 1) ws = c->bps + 1;
            w->wasted[2 + k] = ws; w->sbps[2 + k] = c->bps - ws + k;
        }
    /* limit_min_bitrate (observed on the reference binary, incl. loose mid-side frames): the last independent channel is
     * evaluated with CONSTANT disabled when every earlier one chose CONSTANT; once that happened, mid and side of the same
     * frame are evaluated with CONSTANT disabled as well.  When only mid/side are evaluated (loose mid-side follower
     * frames) nothing is disabled: such frames may still consist of two CONSTANT subframes. */
    int forced = 0;
    if (do_indep) {
        int allc = 1;
        for (uint32_t ch = 0; ch < C; ch++) {
            const int forbid = c->limit_min_bitrate && ch + 1 == C && allc;
            if (forbid) forced = 1;
            process_subframe(c, w, ch, n, min_po, max_po, info ? &info->cand[ch] : NULL, forbid);
            if (w->ws[ch][w->best[ch]].type != 0) allc = 0;
        }
    }
    if (do_ms) {
        for (uint32_t k = 0; k < 2; k++)
            process_subframe(c, w, 2 + k, n, min_po, max_po, info ? &info->cand[2 + k] : NULL, do_indep ? forced : 0);
    }

    uint32_t first = 0, second = 1;
    if (c->do_mid_side) {
        if (c->loose_mid_side && loose && loose->count > 0) ca = loose->last_ca == 0 ? 0 : 3;
        else {
            uint32_t bits[4] = {w->best_bits[0] + w->best_bits[1], w->best_bits[0] + w->best_bits[3],
                                w->best_bits[1] + w->best_bits[3], w->best_bits[2] + w->best_bits[3]};
            uint32_t mn = bits[0];
            for (uint32_t k = c->loose_mid_side ? 3 : 1; k <= 3; k++) if (bits[k] < mn) { mn = bits[k]; ca = k; }
        }
        switch (ca) { case 0: first = 0; second = 1; break; case 1: first = 0; second = 3; break;
                      case 2: first = 3; second = 1; break; default: first = 2; second = 3; break; }
        if (loose) {
            uint32_t period = (uint32_t)((double)c->sample_rate * 0.4 / (double)c->blocksize + 0.5);
            if (period == 0) period = 1;
            loose->count++;
            if (loose->count >= period) loose->count = 0;
            loose->last_ca = ca;
        }
    }

    bw_t b = {out, 0, 0, 0};
    write_frame_header(&b, c, n, ca, frame_number);
    if (c->do_mid_side) {
        write_subframe(&b, &w->ws[first][w->best[first]], w->sig[first], n, w->sbps[first], w->wasted[first]);
        write_subframe(&b, &w->ws[second][w->best[second]], w->sig[second], n, w->sbps[second], w->wasted[second]);
    }
    else
        for (uint32_t ch = 0; ch < C; ch++)
            write_subframe(&b, &w->ws[ch][w->best[ch]], w->sig[ch], n, w->sbps[ch], w->wasted[ch]);
    if (b.n) bw_bits(&b, 0, 8 - b.n);
    {
        uint16_t crc = flo_crc16(out, b.pos);
        bw_bits(&b, crc, 16);
    }
    if (info) {
        info->channel_assignment = ca;
        info->n_candidates = do_ms ? 4 : C;
        info->frame_bytes = (uint32_t)b.pos;
    }
    return b.pos;
}

/* ------------------------------------------------------------------ stream */
size_t flo_stream_header(const flo_config *c, uint32_t minf, uint32_t maxf, uint64_t total,
                         const uint8_t md5[16], uint8_t *out)
{
    bw_t b = {out, 0, 0, 0};
    bw_bits(&b, 0x664C6143u, 32); /* fLaC */
    bw_bits(&b, 0, 1); bw_bits(&b, 0, 7); bw_bits(&b, 34, 24);
    bw_bits(&b, c->blocksize, 16); bw_bits(&b, c->blocksize, 16);
    bw_bits(&b, minf, 24); bw_bits(&b, maxf, 24);
    bw_bits(&b, c->sample_rate, 20); bw_bits(&b, c->channels - 1, 3); bw_bits(&b, c->bps - 1, 5);
    bw_bits64(&b, total, 36);
    for (int i = 0; i < 16; i++) bw_bits(&b, md5 ? md5[i] : 0, 8);
    bw_bits(&b, 1, 1); bw_bits(&b, 4, 7); bw_bits(&b, 8 + (uint32_t)(sizeof VENDOR - 1), 24);
    {
        uint32_t vl = (uint32_t)(sizeof VENDOR - 1);
        for (int i = 0; i < 4; i++) bw_bits(&b, (vl >> (8 * i)) & 0xFF, 8);
        for (uint32_t i = 0; i < vl; i++) bw_bits(&b, (uint8_t)VENDOR[i], 8);
        bw_bits(&b, 0, 32);
    }
    return b.pos;
}

size_t flo_encode_stream(const flo_config *c, const int32_t *in, uint64_t nsamples, int finalize,
                         uint8_t *out, size_t cap, uint32_t *frame_sizes, uint32_t *n_frames)
{
    const uint64_t worst = (uint64_t)c->channels * c->blocksize * ((c->bps + 8) / 8 + 1) + 64;
    size_t pos;
    uint32_t fn = 0, minf = 0, maxf = 0;
    flo_loose_state loose = {0, 0};
    if (cap < 86) return 0;
    pos = flo_stream_header(c, 0, 0, 0, NULL, out);
    for (uint64_t s = 0; s < nsamples; s += c->blocksize, fn++) {
        uint32_t n = (uint32_t)((nsamples - s) < c->blocksize ? (nsamples - s) : c->blocksize);
        if (pos + worst > cap) return 0;
        size_t fb = flo_encode_frame(c, in + s * c->channels, n, fn, &loose, out + pos, NULL);
        if (!fb) return 0;
        if (frame_sizes) frame_sizes[fn] = (uint32_t)fb;
        if (fn == 0 || fb < minf) minf = (uint32_t)fb;
        if (fb > maxf) maxf = (uint32_t)fb;
        pos += fb;
    }
    if (n_frames) *n_frames = fn;
    if (finalize) {
        uint8_t md5[16];
        uint8_t hdr[86];
        memset(md5, 0, sizeof md5);
        if (c->do_md5) flo_md5_pcm(in, nsamples, c->channels, c->bps, md5);
        flo_stream_header(c, minf, maxf, nsamples, md5, hdr);
        memcpy(out, hdr, 42); /* fLaC + STREAMINFO only */
    }
    return pos;
}

/* ================================================================== decoder */
typedef struct { const uint8_t *p; size_t len, pos; uint64_t acc; uint32_t n; int eof; } br_t;
static inline void br_fill(br_t *b)
{
    while (b->n <= 56) {
        if (b->pos < b->len) b->acc = (b->acc << 8) | b->p[b->pos];
        else { b->acc <<= 8; b->eof++; }
        b->pos++; /* virtual position: keeps byte accounting exact past the end */
        b->n += 8;
    }
}
static inline uint32_t br_bits(br_t *b, uint32_t n)
{
    if (n == 0) return 0;
    if (b->n < n) br_fill(b);
    uint32_t v = (uint32_t)((b->acc >> (b->n - n)) & (n == 32 ? 0xFFFFFFFFull : ((1ull << n) - 1)));
    b->n -= n;
    return v;
}
static inline uint64_t br_bits64(br_t *b, uint32_t n)
{
    if (n > 32) { uint64_t hi = br_bits(b, n - 32); return (hi << 32) | br_bits(b, 32); }
    return br_bits(b, n);
}
static inline int64_t br_sbits(br_t *b, uint32_t n)
{
    uint64_t v = br_bits64(b, n);
    if (n < 64 && (v >> (n - 1))) v |= ~0ull << n;
    return (int64_t)v;
}
static inline uint32_t br_unary(br_t *b)
{
    uint32_t z = 0;
    for (;;) {
        if (b->n == 0) br_fill(b);
        uint64_t window = b->acc & ((b->n == 64) ? ~0ull : ((1ull << b->n) - 1));
        if (window) {
            uint32_t lead = (uint32_t)__builtin_clzll(window) - (64 - b->n);
            z += lead; b->n -= lead + 1;
            return z;
        }
        z += b->n; b->n = 0;
        if (b->eof > 16) return z;
    }
}

static int read_residual(br_t *b, uint32_t n, uint32_t order, int32_t *r)
{
    uint32_t method = br_bits(b, 2);
    if (method > 1) return -1;
    uint32_t po = br_bits(b, 4), plen = method ? 5 : 4, esc = method ? 31 : 15;
    uint32_t base = n >> po, k = 0;
    if ((n & ((1u << po) - 1)) != 0 && po > 0) return -1;
    if (base < order && po > 0) return -1;
    if (po == 0 && n < order) return -1;
    for (uint32_t p = 0; p < (1u << po); p++) {
        uint32_t cnt = (po == 0) ? n - order : (p == 0 ? base - order : base);
        uint32_t param = br_bits(b, plen);
        if (param == esc) {
            uint32_t raw = br_bits(b, 5);
            for (uint32_t i = 0; i < cnt; i++) r[k++] = raw ? (int32_t)br_sbits(b, raw) : 0;
        }
        else {
            for (uint32_t i = 0; i < cnt; i++) {
                uint32_t msb = br_unary(b);
                uint32_t u = (msb << param) | br_bits(b, param);
                r[k++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (b->eof > 16) return -1;
    }
    return 0;
}

static int read_subframe(br_t *b, uint32_t n, uint32_t bps, int64_t *out, int32_t *res)
{
    uint32_t hdr = br_bits(b, 8), wasted = 0;
    if (hdr & 0x80) return -1;
    if (hdr & 1) { wasted = br_unary(b) + 1; if (wasted >= bps) return -1; bps -= wasted; }
    hdr = (hdr >> 1) & 0x3F;
    if (hdr == 0) { int64_t v = br_sbits(b, bps); for (uint32_t i = 0; i < n; i++) out[i] = v; }
    else if (hdr == 1) { for (uint32_t i = 0; i < n; i++) out[i] = br_sbits(b, bps); }
    else if (hdr >= 8 && hdr <= 12) {
        uint32_t order = hdr & 7;
        if (order > n) return -1;
        for (uint32_t i = 0; i < order; i++) out[i] = br_sbits(b, bps);
        if (read_residual(b, n, order, res)) return -1;
        for (uint32_t i = order, k = 0; i < n; i++, k++) {
            int64_t r = res[k];
            switch (order) {
            case 0: out[i] = r; break;
            case 1: out[i] = r + out[i - 1]; break;
            case 2: out[i] = r + 2 * out[i - 1] - out[i - 2]; break;
            case 3: out[i] = r + 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
            default: out[i] = r + 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
            }
        }
    }
    else if (hdr >= 32) {
        uint32_t order = (hdr & 31) + 1;
        int32_t q[32];
        if (order > n) return -1;
        for (uint32_t i = 0; i < order; i++) out[i] = br_sbits(b, bps);
        uint32_t prec = br_bits(b, 4) + 1;
        if (prec == 16) return -1;
        int shift = (int)br_sbits(b, 5);
        if (shift < 0) return -1;
        for (uint32_t i = 0; i < order; i++) q[i] = (int32_t)br_sbits(b, prec);
        if (read_residual(b, n, order, res)) return -1;
        for (uint32_t i = order, k = 0; i < n; i++, k++) {
            int64_t sum = 0;
            for (uint32_t j = 0; j < order; j++) sum += (int64_t)q[j] * out[i - 1 - j];
            out[i] = res[k] + (sum >> shift);
        }
    }
    else return -1;
    if (wasted) for (uint32_t i = 0; i < n; i++) out[i] = (int64_t)((uint64_t)out[i] << wasted);
    return 0;
}

int flo_decode_stream(const uint8_t *data, size_t len, int32_t *out, uint64_t cap, flo_decode_result *res,
                      uint32_t *frame_offsets, uint32_t fo_cap)
{
    size_t pos = 0;
    static int64_t *chbuf[FLO_MAX_CHANNELS];
    static int32_t *rbuf;
    static uint32_t bufn = 0;
    memset(res, 0, sizeof *res);
    crc_init();
    /* skip an ID3v2 tag is not handled; require fLaC */
    if (len < 42 || memcmp(data, "fLaC", 4)) return -1;
    pos = 4;
    for (;;) {
        if (pos + 4 > len) return -2;
        uint32_t last = data[pos] >> 7, type = data[pos] & 0x7F;
        uint32_t l = ((uint32_t)data[pos + 1] << 16) | ((uint32_t)data[pos + 2] << 8) | data[pos + 3];
        pos += 4;
        if (pos + l > len) return -2;
        if (type == 0 && l >= 34) {
            const uint8_t *s = data + pos;
            res->min_blocksize = (s[0] << 8) | s[1]; res->max_blocksize = (s[2] << 8) | s[3];
            res->min_framesize = (s[4] << 16) | (s[5] << 8) | s[6];
            res->max_framesize = (s[7] << 16) | (s[8] << 8) | s[9];
            res->sample_rate = ((uint32_t)s[10] << 12) | ((uint32_t)s[11] << 4) | (s[12] >> 4);
            res->channels = ((s[12] >> 1) & 7) + 1;
            res->bps = (((uint32_t)s[12] & 1) << 4 | (s[13] >> 4)) + 1;
            res->total_samples = ((uint64_t)(s[13] & 15) << 32) | ((uint64_t)s[14] << 24) | ((uint64_t)s[15] << 16) | ((uint64_t)s[16] << 8) | s[17];
            memcpy(res->md5, s + 18, 16);
        }
        pos += l;
        if (last) break;
    }
    while (pos + 2 <= len) {
        /* frame sync */
        if (!(data[pos] == 0xFF && (data[pos + 1] & 0xFE) == 0xF8)) {
            if (res->n_errors < 64 && (res->n_errors == 0 || res->errors[res->n_errors - 1] != 0)) res->errors[res->n_errors++] = 0;
            else if (res->n_errors >= 64) res->n_errors++;
            pos++;
            continue;
        }
        br_t b = {data, len, pos, 0, 0, 0};
        size_t start = pos;
        (void)br_bits(&b, 16);
        uint32_t bsc = br_bits(&b, 4), src = br_bits(&b, 4), cac = br_bits(&b, 4), bpc = br_bits(&b, 3);
        uint32_t rsv = br_bits(&b, 1);
        int bad = 0;
        uint32_t n = 0, sr = res->sample_rate, bps = res->bps, channels, ca = 0;
        if (bsc == 0 || src == 15 || cac > 10 || bpc == 3 || rsv) bad = 1;
        /* utf-8 number */
        uint64_t number = 0;
        if (!bad) {
            uint32_t x = br_bits(&b, 8), extra;
            if (!(x & 0x80)) { number = x; extra = 0; }
            else if ((x & 0xE0) == 0xC0) { number = x & 0x1F; extra = 1; }
            else if ((x & 0xF0) == 0xE0) { number = x & 0x0F; extra = 2; }
            else if ((x & 0xF8) == 0xF0) { number = x & 0x07; extra = 3; }
            else if ((x & 0xFC) == 0xF8) { number = x & 0x03; extra = 4; }
            else if ((x & 0xFE) == 0xFC) { number = x & 0x01; extra = 5; }
            else if (x == 0xFE) { number = 0; extra = 6; }
            else { bad = 1; extra = 0; }
            for (uint32_t i = 0; i < extra && !bad; i++) {
                uint32_t y = br_bits(&b, 8);
                if ((y & 0xC0) != 0x80) bad = 1;
                number = (number << 6) | (y & 0x3F);
            }
        }
        if (!bad) {
            switch (bsc) {
            case 1: n = 192; break;
            case 2: case 3: case 4: case 5: n = 576u << (bsc - 2); break;
            case 6: n = br_bits(&b, 8) + 1; break;
            case 7: n = br_bits(&b, 16) + 1; break;
            default: n = 256u << (bsc - 8); break;
            }
            static const uint32_t SR[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
            if (src >= 1 && src <= 11) sr = SR[src];
            else if (src == 12) sr = br_bits(&b, 8) * 1000;
            else if (src == 13) sr = br_bits(&b, 16);
            else if (src == 14) sr = br_bits(&b, 16) * 10;
            static const uint32_t BP[8] = {0, 8, 12, 0, 16, 20, 24, 32};
            if (bpc) bps = BP[bpc];
            size_t hlen = (b.pos - b.n / 8) - start;
            uint8_t crc = (uint8_t)br_bits(&b, 8);
            if (flo_crc8(data + start, hlen) != crc) bad = 1;
        }
        if (bad || b.eof) {
            if (res->n_errors < 64) res->errors[res->n_errors] = 1; /* BAD_HEADER */
            res->n_errors++;
            pos++;
            continue;
        }
        (void)sr; (void)number;
        if (cac < 8) { channels = cac + 1; ca = 0; } else { channels = 2; ca = cac - 7; }
        if (bufn < n) {
            for (int c2 = 0; c2 < FLO_MAX_CHANNELS; c2++) { free(chbuf[c2]); chbuf[c2] = (int64_t *)malloc(sizeof(int64_t) * (n + 8)); }
            free(rbuf); rbuf = (int32_t *)malloc(sizeof(int32_t) * (n + 8));
            bufn = n;
        }
        int err = 0;
        for (uint32_t ch = 0; ch < channels && !err; ch++) {
            uint32_t sb = bps;
            if ((ca == 1 && ch == 1) || (ca == 2 && ch == 0) || (ca == 3 && ch == 1)) sb++;
            err = read_subframe(&b, n, sb, chbuf[ch], rbuf);
        }
        if (err || b.eof > 8) {
            if (res->n_errors < 64) res->errors[res->n_errors] = 0; /* LOST_SYNC */
            res->n_errors++;
            pos++;
            continue;
        }
        b.n -= b.n % 8; /* byte align */
        size_t fend = b.pos - b.n / 8;
        uint16_t crc = (uint16_t)br_bits(&b, 16);
        size_t after = fend + 2;
        int crc_ok = (after <= len) && flo_crc16(data + start, fend - start) == crc;
        if (!crc_ok) {
            if (res->n_errors < 64) res->errors[res->n_errors] = 2; /* FRAME_CRC_MISMATCH */
            res->n_errors++;
        }
        if (frame_offsets && res->n_frames < fo_cap) frame_offsets[res->n_frames] = (uint32_t)start;
        if (out && res->decoded_samples + n <= cap) {
            int32_t *o = out + res->decoded_samples * channels;
            for (uint32_t i = 0; i < n; i++) {
                if (!crc_ok) { for (uint32_t ch = 0; ch < channels; ch++) o[(size_t)i * channels + ch] = 0; continue; }
                if (ca == 0) for (uint32_t ch = 0; ch < channels; ch++) o[(size_t)i * channels + ch] = (int32_t)chbuf[ch][i];
                else if (ca == 1) { o[2 * i] = (int32_t)chbuf[0][i]; o[2 * i + 1] = (int32_t)(chbuf[0][i] - chbuf[1][i]); }
                else if (ca == 2) { o[2 * i] = (int32_t)(chbuf[0][i] + chbuf[1][i]); o[2 * i + 1] = (int32_t)chbuf[1][i]; }
                else {
                    int64_t mid = chbuf[0][i], side = chbuf[1][i];
                    mid = (int64_t)((uint64_t)mid << 1) | (side & 1);
                    o[2 * i] = (int32_t)((mid + side) >> 1);
                    o[2 * i + 1] = (int32_t)((mid - side) >> 1);
                }
            }
        }
        res->decoded_samples += n;
        res->n_frames++;
        pos = after;
    }
    return 0;
}
