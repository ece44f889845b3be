// flac_enc_kernels.hip -- MI355X (gfx950) FLAC frame encoder, one FLAC block per wavefront.
//
// Replaces the per-block hot loop of libFLAC's process_frame_/process_subframes_ that pyFLAC
// reaches through FLAC__stream_encoder_process_interleaved (reference: pyflac/encoder.py:115;
// algorithm: SURVEY.md Appendix A, rows L2-L13 of section 8a).  Output frames are bit-exact with
// libFLAC 1.4.3 for the same settings.
//
// Mapping (wave64, block = one wavefront):
//   * samples are staged once from HBM into LDS (planar int32, two channels at a time);
//   * integer stages (wasted bits, fixed-predictor error sums, FIR residuals, Rice partition sums,
//     bit packing) run lane = sample (i = 64*t + lane), LDS reads are conflict-free;
//   * the order-sensitive double autocorrelation runs lane = (candidate, lag): one serial fp64
//     FMA chain per lag, windowed samples pre-converted to double in LDS chunks;
//   * Levinson-Durbin / order guess / quantiser run lane = (candidate, apodization vector);
//   * the bit packer computes every code's length, wave-scans them into bit offsets, ORs the
//     codes into a small LDS window and streams finished words to the frame's HBM slot;
//   * CRC-16 is computed by 64 lanes over interleaved words and folded with x^(32k) multipliers.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile).  No fast-math: the
// LPC analysis must round exactly like the x86-64 SSE2 double arithmetic of the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fg_dev.h"
#include "fg_types.h"

#define FG_LN2 0.69314718055994530942

typedef unsigned long long u64;
typedef long long i64;

namespace {

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_add(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ u64 wave_add64(u64 v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return __shfl(v, src); }
__device__ __forceinline__ uint32_t ilog2_32(uint32_t v) { return 31u - (uint32_t)__clz(v); }
__device__ __forceinline__ uint32_t ilog2_64(u64 v) { return 63u - (uint32_t)__clzll(v); }
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ void lds_fence() { __syncthreads(); }

// ------------------------------------------------------------------ CRC helpers (poly 0x8005 / 0x07)
__device__ __forceinline__ uint32_t crc16_mulx(uint32_t a, int nbits)
{
    for (int i = 0; i < nbits; i++) a = (a & 0x8000) ? (((a << 1) ^ 0x8005) & 0xFFFF) : ((a << 1) & 0xFFFF);
    return a;
}
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
    // a*b mod x^16+x^15+x^2+1
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000) ? (((r << 1) ^ 0x8005) & 0xFFFF) : ((r << 1) & 0xFFFF);
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}

// ------------------------------------------------------------------ per-wave encoder state
struct Dec {            // decision for one candidate subframe (lives in LDS)
    uint32_t type;      // 0 CONSTANT 1 VERBATIM 2 FIXED 3 LPC
    uint32_t order, prec;
    int32_t shift;
    uint32_t porder, method, bits, wasted, sbps, pad;
    int32_t q[FG_MAX_ORDER];
    uint8_t k[FG_MAX_PARTS];
};

struct Enc {
    // LDS regions
    int32_t *s0, *s1;
    uint32_t sstr;      // element stride of s0/s1: 1 when staged in LDS, channels when read straight from the PCM buffer
    double *dbuf;
    double *autoc;
    int32_t *qres;
    uint32_t *lres;
    u64 *sums;
    uint8_t *tmpk;
    uint32_t *win;
    uint16_t *crct;
    uint32_t *misc;
    Dec *decs;
    // uniform state
    FgEncParams P;
    const float *window;
    int lane;
    uint32_t n;
    int mode;   // 0 independent channels, 1 L/R/M/S
    bool lmb_forced;          // limit_min_bitrate: CONSTANT was disabled for the last independent channel of this frame
    uint32_t lmb_forced_ca;   // the block's forced channel assignment (3: only mid/side count)
    int wide;   // 64-bit arithmetic for M/S derivation
    int ncand;
    uint32_t err;
    // bit writer
    uint32_t bitpos, wbase;
    uint32_t payload_bits = 0;   // the frame's bits in front of its padding
    uint32_t view = 0;           // FgBlockDesc.reserved: one-channel view of a stream of `view` interleaved channels
    uint32_t *outw;
    uint32_t slot_words;

    __device__ __forceinline__ int32_t cval(int c, int32_t L, int32_t R, uint32_t w) const
    {
        if (mode == 0) return (c == 0 ? L : R) >> w;
        if (c == 0) return L >> w;
        if (c == 1) return R >> w;
        if (!wide) return (c == 2 ? ((L + R) >> 1) : (L - R)) >> w;
        i64 a = (c == 2) ? (((i64)L + (i64)R) >> 1) : ((i64)L - (i64)R);
        return (int32_t)(a >> w);
    }
    __device__ __forceinline__ int32_t get(int c, uint32_t i, uint32_t w) const
    {
        int32_t L = 0, R = 0;
        const size_t j = (size_t)i * sstr;
        if (mode == 0) { if (c == 0) L = s0[j]; else R = s1[j]; }
        else {
            if (c != 1) L = s0[j];
            if (c != 0) R = s1[j];
        }
        return cval(c, L, R, w);
    }

    // The side channel of a 32-bit stream has 33 bits (libFLAC's integer_signal_33bit_side).  It is never stored: the exact
    // value is L - R in 64-bit arithmetic wherever it is needed.  is33: the candidate is that channel with no wasted bits
    // (with wasted bits the shifted values fit int32 and take the ordinary path, get_wasted_bits_wide_).
    __device__ __forceinline__ bool is33(int c, uint32_t w) const { return mode == 1 && c == 3 && wide && w == 0; }
    __device__ __forceinline__ i64 get64(int c, uint32_t i, uint32_t w) const
    {
        if (mode == 1 && c == 3 && wide) {
            const size_t j = (size_t)i * sstr;
            return ((i64)s0[j] - (i64)s1[j]) >> w;
        }
        return (i64)get(c, i, w);
    }

    // ---------------------------------------------------------------- staging
    __device__ void stage(const void *pcm, u64 pcm_off, uint32_t ch0, uint32_t nch)
    {
        const uint32_t C = P.channels;
        if (view) {
            // a one-channel view of a stream of `view` interleaved channels (FgBlockDesc.reserved): element offset, element stride
            const i64 lo_ = -((i64)1 << (P.bps - 1)), hi_ = ((i64)1 << (P.bps - 1)) - 1;
            uint32_t bad_ = 0;
            if (P.sig_stride == 0) {
                int32_t *base = (int32_t *)pcm + pcm_off;
                s0 = base; s1 = base; sstr = view;
                for (uint32_t i = lane; i < n; i += 64) { const int32_t v = base[(size_t)i * view]; if ((i64)v < lo_ || (i64)v > hi_) bad_ = 1; }
            }
            else {
                sstr = 1;
                for (uint32_t i = lane; i < n; i += 64) {
                    const u64 idx = pcm_off + (u64)i * view;
                    const int32_t v = P.pcm_i16 ? (int32_t)((const int16_t *)pcm)[idx] : ((const int32_t *)pcm)[idx];
                    if ((i64)v < lo_ || (i64)v > hi_) bad_ = 1;
                    s0[i] = v;
                }
            }
            if (__any(bad_)) err |= FG_ERR_RANGE;
            lds_fence();
            return;
        }
        const i64 lo = -((i64)1 << (P.bps - 1)), hi = ((i64)1 << (P.bps - 1)) - 1;
        uint32_t bad = 0;
        if (P.sig_stride == 0) {
            // block too large for LDS: read the interleaved int32 PCM in place (host guarantees !pcm_i16)
            int32_t *base = (int32_t *)pcm + (pcm_off * C + ch0);
            s0 = base; s1 = base + (nch > 1 ? 1 : 0); sstr = C;
            for (uint32_t i = lane; i < n; i += 64)
                for (uint32_t c = 0; c < nch; c++) { const int32_t v = base[(size_t)i * C + c]; if ((i64)v < lo || (i64)v > hi) bad = 1; }
            if (__any(bad)) err |= FG_ERR_RANGE;
            return;
        }
        sstr = 1;
        for (uint32_t i = lane; i < n; i += 64) {
            for (uint32_t c = 0; c < nch; c++) {
                int32_t v;
                u64 idx = (pcm_off + i) * C + ch0 + c;
                if (P.pcm_i16) v = ((const int16_t *)pcm)[idx];
                else v = ((const int32_t *)pcm)[idx];
                if ((i64)v < lo || (i64)v > hi) bad = 1;
                (c == 0 ? s0 : s1)[i] = v;
            }
        }
        if (__any(bad)) err |= FG_ERR_RANGE;
        lds_fence();
    }

    // ---------------------------------------------------------------- wasted bits
    __device__ void wasted_bits(uint32_t dbase)
    {
        for (int c = 0; c < ncand; c++) {
            uint32_t orl = 0, orh = 0;
            const bool side33 = mode == 1 && c == 3 && wide;
            for (uint32_t i = lane; i < n; i += 64) {
                if (side33) {
                    i64 s = (i64)s0[(size_t)i * sstr] - (i64)s1[(size_t)i * sstr];
                    orl |= (uint32_t)s; orh |= (uint32_t)((u64)s >> 32);
                }
                else orl |= (uint32_t)get(c, i, 0);
            }
            orl = wave_or(orl); orh = wave_or(orh);
            uint32_t w = 0;
            if (orl) w = (uint32_t)__builtin_ctz(orl);
            else if (orh) w = 32;
            else if (side33) w = 1;        // get_wasted_bits_wide_: an all-zero 33-bit signal reports one wasted bit
            uint32_t nominal = P.bps + ((mode == 1 && c == 3) ? 1u : 0u);
            if (w > nominal) w = nominal;
            uint32_t sb = nominal - w;
            if (lane == 0) { decs[dbase + c].wasted = w; decs[dbase + c].sbps = sb; }
        }
        lds_fence();
    }

    // ---------------------------------------------------------------- fixed predictor sums (L4)
    // Error sums the way the reference binary's AVX2 routines take them (oracle/flac_oracle.c avx2_lane_sums): four lanes
    // of len/4 samples whose histories sit at j*(len/4) while the lanes start at (j*len)/4.  Only called when len is not a
    // multiple of four (otherwise this is the plain sum).  d(i) = sample 4+i of the block.
    __device__ void lane_sums(int c, uint32_t w, uint32_t len, u64 a[5], uint32_t *inv)
    {
        const uint32_t q = len >> 2;
        for (uint32_t idx = lane; idx < 4 * q; idx += 64) {
            const uint32_t j = idx / q, i = idx - j * q;
            const uint32_t hb = 4 + j * q, st = 4 + ((j * len) >> 2);
            i64 v[5];
            for (uint32_t m = 0; m < 5; m++) v[m] = i >= m ? get64(c, st + i - m, w) : get64(c, hb + i - m, w);
            const i64 e[5] = {v[0], v[0] - v[1], v[0] - 2 * v[1] + v[2], v[0] - 3 * v[1] + 3 * v[2] - v[3],
                              v[0] - 4 * v[1] + 6 * v[2] - 4 * v[3] + v[4]};
            for (int k = 0; k < 5; k++) {
                const u64 m = (u64)(e[k] < 0 ? -e[k] : e[k]);
                a[k] += m;
                if (m > 0x7FFFFFFFull) *inv |= 1u << k;
            }
        }
    }

    __device__ uint32_t fixed_sums(int c, uint32_t w, uint32_t sb, u64 tot[5], float rb[5])
    {
        u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        uint32_t order;
        const uint32_t len = n - 4;
        if (sb < 28) {
            // 32-bit accumulators (exact) while sb + ilog2((n-4)*17) < 32, else the _wide routine: AVX2 lanes, no remainder
            if ((len & 3) == 0 || sb + ilog2_32(len * 17) < 32) {
                for (uint32_t i = 4 + lane; i < n; i += 64) {
                    int32_t v0 = get(c, i, w), v1 = get(c, i - 1, w), v2 = get(c, i - 2, w), v3 = get(c, i - 3, w),
                            v4 = get(c, i - 4, w);
                    int32_t e1 = v0 - v1, d1 = v1 - v2, d2 = v2 - v3, d3 = v3 - v4;
                    int32_t e2 = e1 - d1, f2 = d1 - d2, g2 = d2 - d3;
                    int32_t e3 = e2 - f2, f3 = f2 - g2;
                    int32_t e4 = e3 - f3;
                    a0 += (uint32_t)abs(v0); a1 += (uint32_t)abs(e1); a2 += (uint32_t)abs(e2);
                    a3 += (uint32_t)abs(e3); a4 += (uint32_t)abs(e4);
                }
            }
            else {
                u64 a[5] = {0, 0, 0, 0, 0};
                uint32_t inv = 0;
                lane_sums(c, w, len, a, &inv);
                a0 = a[0]; a1 = a[1]; a2 = a[2]; a3 = a[3]; a4 = a[4];
            }
            tot[0] = wave_add64(a0); tot[1] = wave_add64(a1); tot[2] = wave_add64(a2);
            tot[3] = wave_add64(a3); tot[4] = wave_add64(a4);
            u64 m34 = tot[3] < tot[4] ? tot[3] : tot[4];
            u64 m234 = tot[2] < m34 ? tot[2] : m34;
            u64 m1234 = tot[1] < m234 ? tot[1] : m234;
            if (tot[0] <= m1234) order = 0;
            else if (tot[1] <= m234) order = 1;
            else if (tot[2] <= m34) order = 2;
            else if (tot[3] <= tot[4]) order = 3;
            else order = 4;
            const double dlen = (double)len;
            for (int k = 0; k < 5; k++)
                rb[k] = (float)((tot[k] > 0) ? log(FG_LN2 * (double)tot[k] / dlen) / FG_LN2 : 0.0);
        }
        else {
            // sbps >= 28: libFLAC's _limit_residual variants (oracle/flac_oracle.c fixed_best_predictor).  33 bit: plain
            // sums over every sample.  <= 32 bit: warm-up positions + AVX2 lanes + the len%4 samples at the end.
            uint32_t inv = 0;
            const bool lanes = sb <= 32 && (len & 3) != 0;
            const uint32_t tail0 = 4 + (len & ~3u);
            for (uint32_t i = lane; i < n; i += 64) {
                if (lanes && i >= 4 && i < tail0) continue;
                i64 v0 = get64(c, i, w);
                i64 v1 = i >= 1 ? get64(c, i - 1, w) : 0, v2 = i >= 2 ? get64(c, i - 2, w) : 0;
                i64 v3 = i >= 3 ? get64(c, i - 3, w) : 0, v4 = i >= 4 ? get64(c, i - 4, w) : 0;
                u64 e0 = (u64)(v0 < 0 ? -v0 : v0), e1 = 0, e2 = 0, e3 = 0, e4 = 0;
                i64 t;
                if (i >= 1) { t = v0 - v1; e1 = (u64)(t < 0 ? -t : t); }
                if (i >= 2) { t = v0 - 2 * v1 + v2; e2 = (u64)(t < 0 ? -t : t); }
                if (i >= 3) { t = v0 - 3 * v1 + 3 * v2 - v3; e3 = (u64)(t < 0 ? -t : t); }
                if (i >= 4) { t = v0 - 4 * v1 + 6 * v2 - 4 * v3 + v4; e4 = (u64)(t < 0 ? -t : t); }
                a0 += e0; a1 += e1; a2 += e2; a3 += e3; a4 += e4;
                if (e0 > 0x7FFFFFFFull) inv |= 1;
                if (e1 > 0x7FFFFFFFull) inv |= 2;
                if (e2 > 0x7FFFFFFFull) inv |= 4;
                if (e3 > 0x7FFFFFFFull) inv |= 8;
                if (e4 > 0x7FFFFFFFull) inv |= 16;
            }
            if (lanes) {
                u64 a[5] = {0, 0, 0, 0, 0};
                lane_sums(c, w, len, a, &inv);
                a0 += a[0]; a1 += a[1]; a2 += a[2]; a3 += a[3]; a4 += a[4];
            }
            inv = wave_or(inv);
            tot[0] = wave_add64(a0); tot[1] = wave_add64(a1); tot[2] = wave_add64(a2);
            tot[3] = wave_add64(a3); tot[4] = wave_add64(a4);
            u64 smallest = ~0ull;
            order = 0;
            const double dlen = (double)len;
            for (int k = 4; k >= 0; k--) {
                if (!((inv >> k) & 1) && tot[k] <= smallest) {
                    order = (uint32_t)k; smallest = tot[k];
                    rb[k] = (float)((tot[0] > 0) ? log(FG_LN2 * (double)tot[0] / dlen) / FG_LN2 : 0.0);
                }
                else rb[k] = 34.0f;
            }
        }
        return order;
    }

    __device__ bool is_constant(int c, uint32_t w)
    {
        const i64 x0 = get64(c, 0, w);
        uint32_t ne = 0;
        for (uint32_t i = lane; i < n; i += 64) ne |= (get64(c, i, w) != x0);
        return !__any(ne);
    }

    // ---------------------------------------------------------------- autocorrelation (L5, L6)
    // One vector for every candidate at once.  vec_len samples; if part == 0 the plain window over the
    // whole block, else FLAC__lpc_window_data_partial(part_size = part, data_shift = sh).
    __device__ void autocorr_vector(uint32_t dbase, uint32_t v, uint32_t vec_len, uint32_t part, uint32_t sh,
                                    uint32_t mo, uint32_t LP)
    {
        const uint32_t mo1 = mo + 1;
        const uint32_t cpp = 64 / LP;                // candidates per pass
        const uint32_t DSTR = FG_DH + FG_DK;
        for (uint32_t c0 = 0; c0 < (uint32_t)ncand; c0 += cpp) {
            const uint32_t cl = lane / LP, l = lane % LP;   // candidate slot / lag of this lane
            const uint32_t c = c0 + cl;
            double acc = 0.0;
            // zero the history
            for (uint32_t j = lane; j < cpp * DSTR; j += 64)
                if ((j % DSTR) < FG_DH) dbuf[j] = 0.0;
            lds_fence();
            for (uint32_t k0 = 0; k0 < vec_len; k0 += FG_DK) {
                const uint32_t kn = (vec_len - k0) < FG_DK ? (vec_len - k0) : FG_DK;
                for (uint32_t cc = 0; cc < cpp && c0 + cc < (uint32_t)ncand; cc++) {
                    const uint32_t w = decs[dbase + c0 + cc].wasted;
                    for (uint32_t j = lane; j < kn; j += 64) {
                        const uint32_t i = k0 + j;
                        float d;
                        // (float)int64 for the 33-bit side channel: FLAC__lpc_window_data_wide
                        const bool w33 = is33((int)(c0 + cc), w);
                        if (part == 0) d = (w33 ? (float)get64(c0 + cc, i, w) : (float)get(c0 + cc, i, w)) * window[i];
                        else if (i < part) d = (w33 ? (float)get64(c0 + cc, sh + i, w) : (float)get(c0 + cc, sh + i, w)) * window[i];
                        else if (i < 2 * part) d = (w33 ? (float)get64(c0 + cc, sh + i, w) : (float)get(c0 + cc, sh + i, w)) * window[n - 2 * part + i];
                        else d = 0.0f;
                        dbuf[cc * DSTR + FG_DH + j] = (double)d;
                    }
                }
                lds_fence();
                if (c < (uint32_t)ncand && l < mo1) {
                    const double *cur = dbuf + cl * DSTR + FG_DH;
                    const double *hist = cur - l;
                    for (uint32_t j = 0; j < kn; j++) acc = __builtin_fma(cur[j], hist[j], acc);
                }
                lds_fence();
                // keep the last FG_DH entries as history for the next chunk
                if (k0 + kn < vec_len) {
                    double t[(64 / 16) * FG_DH / 64 + 1];
                    int cnt = 0;
                    for (uint32_t j = lane; j < cpp * FG_DH; j += 64)
                        t[cnt++] = dbuf[(j / FG_DH) * DSTR + kn + (j % FG_DH)];
                    lds_fence();
                    cnt = 0;
                    for (uint32_t j = lane; j < cpp * FG_DH; j += 64) dbuf[(j / FG_DH) * DSTR + (j % FG_DH)] = t[cnt++];
                    lds_fence();
                }
            }
            if (c < (uint32_t)ncand && l < mo1) autoc[(c * P.nvec + v) * (FG_MAX_ORDER + 1) + l] = acc;
            lds_fence();
        }
    }

    __device__ __forceinline__ double ebps(double e, double scale) const
    {
        if (e > 0.0) {
            double b = 0.5 * log(scale * e) / FG_LN2;
            return b >= 0.0 ? b : 0.0;
        }
        else if (e < 0.0) return 1e32;
        return 0.0;
    }

    // ---------------------------------------------------------------- Levinson / order guess / quantise (L7-L9)
    // lane = (candidate, vector).  Results: lres[idx] = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25,
    // qres[idx*32+j].  Scratch (aliases dbuf): lpcw[mo][LS] doubles + lpf[mo][LS] floats.
    __device__ void lpc_decide(uint32_t dbase, uint32_t nv, uint32_t mo, const uint32_t vflags[])
    {
        const uint32_t nidx = (uint32_t)ncand * P.nvec;
        const uint32_t LS = nidx;
        double *lpcw = dbuf;
        float *lpf = (float *)(dbuf + (size_t)mo * LS);
        const uint32_t idx = lane;
        if (idx < nidx) {
            const uint32_t c = idx / P.nvec, v = idx % P.nvec;
            uint32_t vf = 0;
            for (int cc = 0; cc < ncand; cc++) if ((uint32_t)cc == c) vf = vflags[cc];
            const double *A = autoc + (c * P.nvec + v) * (FG_MAX_ORDER + 1);
            bool on = v < nv && ((vf >> v) & 1);
            if (on && A[0] == 0.0) on = false;
            const uint32_t sb = decs[dbase + c].sbps;
            const double a0 = on ? A[0] : 1.0;
            const uint32_t overhead = sb + P.qlp_precision;
            const double scale = 0.5 / (double)n;
            double err = a0, bestb = 4294967295.0;
            uint32_t besti = 0;
            bool stopped = false;
            for (uint32_t i = 0; i < mo; i++) {
                double r = on ? -A[i + 1] : 0.0;
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + idx] * (on ? A[i - j] : 0.0);
                r /= err;
                lpcw[i * LS + idx] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + idx], t2 = lpcw[(i - 1 - j) * LS + idx];
                    lpcw[j * LS + idx] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + idx] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + idx]; lpcw[j * LS + idx] = t + t * r; }
                err *= (1.0 - r * r);
                if (!stopped) {
                    const uint32_t o = i + 1;
                    const double bits = ebps(err, scale) * (double)(n - o) + (double)(o * overhead);
                    if (bits < bestb) { besti = i; bestb = bits; }
                    if (err == 0.0) stopped = true;
                }
            }
            const uint32_t ostar = besti + 1;
            // second pass: recompute the recursion and capture the coefficients of order ostar
            double err2 = a0, err_at = 0.0;
            for (uint32_t i = 0; i < ostar; i++) {
                double r = on ? -A[i + 1] : 0.0;
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + idx] * (on ? A[i - j] : 0.0);
                r /= err2;
                lpcw[i * LS + idx] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + idx], t2 = lpcw[(i - 1 - j) * LS + idx];
                    lpcw[j * LS + idx] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + idx] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + idx]; lpcw[j * LS + idx] = t + t * r; }
                err2 *= (1.0 - r * r);
            }
            err_at = err2;
            for (uint32_t jj = 0; jj < ostar; jj++) lpf[jj * LS + idx] = (float)(-lpcw[jj * LS + idx]);
            uint32_t result = 0;
            if (on) {
                bool ok = !(ebps(err_at, 0.5 / (double)(n - ostar)) >= (double)sb);
                uint32_t prec = P.qlp_precision;
                if (sb <= 17) { const uint32_t lim = 32 - sb - ilog2_32(ostar); if (lim < prec) prec = lim; }
                int shift = 0;
                if (ok) {
                    // FLAC__lpc_quantize_coefficients
                    const int p1 = (int)prec - 1;
                    const int32_t qmax = (1 << p1) - 1, qmin = -(1 << p1);
                    double cmax = 0.0;
                    for (uint32_t j = 0; j < ostar; j++) { const double d = fabs((double)lpf[j * LS + idx]); if (d > cmax) cmax = d; }
                    if (cmax <= 0.0) ok = false;
                    else {
                        const int e = (int)((__double_as_longlong(cmax) >> 52) & 0x7FF) - 1022;  // frexp exponent
                        shift = p1 - (e - 1) - 1;
                        if (shift > 15) shift = 15;
                        else if (shift < -16) ok = false;
                    }
                    if (ok) {
                        double error = 0.0;
                        const bool neg = shift < 0;
                        const double mul = neg ? (double)(1 << (-shift)) : (double)(1 << shift);
                        for (uint32_t j = 0; j < ostar; j++) {
                            const double lpv = (double)lpf[j * LS + idx];
                            error += neg ? lpv / mul : lpv * mul;
                            const double rq = round(error);
                            int32_t qv = (int32_t)(i64)rq;
                            if (qv > qmax) qv = qmax; else if (qv < qmin) qv = qmin;
                            error -= (double)qv;
                            qres[idx * FG_MAX_ORDER + j] = qv;
                        }
                        if (neg) shift = 0;
                    }
                }
                result = ostar | (prec << 8) | (((uint32_t)shift & 0xFF) << 16) | ((ok ? 1u : 0u) << 24) | (1u << 25);
            }
            lres[idx] = result;
        }
        lds_fence();
    }

    // ---------------------------------------------------------------- residual of one sample
    // kind 0: fixed predictor of order `order`; kind 1: LPC with coefficients q (LDS), shift.
    // Returns false through *ovf when the value does not fit int32 (libFLAC's _limit_residual guard).
    __device__ __forceinline__ int32_t residual_at(int c, uint32_t w, uint32_t i, int kind, uint32_t order,
                                                   const int32_t *q, int shift, bool narrow, uint32_t *ovf) const
    {
        if (is33(c, w)) {
            // 33-bit side channel: 64-bit arithmetic (_wide_33bit / _limit_residual_33bit); a chosen predictor's residual fits
            // int32 (fixed: guarded by the order guess; LPC: guarded here)
            i64 r;
            if (kind == 0) {
                const i64 v0 = get64(c, i, w);
                if (order == 0) r = v0;
                else if (order == 1) r = v0 - get64(c, i - 1, w);
                else if (order == 2) r = v0 - 2 * get64(c, i - 1, w) + get64(c, i - 2, w);
                else if (order == 3) r = v0 - 3 * get64(c, i - 1, w) + 3 * get64(c, i - 2, w) - get64(c, i - 3, w);
                else r = v0 - 4 * get64(c, i - 1, w) + 6 * get64(c, i - 2, w) - 4 * get64(c, i - 3, w) + get64(c, i - 4, w);
                return (int32_t)r;
            }
            i64 sum = 0;
            for (uint32_t j = 0; j < order; j++) sum += (i64)q[j] * get64(c, i - 1 - j, w);
            r = get64(c, i, w) - (sum >> shift);
            if (r <= (i64)INT32_MIN || r > (i64)INT32_MAX) *ovf = 1;
            return (int32_t)r;
        }
        if (kind == 0) {
            int32_t v0 = get(c, i, w);
            if (order == 0) return v0;
            int32_t v1 = get(c, i - 1, w);
            if (order == 1) return v0 - v1;
            int32_t v2 = get(c, i - 2, w);
            if (order == 2) return v0 - 2 * v1 + v2;
            int32_t v3 = get(c, i - 3, w);
            if (order == 3) return v0 - 3 * v1 + 3 * v2 - v3;
            int32_t v4 = get(c, i - 4, w);
            return v0 - 4 * v1 + 6 * v2 - 4 * v3 + v4;
        }
        if (narrow) {
            int32_t sum = 0;
            for (uint32_t j = 0; j < order; j++) sum += q[j] * get(c, i - 1 - j, w);
            return get(c, i, w) - (sum >> shift);
        }
        i64 sum = 0;
        for (uint32_t j = 0; j < order; j++) sum += (i64)q[j] * (i64)get(c, i - 1 - j, w);
        i64 r = (i64)get(c, i, w) - (sum >> shift);
        if (r <= (i64)INT32_MIN || r > (i64)INT32_MAX) *ovf = 1;
        return (int32_t)r;
    }

    __device__ bool narrow_ok(uint32_t sb, uint32_t order, const int32_t *q) const
    {
        // FLAC__lpc_max_prediction_before_shift_bps(...) <= 32: the 32-bit sum cannot overflow
        uint32_t s = 0;
        for (uint32_t j = 0; j < order; j++) s += (uint32_t)abs(q[j]);
        if (s == 0) s = 1;
        return sb + (ilog2_32(s) + 2) <= 32;
    }

    // ---------------------------------------------------------------- partition sums at order pmax (L11)
    __device__ bool partition_sums(int c, uint32_t w, uint32_t sb, int kind, uint32_t order, const int32_t *q,
                                   int shift, uint32_t pmax)
    {
        const uint32_t parts = 1u << pmax, psz = n >> pmax;
        const bool narrow = kind == 1 ? narrow_ok(sb, order, q) : true;
        const bool wrap32 = (sb + 4) < (32 - ilog2_32(psz));
        uint32_t ovf = 0;
        if ((psz & 63) == 0) {
            const uint32_t ipp = psz >> 6;      // iterations per partition
            uint32_t t = 0;
            for (uint32_t p = 0; p < parts; p++) {
                u64 a = 0;
                for (uint32_t k = 0; k < ipp; k++, t++) {
                    const uint32_t i = (t << 6) + lane;
                    if (i >= order) {
                        int32_t r = residual_at(c, w, i, kind, order, q, shift, narrow, &ovf);
                        a += (uint32_t)abs(r);
                    }
                }
                a = wave_add64(a);
                if (wrap32) a &= 0xFFFFFFFFull;
                if (lane == 0) sums[p] = a;
            }
        }
        else {
            for (uint32_t p = lane; p < parts; p += 64) sums[p] = 0;
            lds_fence();
            const uint32_t inv = psz > 1 ? (uint32_t)(0xFFFFFFFFull / psz) + 1 : 0;   // exact for i, psz < 2^16
            for (uint32_t i = lane; i < n; i += 64) {
                if (i >= order) {
                    int32_t r = residual_at(c, w, i, kind, order, q, shift, narrow, &ovf);
                    uint32_t p = psz > 1 ? (uint32_t)(((u64)i * inv) >> 32) : i;
                    atomicAdd((unsigned long long *)&sums[p], (unsigned long long)(uint32_t)abs(r));
                }
            }
            lds_fence();
            if (wrap32) for (uint32_t p = lane; p < parts; p += 64) sums[p] &= 0xFFFFFFFFull;
        }
        lds_fence();
        return !__any(ovf);
    }

    // ---------------------------------------------------------------- Rice parameter / partition order search (L11)
    // Leaves the parameters of the best order in tmpk[]; returns estimated bits.
    __device__ uint32_t rice_search(uint32_t order, uint32_t pmax, uint32_t pmin, uint32_t *best_po)
    {
        // lower levels by pairwise merge
        uint32_t from = 0, to = 1u << pmax;
        for (int po = (int)pmax - 1; po >= (int)pmin; po--) {
            const uint32_t parts = 1u << po;
            for (uint32_t p = lane; p < parts; p += 64) sums[to + p] = sums[from + 2 * p] + sums[from + 2 * p + 1];
            from = to; to += parts;
            lds_fence();
        }
        uint32_t best_bits = 0, bpo = 0, base = 0;
        const uint32_t limit = P.rice_limit;
        for (int po = (int)pmax; po >= (int)pmin; po--) {
            const uint32_t parts = 1u << po;
            const uint32_t pbase = n >> po;
            const uint32_t divb = 0x40000u / pbase;
            u64 total = 0;
            uint32_t kk[FG_MAX_PARTS / 64];
            int cnt = 0;
            for (uint32_t p = lane; p < ((parts + 63) & ~63u); p += 64, cnt++) {
                uint32_t k = 0;
                if (p < parts) {
                    uint32_t np = pbase, div = divb;
                    if (p == 0) { np -= order; div = 0x40000u / np; }
                    const u64 mean = sums[base + p];
                    if (mean >= 2) {
                        const u64 qv = ((mean - 1) * div) >> 18;
                        if (qv != 0) k = ilog2_64(qv) + 1;
                    }
                    if (k >= limit) k = limit - 1;
                    u64 pb = (u64)4 + (u64)(1 + k) * np + (k ? (mean >> (k - 1)) : (mean << 1)) - (np >> 1);
                    if (pb > 0xFFFFFFFFull) pb = 0xFFFFFFFFull;
                    total += pb;
                }
                kk[cnt] = k;
            }
            total = wave_add64(total) + 6;
            const uint32_t bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
            if (best_bits == 0 || bits < best_bits) {
                best_bits = bits; bpo = (uint32_t)po;
                cnt = 0;
                for (uint32_t p = lane; p < parts; p += 64, cnt++) tmpk[p] = (uint8_t)kk[cnt];
            }
            base += parts;
            lds_fence();
        }
        *best_po = bpo;
        return best_bits;
    }

    __device__ uint32_t limit_pmax(uint32_t pmax, uint32_t order) const
    {
        while (pmax > 0 && (n >> pmax) <= order) pmax--;
        return pmax;
    }

    // Evaluate one predictor for candidate c; on a strict win, record it in decs[di].
    __device__ void evaluate(int c, uint32_t di, int kind, uint32_t order, uint32_t prec, int shift,
                             const int32_t *q, uint32_t pmax0, uint32_t pmin0, uint32_t *best_bits,
                             uint32_t *out_bits)
    {
        const uint32_t w = decs[di].wasted, sb = decs[di].sbps;
        uint32_t pmax = limit_pmax(pmax0, order);
        uint32_t pmin = pmin0 < pmax ? pmin0 : pmax;
        *out_bits = 0;
        if (!partition_sums(c, w, sb, kind, order, q, shift, pmax)) return;
        uint32_t bpo;
        uint32_t rb = rice_search(order, pmax, pmin, &bpo);
        uint32_t est = kind == 0 ? (8 + w + order * sb) : (8 + w + 4 + 5 + order * (prec + sb));
        if (rb < 0xFFFFFFFFu - est) est += rb; else est = 0xFFFFFFFFu;
        *out_bits = est;
        if (est > 0 && est < *best_bits) {
            *best_bits = est;
            Dec *d = &decs[di];
            if (lane == 0) {
                d->type = kind == 0 ? 2 : 3; d->order = order; d->prec = prec; d->shift = shift;
                d->porder = bpo; d->bits = est;
            }
            if (kind == 1 && (uint32_t)lane < order) d->q[lane] = q[lane];
            uint32_t anyhi = 0;
            for (uint32_t p = lane; p < (1u << bpo); p += 64) { d->k[p] = tmpk[p]; anyhi |= tmpk[p] >= 15; }
            anyhi = __any(anyhi);
            if (lane == 0) d->method = anyhi ? 1 : 0;
        }
        lds_fence();
    }

    // ---------------------------------------------------------------- analysis of the staged group (A.5)
    __device__ __forceinline__ void stamp(FgDebugRec *dbg, int k) const
    {
        if (dbg && lane == 0) dbg->t[k] = clock64();
    }

    __device__ void analyse(uint32_t dbase, FgDebugRec *dbg)
    {
        stamp(dbg, 1);
        const uint32_t pmax_blk = [&] { uint32_t o = 0, b = n; while (!(b & 1)) { o++; b >>= 1; } return o < 15 ? o : 15; }();
        uint32_t pmax0 = pmax_blk < P.max_po ? pmax_blk : P.max_po;
        uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;
        wasted_bits(dbase);
        uint32_t best[FG_MAX_CAND], guess[FG_MAX_CAND], fixed_ok[FG_MAX_CAND], is_const[FG_MAX_CAND];
        uint32_t vflags[FG_MAX_CAND];
        bool any_lpc = false;
        for (int c = 0; c < ncand; c++) {
            const uint32_t w = decs[dbase + c].wasted, sb = decs[dbase + c].sbps;
            u64 vb = (u64)8 + w + (u64)n * sb;
            best[c] = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
            if (lane == 0) { decs[dbase + c].type = 1; decs[dbase + c].bits = best[c]; decs[dbase + c].order = 0; }
            guess[c] = 0; fixed_ok[c] = 0; is_const[c] = 0; vflags[c] = 0;
            if (n > 4) {
                u64 tot[5];
                float rb[5];
                guess[c] = fixed_sums(c, w, sb, tot, rb);
                if (dbg && lane == 0) {
                    for (int k = 0; k < 5; k++) dbg->cand[c].fixed_tot[k] = tot[k];
                    dbg->cand[c].fixed_guess = guess[c];
                }
                // limit_min_bitrate: see flac_enc_fast_impl.h / oracle process_subframe(forbid_constant)
                bool forbid = false;
                if (P.limit_min_bitrate) {
                    if (mode == 1) {
                        if (c == 1 && lmb_forced_ca != 3) { forbid = decs[dbase + 0].type == 0; lmb_forced = forbid; }
                        else if (c >= 2) forbid = lmb_forced;
                    }
                    else if (dbase + (uint32_t)c + 1 == P.channels) {       // decs[] is indexed by the channel here
                        forbid = true;
                        for (uint32_t cc = 0; cc < dbase + (uint32_t)c; cc++) if (decs[cc].type != 0) forbid = false;
                    }
                }
                if (!forbid && rb[1] == 0.0f && is_constant(c, w)) {
                    is_const[c] = 1;
                    uint32_t cb = 8 + w + sb;
                    if (cb < best[c]) { best[c] = cb; if (lane == 0) { decs[dbase + c].type = 0; decs[dbase + c].bits = cb; } }
                }
                else {
                    uint32_t fo = guess[c];
                    if (fo >= n) fo = n - 1;
                    guess[c] = fo;
                    fixed_ok[c] = !(rb[fo] >= (float)sb);
                    if (P.max_lpc_order > 0) { vflags[c] = 0xFFFFu; any_lpc = true; }
                }
            }
        }
        lds_fence();
        stamp(dbg, 2);
        // fixed predictor evaluation (comes before LPC in libFLAC's candidate order)
        for (int c = 0; c < ncand; c++) {
            if (!fixed_ok[c]) continue;
            uint32_t ob;
            evaluate(c, dbase + c, 0, guess[c], 0, 0, nullptr, pmax0, pmin0, &best[c], &ob);
            if (dbg && lane == 0) dbg->cand[c].fixed_bits = ob;
        }
        // LPC
        stamp(dbg, 3);
        uint32_t nv = 0;
        if (any_lpc) {
            const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
            if (mo > 0) {
                const uint32_t LP = (mo + 1) <= 16 ? 16 : ((mo + 1) <= 32 ? 32 : 64);
                autocorr_vector(dbase, 0, n, 0, 0, mo, LP);
                nv = 1;
                if (P.apod_parts >= 2) {
                    for (uint32_t b = 2; b <= P.apod_parts; b++) {
                        const uint32_t cmax = (b == 2) ? 2 : 2 * b - 1;
                        for (uint32_t cc = 0; cc <= cmax; cc += (b == 2 ? 2 : 1)) {
                            if (n / b <= 32) continue;      // no vector at all for this depth
                            if (!(cc & 1)) autocorr_vector(dbase, nv, n / b, n / b / 2, (cc / 2 * n) / b, mo, LP);
                            else {
                                // punch-out: root - previous partial for lags < mo; lag mo keeps the partial (upstream quirk)
                                const uint32_t total = (uint32_t)ncand * (mo + 1);
                                for (uint32_t j = lane; j < total; j += 64) {
                                    const uint32_t c = j / (mo + 1), l = j % (mo + 1);
                                    double *base = autoc + c * P.nvec * (FG_MAX_ORDER + 1);
                                    const double prev = base[(nv - 1) * (FG_MAX_ORDER + 1) + l];
                                    base[nv * (FG_MAX_ORDER + 1) + l] = (l < mo) ? base[l] - prev : prev;
                                }
                                lds_fence();
                            }
                            nv++;
                        }
                    }
                }
                if (dbg) {
                    for (uint32_t j = lane; j < (uint32_t)ncand * nv * (mo + 1); j += 64) {
                        const uint32_t c = j / (nv * (mo + 1)), r = j % (nv * (mo + 1)), v = r / (mo + 1), l = r % (mo + 1);
                        dbg->cand[c].autoc[v][l] = autoc[(c * P.nvec + v) * (FG_MAX_ORDER + 1) + l];
                    }
                    if (lane < ncand) dbg->cand[lane].nvec = nv;
                }
                stamp(dbg, 4);
                lpc_decide(dbase, nv, mo, vflags);
                stamp(dbg, 5);
                for (int c = 0; c < ncand; c++) {
                    if (!vflags[c]) continue;
                    for (uint32_t v = 0; v < nv; v++) {
                        const uint32_t idx = (uint32_t)c * P.nvec + v;
                        const uint32_t r = lres[idx];
                        const uint32_t o = r & 0xFF, prec = (r >> 8) & 0xFF;
                        const int shift = (int)(int8_t)((r >> 16) & 0xFF);
                        uint32_t ob = 0;
                        if (dbg && lane == 0) dbg->cand[c].lpc_guess[v] = (r >> 25) & 1 ? o : 0;
                        if ((r >> 24) & 1) evaluate(c, dbase + c, 1, o, prec, shift, qres + idx * FG_MAX_ORDER, pmax0, pmin0, &best[c], &ob);
                        if (dbg && lane == 0) dbg->cand[c].lpc_bits[v] = ob;
                    }
                }
            }
        }
        lds_fence();
        stamp(dbg, 6);
    }

    // ---------------------------------------------------------------- bit writer
    __device__ void bw_init(uint32_t *out, uint32_t words)
    {
        outw = out; slot_words = words; bitpos = 0; wbase = 0;
        for (uint32_t j = lane; j < FG_WINW + 2; j += 64) win[j] = 0;
        lds_fence();
    }
    // Flush every complete word below bit position `newpos`; window keeps the partial word.
    __device__ void bw_flush(uint32_t newpos)
    {
        uint32_t nfull = (newpos >> 5) - wbase;
        if (nfull == 0) return;
        if (wbase + nfull > slot_words) { err |= FG_ERR_SLOT; nfull = slot_words > wbase ? slot_words - wbase : 0; }
        for (uint32_t j = lane; j < nfull; j += 64) outw[wbase + j] = bswap32(win[j]);
        const uint32_t carry = ((newpos >> 5) - wbase) < FG_WINW + 2 ? win[(newpos >> 5) - wbase] : 0;
        lds_fence();
        for (uint32_t j = lane; j < FG_WINW + 2; j += 64) win[j] = 0;
        lds_fence();
        if (lane == 0) win[0] = carry;
        wbase = newpos >> 5;
        lds_fence();
    }
    __device__ __forceinline__ void bw_or(uint32_t pos, uint32_t val, uint32_t vbits)
    {
        // place the low `vbits` bits of val at absolute bit position pos (MSB first)
        const uint32_t rel = pos - (wbase << 5);
        const uint32_t word = rel >> 5, sh = rel & 31;
        const u64 x = (u64)val << (64 - sh - vbits);
        atomicOr(&win[word], (uint32_t)(x >> 32));
        const uint32_t lo = (uint32_t)x;
        if (lo) atomicOr(&win[word + 1], lo);
    }
    // Skip z zero bits (uniform).
    __device__ void bw_zeros(uint32_t z)
    {
        while (((bitpos + z) >> 5) - wbase >= FG_WINW) {
            const uint32_t np = (wbase + FG_WINW) << 5;
            z -= np - bitpos;
            bitpos = np;
            bw_flush(np);
        }
        bitpos += z;
    }
    // One packing round.  Every lane may contribute a prefix field (pv, pb bits, pb <= 32) followed by a
    // code of nb bits whose low vb bits are val and whose leading nb - vb bits are zero (vb <= 32).
    __device__ void bw_round(uint32_t pv, uint32_t pb, uint32_t val, uint32_t vb, uint32_t nb)
    {
        const uint32_t mine = pb + nb;
        const uint32_t incl = wave_incl_scan(mine, lane);
        const uint32_t total = bcast(incl, 63);
        if (total == 0) return;
        // 32-bit bit positions: a frame is far below 2^32 bits; guard the scan against wrap anyway
        const uint32_t anybig = __any(nb > (1u << 26));
        if (!anybig && (bitpos & 31) + total <= 32u * FG_WINW) {
            const uint32_t off = bitpos + incl - mine;
            if (pb) bw_or(off, pv, pb);
            if (vb) bw_or(off + pb + nb - vb, val, vb);
            lds_fence();
            bitpos += total;
            bw_flush(bitpos);
        }
        else {
            // rare: a round that does not fit the window (very long unary runs); serialise the lanes
            for (int L = 0; L < 64; L++) {
                const uint32_t lpv = bcast(pv, L), lpb = bcast(pb, L), lval = bcast(val, L), lvb = bcast(vb, L),
                               lnb = bcast(nb, L);
                if (lpb) {
                    if (lane == 0) bw_or(bitpos, lpv, lpb);
                    lds_fence();
                    bitpos += lpb;
                    bw_flush(bitpos);
                }
                if (lnb) {
                    bw_zeros(lnb - lvb);
                    bw_flush(bitpos);
                    if (lvb) {
                        if (lane == 0) bw_or(bitpos, lval, lvb);
                        lds_fence();
                        bitpos += lvb;
                        bw_flush(bitpos);
                    }
                }
            }
        }
    }
    __device__ void bw_put(uint32_t val, uint32_t bits)   // uniform single field, bits <= 32
    {
        bw_round(0, 0, lane == 0 ? (bits < 32 ? (val & ((1u << bits) - 1)) : val) : 0, lane == 0 ? bits : 0,
                 lane == 0 ? bits : 0);
    }
    // Flush everything including the partial last word (zero padded); the partial word stays in the
    // window so later fields can still be ORed in (it is simply stored again, as a superset).
    __device__ void bw_flush_all()
    {
        bw_flush(bitpos);
        if ((bitpos & 31) && lane == 0) {
            if (wbase < slot_words) outw[wbase] = bswap32(win[0]);
            else err |= FG_ERR_SLOT;
        }
        err = wave_or(err);
    }

    // ---------------------------------------------------------------- frame header (A.8)
    __device__ void write_header(uint32_t ca, uint32_t frame_number)
    {
        uint8_t *hb = (uint8_t *)misc;      // header bytes assembled by lane 0
        uint32_t hl = 0;
        if (lane == 0) {
            uint32_t u, bs_hint = 0, sr_hint = 0;
            hb[hl++] = 0xFF; hb[hl++] = 0xF8;
            switch (n) {
            case 192: u = 1; break; case 576: u = 2; break; case 1152: u = 3; break; case 2304: u = 4; break;
            case 4608: u = 5; break; case 256: u = 8; break; case 512: u = 9; break; case 1024: u = 10; break;
            case 2048: u = 11; break; case 4096: u = 12; break; case 8192: u = 13; break; case 16384: u = 14; break;
            case 32768: u = 15; break;
            default: bs_hint = u = (n <= 0x100) ? 6 : 7; break;
            }
            uint32_t b2 = u << 4;
            const uint32_t sr = P.sample_rate;
            switch (sr) {
            case 88200: u = 1; break; case 176400: u = 2; break; case 192000: u = 3; break; case 8000: u = 4; break;
            case 16000: u = 5; break; case 22050: u = 6; break; case 24000: u = 7; break; case 32000: u = 8; break;
            case 44100: u = 9; break; case 48000: u = 10; break; case 96000: u = 11; break;
            default:
                if (sr <= 255000 && sr % 1000 == 0) sr_hint = u = 12;
                else if (sr <= 655350 && sr % 10 == 0) sr_hint = u = 14;
                else if (sr <= 0xffff) sr_hint = u = 13;
                else u = 0;
                break;
            }
            hb[hl++] = (uint8_t)(b2 | u);
            switch (ca) { case 0: u = P.channels - 1; break; case 1: u = 8; break; case 2: u = 9; break; default: u = 10; break; }
            uint32_t b3 = u << 4;
            switch (P.bps) { case 8: u = 1; break; case 12: u = 2; break; case 16: u = 4; break; case 20: u = 5; break;
                             case 24: u = 6; break; case 32: u = 7; break; default: u = 0; break; }
            hb[hl++] = (uint8_t)(b3 | (u << 1));
            const uint32_t v = frame_number;
            if (v < 0x80) hb[hl++] = (uint8_t)v;
            else if (v < 0x800) { hb[hl++] = 0xC0 | (v >> 6); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x10000) { hb[hl++] = 0xE0 | (v >> 12); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x200000) { hb[hl++] = 0xF0 | (v >> 18); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x4000000) { hb[hl++] = 0xF8 | (v >> 24); hb[hl++] = 0x80 | ((v >> 18) & 0x3F); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else { hb[hl++] = 0xFC | (v >> 30); hb[hl++] = 0x80 | ((v >> 24) & 0x3F); hb[hl++] = 0x80 | ((v >> 18) & 0x3F); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            if (bs_hint == 6) hb[hl++] = (uint8_t)(n - 1);
            else if (bs_hint == 7) { hb[hl++] = (uint8_t)((n - 1) >> 8); hb[hl++] = (uint8_t)(n - 1); }
            if (sr_hint == 12) hb[hl++] = (uint8_t)(sr / 1000);
            else if (sr_hint == 13) { hb[hl++] = (uint8_t)(sr >> 8); hb[hl++] = (uint8_t)sr; }
            else if (sr_hint == 14) { hb[hl++] = (uint8_t)((sr / 10) >> 8); hb[hl++] = (uint8_t)(sr / 10); }
            uint32_t c8 = 0;
            for (uint32_t i = 0; i < hl; i++) {
                c8 ^= hb[i];
                for (int b = 0; b < 8; b++) c8 = (c8 & 0x80) ? (((c8 << 1) ^ 0x07) & 0xFF) : ((c8 << 1) & 0xFF);
            }
            hb[hl++] = (uint8_t)c8;
        }
        hl = bcast(hl, 0);
        lds_fence();
        const uint32_t v = (uint32_t)lane < hl ? hb[lane] : 0, b = (uint32_t)lane < hl ? 8 : 0;
        lds_fence();
        bw_round(0, 0, v, b, b);
    }

    // ---------------------------------------------------------------- one subframe (A.8)
    __device__ void write_subframe(int c, uint32_t di)
    {
        const Dec *d = &decs[di];
        const uint32_t type = d->type, order = d->order, w = d->wasted, sb = d->sbps;
        const uint32_t mask = sb < 32 ? ((1u << sb) - 1) : 0xFFFFFFFFu;
        uint32_t hdr;
        switch (type) {
        case 0: hdr = 0x00; break;
        case 1: hdr = 0x02; break;
        case 2: hdr = 0x10 | (order << 1); break;
        default: hdr = 0x40 | ((order - 1) << 1); break;
        }
        bw_put(hdr | (w ? 1 : 0), 8);
        if (w) {
            // unary: w-1 zeros then a one
            const uint32_t nb = lane == 0 ? w : 0;
            bw_round(0, 0, lane == 0 ? 1 : 0, lane == 0 ? 1 : 0, nb);
        }
        // a sample field is sb bits wide; sb == 33 is sent as its sign bit followed by 32 bits
        const uint32_t sb_lo = sb > 32 ? 32 : sb, sb_hi = sb > 32 ? 1 : 0;
        if (type == 0) {
            const i64 v = get64(c, 0, w);
            bw_round(lane == 0 ? (uint32_t)(v < 0) : 0, lane == 0 ? sb_hi : 0, lane == 0 ? ((uint32_t)v & mask) : 0, lane == 0 ? sb_lo : 0,
                     lane == 0 ? sb_lo : 0);
            return;
        }
        if (type == 1) {
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool on = i < n;
                const i64 v = on ? get64(c, i, w) : 0;
                bw_round(on ? (uint32_t)(v < 0) : 0, on ? sb_hi : 0, on ? ((uint32_t)v & mask) : 0, on ? sb_lo : 0, on ? sb_lo : 0);
            }
            return;
        }
        {   // warm-up samples
            const bool on = (uint32_t)lane < order;
            const i64 v = on ? get64(c, lane, w) : 0;
            bw_round(on ? (uint32_t)(v < 0) : 0, on ? sb_hi : 0, on ? ((uint32_t)v & mask) : 0, on ? sb_lo : 0, on ? sb_lo : 0);
        }
        const int32_t *q = d->q;
        const int shift = d->shift;
        if (type == 3) {
            const uint32_t prec = d->prec;
            // lane 0: precision-1 (4 bits) as prefix + shift (5 bits); lanes 1..order: coefficients
            uint32_t pv = 0, pb = 0, val = 0, vb = 0;
            if (lane == 0) { pv = prec - 1; pb = 4; val = (uint32_t)shift & 31; vb = 5; }
            else if ((uint32_t)lane <= order) { val = (uint32_t)q[lane - 1] & ((1u << prec) - 1); vb = prec; }
            bw_round(pv, pb, val, vb, vb);
        }
        const uint32_t po = d->porder, method = d->method;
        bw_put((method << 4) | po, 6);
        const uint32_t plen = method ? 5 : 4;
        const uint32_t psz = n >> po;
        const uint32_t inv = psz > 1 ? (uint32_t)(0xFFFFFFFFull / psz) + 1 : 0;
        const int kind = type == 2 ? 0 : 1;
        const bool narrow = kind == 1 ? narrow_ok(sb, order, q) : true;
        uint32_t ovf = 0;
        for (uint32_t i0 = (order / 64) * 64; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint32_t pv = 0, pb = 0, val = 0, vb = 0, nb = 0;
            if (i < n && i >= order) {
                const int32_t r = residual_at(c, w, i, kind, order, q, shift, narrow, &ovf);
                const uint32_t p = po ? (psz > 1 ? (uint32_t)(((u64)i * inv) >> 32) : i) : 0;
                const uint32_t k = d->k[p];
                const uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
                const uint32_t msb = u >> k;
                val = (1u << k) | (u & ((1u << k) - 1));
                vb = k + 1;
                nb = msb + 1 + k;
                const uint32_t pstart = p == 0 ? order : p * psz;
                if (i == pstart) { pv = k; pb = plen; }
            }
            bw_round(pv, pb, val, vb, nb);
        }
    }

    // ---------------------------------------------------------------- frame footer: pad + CRC-16
    __device__ uint32_t finish_frame()
    {
        payload_bits = bitpos;
        if (bitpos & 7) bitpos += 8 - (bitpos & 7);
        bw_flush(bitpos);
        bw_flush_all();
        __threadfence_block();
        const uint32_t nbytes = bitpos >> 3;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        // lanes take words q = 64*t + lane - pad (front padded with zero words: no effect, CRC init is 0)
        const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
        uint32_t s = 0;
        const uint16_t *t0 = crct, *thi = crct + 256, *tlo = crct + 512;
        for (uint32_t t = 0; t < T; t++) {
            const int qi = (int)(t * 64 + lane) - (int)pad;
            uint32_t wv = 0;
            if (qi >= 0) wv = bswap32(__builtin_nontemporal_load(&outw[qi]));
            // state * x^2048  (+) crc of the 4 bytes
            s = thi[s >> 8] ^ tlo[s & 0xFF];
            uint32_t cw = 0;
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 24)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 16)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 8)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ wv) & 0xFF];
            s ^= cw;
        }
        // fold: lane's polynomial is followed by (63 - lane) words
        const uint32_t mult = misc[64 + (63 - lane)];
        s = gf16_mul(s, mult);
        uint32_t crc = wave_xor(s);
        if (tail) {
            const uint32_t wv = W < slot_words ? bswap32(__builtin_nontemporal_load(&outw[W])) : 0;
            for (uint32_t b = 0; b < tail; b++)
                crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ (wv >> (24 - 8 * b))) & 0xFF];
        }
        bw_put(crc, 16);
        bw_flush_all();
        return bitpos >> 3;
    }
};

// (the CRC-16 tables -- crctab -- are built on the host and copied at context creation: fg_ctx.cpp fg_crc_tables_host, which also
// describes their layout)
__global__ void __launch_bounds__(64)
fg_encode_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, uint8_t *out,
                 FgBlockResult *results, FgDebugRec *dbg, const uint16_t *crctab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FgBlockDesc d = descs[blockIdx.x];
    Enc e;
    e.P = P;
    e.lane = threadIdx.x;
    e.n = d.n;
    e.view = d.reserved;
    e.err = 0;
    e.lmb_forced = false; e.lmb_forced_ca = d.forced_ca;
    e.window = windows + d.win_off;
    // ---- LDS carve (all offsets multiples of 16)
    size_t off = 0;
    auto carve = [&](size_t bytes) { unsigned char *p = smem + off; off += (bytes + 15) & ~(size_t)15; return p; };
    e.s0 = (int32_t *)carve((size_t)P.sig_stride * 4);
    e.s1 = (int32_t *)carve((size_t)P.sig_stride * 4);
    e.sstr = 1;
    e.dbuf = (double *)carve(P.lds_dbuf_bytes);
    e.autoc = (double *)carve((size_t)FG_MAX_CAND * P.nvec * (FG_MAX_ORDER + 1) * 8);
    e.qres = (int32_t *)carve((size_t)FG_MAX_CAND * P.nvec * FG_MAX_ORDER * 4);
    e.lres = (uint32_t *)carve((size_t)FG_MAX_CAND * P.nvec * 4);
    e.sums = (u64 *)carve((size_t)(2u << P.max_po) * 8);
    e.tmpk = (uint8_t *)carve(FG_MAX_PARTS);
    e.win = (uint32_t *)carve((FG_WINW + 2) * 4);
    e.crct = (uint16_t *)carve(768 * 2);
    e.misc = (uint32_t *)carve(128 * 4);
    e.decs = (Dec *)carve(sizeof(Dec) * 8);
    for (int j = e.lane; j < 768; j += 64) e.crct[j] = crctab[j];
    e.misc[64 + e.lane] = crctab[768 + e.lane];
    lds_fence();

    const uint32_t C = P.channels;
    e.wide = P.bps > 30;
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
    e.stamp(mydbg, 0);
    uint32_t ca = 0;
    uint32_t sub_c[8], sub_d[8];
    uint32_t nsub = C;
    if (C == 2 && P.do_mid_side) {
        e.mode = 1; e.ncand = 4;
        e.stage(pcm, d.pcm_off, 0, 2);
        e.analyse(0, mydbg);
        uint32_t b0 = e.decs[0].bits, b1 = e.decs[1].bits, b2 = e.decs[2].bits, b3 = e.decs[3].bits;
        if (d.forced_ca != 0xFF) ca = d.forced_ca & 0x7F;       // (bit 7: a loose mid-side DECISION frame, see FgBlockDesc)
        else {
            uint32_t bits[4] = {b0 + b1, b0 + b3, b1 + b3, b2 + b3};
            uint32_t mn = bits[0];
            for (uint32_t k = 1; k <= 3; k++) if (bits[k] < mn) { mn = bits[k]; ca = k; }
        }
        switch (ca) { case 0: sub_c[0] = 0; sub_c[1] = 1; break; case 1: sub_c[0] = 0; sub_c[1] = 3; break;
                      case 2: sub_c[0] = 3; sub_c[1] = 1; break; default: sub_c[0] = 2; sub_c[1] = 3; break; }
        sub_d[0] = sub_c[0]; sub_d[1] = sub_c[1];
        if (e.lane == 0) {
            FgBlockResult *r = &results[d.out_slot];
            r->best_bits[0] = b0; r->best_bits[1] = b1; r->best_bits[2] = b2; r->best_bits[3] = b3;
        }
    }
    else {
        e.mode = 0;
        for (uint32_t ch0 = 0; ch0 < C; ch0 += 2) {
            const uint32_t nch = (C - ch0) < 2 ? (C - ch0) : 2;
            e.ncand = (int)nch;
            e.stage(pcm, d.pcm_off, ch0, nch);
            e.analyse(ch0, (mydbg && ch0 == 0) ? mydbg : nullptr);
        }
        for (uint32_t ch = 0; ch < C; ch++) { sub_c[ch] = ch & 1; sub_d[ch] = ch; }
        if (e.lane == 0) {
            FgBlockResult *r = &results[d.out_slot];
            for (uint32_t k = 0; k < 4; k++) r->best_bits[k] = k < C ? e.decs[k].bits : 0;
        }
    }
    if (mydbg && e.lane < 4 && e.lane < ((C == 2 && P.do_mid_side) ? 4 : (C < 2 ? C : 2))) {
        const Dec *dd = &e.decs[e.lane];
        FgDebugCand *dc = &mydbg->cand[e.lane];
        dc->wasted = dd->wasted; dc->sbps = dd->sbps; dc->type = dd->type; dc->order = dd->type >= 2 ? dd->order : 0;
        dc->precision = dd->type == 3 ? dd->prec : 0; dc->shift = dd->type == 3 ? dd->shift : 0;
        dc->bits = dd->bits; dc->porder = dd->type >= 2 ? dd->porder : 0; dc->rice_method = dd->type >= 2 ? dd->method : 0;
        for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (dd->type == 3 && j < dd->order) ? dd->q[j] : 0;
        for (uint32_t j = 0; j < FG_MAX_PARTS; j++) dc->rice_params[j] = (dd->type >= 2 && j < (1u << dd->porder)) ? dd->k[j] : 0;
    }
    // ---- pack
    e.bw_init((uint32_t *)(out + (size_t)d.out_slot * P.slot_bytes), P.slot_bytes / 4);
    e.stamp(mydbg, 7);
    e.write_header(ca, d.frame_number);
    uint32_t staged_group = (C <= 2) ? 0 : 0xFFFFFFFFu;
    for (uint32_t sidx = 0; sidx < nsub; sidx++) {
        if (e.mode == 0 && C > 2) {
            const uint32_t g = sidx / 2;
            if (g != staged_group) {
                const uint32_t ch0 = g * 2, nch = (C - ch0) < 2 ? (C - ch0) : 2;
                e.stage(pcm, d.pcm_off, ch0, nch);
                staged_group = g;
            }
        }
        e.write_subframe((int)sub_c[sidx], sub_d[sidx]);
    }
    e.stamp(mydbg, 8);
    const uint32_t bytes = e.finish_frame();
    e.stamp(mydbg, 9);
    if (e.lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->bytes = bytes; r->ca = ca; r->err = e.err; r->reserved = 0;
        if (C == 1) r->best_bits[3] = e.payload_bits;
    }
}

// ------------------------------------------------------------------ compaction: slots -> contiguous stream
// offsets[b] = sum of bytes of blocks < b (exclusive), offsets[nblocks] = total, offsets[nblocks+1] = OR of errors.
// One workgroup of 1024 threads walks the blocks in tiles of 8192: the sizes are fetched coalesced into LDS (a pipeline
// block -- results.reserved == 4 -- gets its size here, from the bit counts of its four chunks: ceil(sum / 8) + 2 for the
// CRC-16), every thread then sums eight neighbours, one workgroup scan per tile, carry into the next tile.  All loads of a
// tile are issued before anything is stored (the stores into `results` would otherwise fence the later loads).
#define FG_SCAN_TILE 8192
#define FG_SCAN_PER (FG_SCAN_TILE / 1024)
// `all_pipe` (every block went through the pipeline, first pass): the sizes come from the chunk bit counts alone -- 16 bytes
// a block, coalesced -- and the OR of the error flags from the pipeline's own word (errs_in); nothing is read from the
// 32-byte result records, whose 21 k strided lines through one CU's L1 were most of this kernel's time.  (A block that
// waits for the generic kernel gets a wrong size here: the host sees FG_ERR_REDO in the flags and repeats the scan in the
// general form after the redo.)
// one tile; returns its total.  `write`: offsets and (general form) the sizes of the pipeline's blocks are stored
__device__ __forceinline__ u64 fg_scan_tile(FgBlockResult *results, const uint32_t *chunk_bits, uint32_t nblocks, u64 *offsets, uint32_t all_pipe,
                                            uint32_t t0, u64 carry, uint32_t *sz, u64 *wtot, uint32_t &e, bool write)
{
    const uint32_t tid = threadIdx.x;
    uint4 r[FG_SCAN_PER], cb[FG_SCAN_PER];
    uint32_t kind[FG_SCAN_PER];
#pragma unroll
    for (uint32_t j = 0; j < FG_SCAN_PER; j++) {
        const uint32_t b = t0 + j * 1024 + tid;
        r[j] = make_uint4(0, 0, 0, 0); cb[j] = r[j]; kind[j] = 0;
        if (b < nblocks) {
            if (all_pipe) kind[j] = 4;
            else {
                r[j] = *(const uint4 *)&results[b];                    // bytes, ca, err, best_bits[0]
                kind[j] = results[b].reserved;
            }
            if (chunk_bits) cb[j] = *(const uint4 *)&chunk_bits[(size_t)b * 4];
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < FG_SCAN_PER; j++) {
        const uint32_t b = t0 + j * 1024 + tid;
        uint32_t bytes = r[j].x;
        e |= r[j].z;
        if (chunk_bits && kind[j] == 4) {
            bytes = (b >= nblocks || (r[j].z & FG_ERR_REDO)) ? 0u : ((cb[j].x + cb[j].y + cb[j].z + cb[j].w + 7) >> 3) + 2;
            if (write && !all_pipe && b < nblocks) results[b].bytes = bytes;       // (all_pipe: the assembly kernel writes it back)
        }
        sz[j * 1024 + tid] = bytes;
    }
    __syncthreads();
    uint32_t v[FG_SCAN_PER];
    u64 mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < FG_SCAN_PER; j += 4) {
        const uint4 t = *(const uint4 *)&sz[tid * FG_SCAN_PER + j];
        v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
        mine += (u64)t.x + t.y + t.z + t.w;
    }
    u64 total;
    u64 run = carry + fgdev::block_scan_excl_u64(mine, wtot, &total);
    if (write) {
#pragma unroll
        for (uint32_t j = 0; j < FG_SCAN_PER; j++) {
            const uint32_t b = t0 + tid * FG_SCAN_PER + j;
            if (b < nblocks) offsets[b] = run;
            run += v[j];
        }
    }
    return total;
}

__global__ void __launch_bounds__(1024)
fg_scan_sizes_kernel(FgBlockResult *results, const uint32_t *chunk_bits, uint32_t nblocks, u64 *offsets, uint32_t all_pipe,
                     const unsigned long long *errs_in)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t errs;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_SCAN_TILE];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) errs = 0;
    uint32_t e = 0;
    u64 carry = 0;
    for (uint32_t t0 = 0; t0 < nblocks; t0 += FG_SCAN_TILE) {
        __syncthreads();                    // (previous tile's readers are done with sz[]; errs initialised)
        carry += fg_scan_tile(results, chunk_bits, nblocks, offsets, all_pipe, t0, carry, sz, wtot, e, true);
    }
    if (e) atomicOr(&errs, e);
    __syncthreads();
    if (tid == 0) { offsets[nblocks] = carry; offsets[nblocks + 1] = (all_pipe && errs_in) ? (u64)errs_in[0] : (u64)errs; }
}

// More than one tile (batches of many streams: 90 112 blocks took one workgroup 0.22 ms): one workgroup per tile.  First
// every tile's total and error flags (tsum[t], tsum[ntiles + t]; nothing else is stored), then every workgroup adds up the
// totals in front of its tile and scans it; workgroup 0 also writes the grand total and the flags.
__global__ void __launch_bounds__(1024)
fg_scan_sums_kernel(FgBlockResult *results, const uint32_t *chunk_bits, uint32_t nblocks, uint32_t all_pipe, u64 *tsum, uint32_t ntiles)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t errs;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_SCAN_TILE];
    if (threadIdx.x == 0) errs = 0;
    __syncthreads();
    uint32_t e = 0;
    const u64 total = fg_scan_tile(results, chunk_bits, nblocks, nullptr, all_pipe, blockIdx.x * FG_SCAN_TILE, 0, sz, wtot, e, false);
    if (e) atomicOr(&errs, e);
    __syncthreads();
    if (threadIdx.x == 0) { tsum[blockIdx.x] = total; tsum[ntiles + blockIdx.x] = errs; }
}

__global__ void __launch_bounds__(1024)
fg_scan_tiles_kernel(FgBlockResult *results, const uint32_t *chunk_bits, uint32_t nblocks, u64 *offsets, uint32_t all_pipe,
                     const unsigned long long *errs_in, const u64 *tsum, uint32_t ntiles)
{
    __shared__ u64 wtot[16];
    __shared__ uint32_t errs;
    __shared__ __attribute__((aligned(16))) uint32_t sz[FG_SCAN_TILE];
    const uint32_t tid = threadIdx.x, t = blockIdx.x;
    if (tid == 0) errs = 0;
    u64 before = 0, all = 0;
    uint32_t eo = 0;
    for (uint32_t u = tid; u < ntiles; u += 1024) {
        const u64 x = tsum[u];
        all += x;
        if (u < t) before += x;
        eo |= (uint32_t)tsum[ntiles + u];
    }
    u64 base, grand;
    (void)fgdev::block_scan_excl_u64(before, wtot, &base);
    __syncthreads();
    (void)fgdev::block_scan_excl_u64(all, wtot, &grand);
    if (t == 0 && eo) atomicOr(&errs, eo);
    __syncthreads();
    if (t == 0 && tid == 0) { offsets[nblocks] = grand; offsets[nblocks + 1] = (all_pipe && errs_in) ? (u64)errs_in[0] : (u64)errs; }
    uint32_t e = 0;
    (void)fg_scan_tile(results, chunk_bits, nblocks, offsets, all_pipe, t * FG_SCAN_TILE, base, sz, wtot, e, true);
}

__global__ void __launch_bounds__(256)
fg_compact_kernel(const uint8_t *slots, uint32_t slot_bytes, const FgBlockResult *results, const u64 *offsets,
                  uint8_t *dst, u64 dst_cap)
{
    const uint32_t b = blockIdx.x;
    const uint32_t nb = results[b].bytes;
    if (offsets[b] + nb > dst_cap) return;          // the host reports the short buffer once it has read the total
    const uint8_t *src = slots + (size_t)b * slot_bytes;
    uint8_t *d = dst + offsets[b];
    // head bytes up to a 4-byte boundary of dst, then word copies with a funnel shift on the source
    const uint32_t mis = (uint32_t)((uintptr_t)d & 3);
    const uint32_t head = mis ? (4 - mis) : 0;
    const uint32_t h = head < nb ? head : nb;
    if (threadIdx.x < h) d[threadIdx.x] = src[threadIdx.x];
    const uint32_t nw = (nb - h) >> 2;
    const uint32_t *sw = (const uint32_t *)src;   // slot is 4-byte aligned
    uint32_t *dw = (uint32_t *)(d + h);
    const uint32_t shb = h * 8;                   // source byte offset h (0..3) -> bit shift
    for (uint32_t j = threadIdx.x; j < nw; j += 256) {
        uint32_t lo = sw[j], v;
        if (shb == 0) v = lo;
        else { uint32_t hi = sw[j + 1]; v = (lo >> shb) | (hi << (32 - shb)); }
        dw[j] = v;
    }
    const uint32_t done = h + nw * 4;
    if (threadIdx.x < nb - done) d[done + threadIdx.x] = src[done + threadIdx.x];
}


// ------------------------------------------------------------------ end-of-call hand-over to the host
// The batch calls end with one tiny kernel instead of device-to-host copies, an event and a stream synchronisation: it writes
// the few words the host wants (totals, error flags, counters) and two time stamps (constant-rate wall clock) into a pinned
// landing area and then raises a sequence number there, which the host polls.  The stream orders it behind every kernel
// of the call.  host[0] = sequence, host[2 .. 10) = payload, host[10] = start stamp, host[11] = end stamp.
__global__ void fg_stamp_kernel(u64 *stamp)
{
    if (threadIdx.x == 0) stamp[0] = wall_clock64();
}

__device__ __forceinline__ void fg_signal_tail(const u64 *src0, uint32_t n0, const u64 *src1, uint32_t n1, const u64 *stamp, u64 *host, u64 seq,
                                               u64 *reset = nullptr)
{
    for (uint32_t i = 0; i < n0; i++) host[2 + i] = src0[i];
    for (uint32_t i = 0; i < n1; i++) host[2 + n0 + i] = src1[i];
    // (reset: the encoder pipeline's guard counters, read by now, go back to their start values for the next call -- FgPipeLaunch.guard_clean)
    if (reset) { reset[0] = 0ull; reset[1] = 0x7FF0000000000000ull; reset[2] = 0ull; }
    host[10] = stamp ? stamp[0] : 0;
    host[11] = wall_clock64();
    __threadfence_system();
    __hip_atomic_store(&host[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void fg_signal_kernel(const u64 *src0, uint32_t n0, const u64 *src1, uint32_t n1, const u64 *stamp, u64 *host, u64 seq, u64 *reset)
{
    if (threadIdx.x == 0) fg_signal_tail(src0, n0, src1, n1, stamp, host, seq, reset);
}
// end of a direct-packing call without blocks in the chunk form: total from the last frame's workgroup, the error flags and the guard
// counters straight from the pipeline's words (guard[2], [0], [1])
__global__ void fg_signal_direct_kernel(const u64 *total, const u64 *guard, const u64 *stamp, u64 *host, u64 seq, u64 *reset)
{
    if (threadIdx.x != 0) return;
    host[2] = total[0]; host[3] = guard[2]; host[4] = guard[0]; host[5] = guard[1];
    fg_signal_tail(nullptr, 0, nullptr, 0, stamp, host, seq, reset);
}

// ------------------------------------------------------------------ STREAMINFO MD5 of device-resident streams (stage L1, format.h:543)
// MD5 is a chain over the 64-byte blocks of ONE stream, so the only parallelism is across streams: thread = stream.  The bytes are
// the samples little-endian at (bits per sample + 7) / 8 bytes each, channels interleaved, exactly what libFLAC hashes; a stream
// of 60 s of 16-bit stereo is 180 000 blocks of ~550 dependent instructions -- about 0.2 s for any number of streams up to the
// lanes of the chip, which is why the batch entry point does not do it unasked.
struct FgMd5Job { unsigned long long pcm_off, nsamples; };      // first inter-channel sample, inter-channel samples
__global__ void __launch_bounds__(64)
fg_md5_streams_kernel(const void *pcm, uint32_t pcm_i16, uint32_t channels, uint32_t bps, const FgMd5Job *jobs, uint32_t njobs, uint32_t *out)
{
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const uint32_t nbytes = (bps + 7) / 8;
    const u64 nvalues = jobs[j].nsamples * channels, first = jobs[j].pcm_off * channels;
    const u64 total = nvalues * nbytes;                 // bytes of the message
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const uint8_t R[16] = {7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21};
    u64 vi = 0;                    // next value of the stream
    u64 acc = 0;                   // bytes not yet handed out, lowest first
    uint32_t have = 0;             // ... and how many
    auto value = [&](u64 i) -> uint32_t {
        return pcm_i16 ? (uint32_t)(int32_t)((const int16_t *)pcm)[first + i] : (uint32_t)((const int32_t *)pcm)[first + i];
    };
    const u64 nblocks = (total + 8) / 64 + 1;      // message, the 0x80 byte, zeros, eight bytes of length
    auto rounds = [&](const uint32_t (&w)[16]) __attribute__((always_inline)) {
        uint32_t A = a, Bv = b, Cv = c, Dv = d;
#pragma unroll
        for (int i = 0; i < 64; i++) {
            uint32_t f, g;
            if (i < 16) { f = (Bv & Cv) | (~Bv & Dv); g = (uint32_t)i; }
            else if (i < 32) { f = (Dv & Bv) | (~Dv & Cv); g = (5u * i + 1) & 15; }
            else if (i < 48) { f = Bv ^ Cv ^ Dv; g = (3u * i + 5) & 15; }
            else { f = Cv ^ (Bv | ~Dv); g = (7u * i) & 15; }
            const uint32_t x = A + f + K[i] + w[g];
            const uint32_t r = R[(i >> 4) * 4 + (i & 3)];
            A = Dv; Dv = Cv; Cv = Bv;
            Bv = Bv + ((x << r) | (x >> (32 - r)));
        }
        a += A; b += Bv; c += Cv; d += Dv;
    };
    // The whole 64-byte blocks of 16-bit material in an int32 container -- the batch encoder's input --: 32 values a block, fetched as
    // eight 16-byte loads, the NEXT block's while this one's 64 rounds run (a lane walks its own stream, so every load is a line of its
    // own: one value at a time the kernel ran at the latency of a load per value -- 1.9 s for 128 streams of 60 s, 6 MB/s a stream).
    u64 blk0 = 0;
    if (nbytes == 2 && !pcm_i16) {
        struct __attribute__((packed, aligned(4))) V4 { uint32_t v[4]; };
        const V4 *src = (const V4 *)((const int32_t *)pcm + first);
        const u64 full = total / 64;
        V4 cur[8], nxt[8];
        if (full) {
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = src[k];
        }
        for (u64 blk = 0; blk < full; blk++) {
            if (blk + 1 < full) {
#pragma unroll
                for (int k = 0; k < 8; k++) nxt[k] = src[(blk + 1) * 8 + k];
            }
            uint32_t w[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                w[2 * k] = (cur[k].v[0] & 0xFFFFu) | (cur[k].v[1] << 16);
                w[2 * k + 1] = (cur[k].v[2] & 0xFFFFu) | (cur[k].v[3] << 16);
            }
            rounds(w);
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = nxt[k];
        }
        blk0 = full;
        vi = full * 32;
    }
    bool pad_done = false;
    for (u64 blk = blk0; blk < nblocks; blk++) {
        uint32_t w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            while (have < 4 && vi < nvalues) {
                const u64 v = (u64)(value(vi) & (nbytes == 4 ? 0xFFFFFFFFu : ((1u << (8 * nbytes)) - 1)));
                acc |= v << (8 * have);
                have += nbytes;
                vi++;
            }
            if (have < 4 && !pad_done) { acc |= (u64)0x80 << (8 * have); have = 8; pad_done = true; }    // (the zeros behind it come for free)
            w[t] = (uint32_t)acc;
            acc >>= 32;
            have = have >= 4 ? have - 4 : 0;
        }
        if (blk + 1 == nblocks) { w[14] = (uint32_t)(total * 8); w[15] = (uint32_t)((total * 8) >> 32); }
        rounds(w);
    }
    out[4 * j] = a; out[4 * j + 1] = b; out[4 * j + 2] = c; out[4 * j + 3] = d;
}

}  // namespace

// ------------------------------------------------------------------ host-callable launchers (C ABI, used by flacgpu_api.cpp)
extern "C" int fg_func_set_lds(const void *fn, size_t bytes);   // fg_ctx.cpp: per device, thread-safe
extern "C" {

size_t fg_enc_lds_bytes(const FgEncParams *P)
{
    size_t off = 0;
    auto add = [&](size_t b) { off += (b + 15) & ~(size_t)15; };
    add((size_t)P->sig_stride * 4);
    add((size_t)P->sig_stride * 4);
    add(P->lds_dbuf_bytes);
    add((size_t)FG_MAX_CAND * P->nvec * (FG_MAX_ORDER + 1) * 8);
    add((size_t)FG_MAX_CAND * P->nvec * FG_MAX_ORDER * 4);
    add((size_t)FG_MAX_CAND * P->nvec * 4);
    add((size_t)(2u << P->max_po) * 8);
    add(FG_MAX_PARTS);
    add((FG_WINW + 2) * 4);
    add(768 * 2);
    add(128 * 4);
    add(sizeof(Dec) * 8);
    return off;
}

int fg_launch_stamp(unsigned long long *d_stamp, hipStream_t stream)
{
    hipLaunchKernelGGL(fg_stamp_kernel, dim3(1), dim3(64), 0, stream, (u64 *)d_stamp);
    return (int)hipGetLastError();
}

int fg_launch_signal(const unsigned long long *src0, uint32_t n0, const unsigned long long *src1, uint32_t n1,
                     const unsigned long long *d_stamp, unsigned long long *h_sig, unsigned long long seq, hipStream_t stream,
                     unsigned long long *d_reset)
{
    hipLaunchKernelGGL(fg_signal_kernel, dim3(1), dim3(64), 0, stream, (const u64 *)src0, n0, (const u64 *)src1, n1, (const u64 *)d_stamp,
                       (u64 *)h_sig, (u64)seq, (u64 *)d_reset);
    return (int)hipGetLastError();
}

int fg_launch_signal_direct(const unsigned long long *d_total, const unsigned long long *d_guard, const unsigned long long *d_stamp,
                            unsigned long long *h_sig, unsigned long long seq, hipStream_t stream, unsigned long long *d_reset)
{
    hipLaunchKernelGGL(fg_signal_direct_kernel, dim3(1), dim3(64), 0, stream, (const u64 *)d_total, (const u64 *)d_guard, (const u64 *)d_stamp,
                       (u64 *)h_sig, (u64)seq, (u64 *)d_reset);
    return (int)hipGetLastError();
}

int fg_launch_md5_streams(const void *d_pcm, uint32_t pcm_i16, uint32_t channels, uint32_t bps, const void *d_jobs, uint32_t njobs,
                          uint32_t *d_out, hipStream_t stream)
{
    if (njobs == 0) return 0;
    hipLaunchKernelGGL(fg_md5_streams_kernel, dim3((njobs + 63) / 64), dim3(64), 0, stream, d_pcm, pcm_i16, channels, bps, (const FgMd5Job *)d_jobs,
                       njobs, d_out);
    return (int)hipGetLastError();
}

int fg_launch_encode(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                     uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                     const uint16_t *d_crctab, hipStream_t stream)
{
    if (nblocks == 0) return 0;
    const size_t lds = fg_enc_lds_bytes(P);
    { const int e = fg_func_set_lds((const void *)fg_encode_kernel, lds); if (e != 0) return e; }
    hipLaunchKernelGGL(fg_encode_kernel, dim3(nblocks), dim3(64), lds, stream, d_pcm, d_descs, d_windows, *P, d_slots,
                       d_results, d_dbg, d_crctab);
    return (int)hipGetLastError();
}

// int16 -> int32 (the stream encoder uploads 16-bit input as it is)
__global__ void fg_widen16_kernel(const int16_t *src, int32_t *dst, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
int fg_launch_widen16(const int16_t *d_src, int32_t *d_dst, unsigned long long n, hipStream_t stream)
{
    if (n == 0) return 0;
    unsigned long long wg = (n + 255) / 256;
    if (wg > 4096) wg = 4096;
    hipLaunchKernelGGL(fg_widen16_kernel, dim3((unsigned)wg), dim3(256), 0, stream, d_src, d_dst, n);
    return (int)hipGetLastError();
}

// 64-bit words the offsets array of fg_launch_scan needs for `nblocks` blocks
size_t fg_scan_words(uint32_t nblocks) { return (size_t)nblocks + 4 + 2 * (size_t)((nblocks + FG_SCAN_TILE - 1) / FG_SCAN_TILE) + 2; }

int fg_launch_scan(FgBlockResult *d_results, const uint32_t *d_chunk_bits, uint32_t nblocks, unsigned long long *d_offsets, int all_pipe,
                   const unsigned long long *d_errs, hipStream_t stream)
{
    if (nblocks == 0) return 0;
    const uint32_t ap = (uint32_t)((all_pipe && d_chunk_bits && d_errs) ? 1 : 0);
    const uint32_t ntiles = (nblocks + FG_SCAN_TILE - 1) / FG_SCAN_TILE;
    if (ntiles == 1) hipLaunchKernelGGL(fg_scan_sizes_kernel, dim3(1), dim3(1024), 0, stream, d_results, d_chunk_bits, nblocks, d_offsets, ap, d_errs);
    else {
        // (the tile totals lie behind offsets[nblocks .. nblocks + 3]: the caller sizes the array with fg_scan_words())
        u64 *tsum = (u64 *)d_offsets + nblocks + 4;
        hipLaunchKernelGGL(fg_scan_sums_kernel, dim3(ntiles), dim3(1024), 0, stream, d_results, d_chunk_bits, nblocks, ap, tsum, ntiles);
        hipLaunchKernelGGL(fg_scan_tiles_kernel, dim3(ntiles), dim3(1024), 0, stream, d_results, d_chunk_bits, nblocks, (u64 *)d_offsets, ap, d_errs,
                           (const u64 *)tsum, ntiles);
    }
    return (int)hipGetLastError();
}

int fg_launch_copy(const uint8_t *d_slots, uint32_t slot_bytes, const FgBlockResult *d_results, uint32_t nblocks,
                   const unsigned long long *d_offsets, uint8_t *d_dst, hipStream_t stream, uint64_t dst_cap)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(fg_compact_kernel, dim3(nblocks), dim3(256), 0, stream, d_slots, slot_bytes, d_results, d_offsets, d_dst, (u64)dst_cap);
    return (int)hipGetLastError();
}

}  // extern "C"
