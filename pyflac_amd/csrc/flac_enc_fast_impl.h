// flac_enc_fast_impl.h -- specialised one-block-per-wavefront FLAC frame encoder for the common shapes:
// 1 or 2 channels, block length a multiple of 64 whose finest Rice partition is a multiple of 64 samples
// (4096 at partition order <= 6), max LPC order <= 12, bits-per-sample <= 24.  Everything else goes to
// the generic kernel in flac_enc_kernels.hip; both produce identical bytes (tests/test_gpu_encode.py).
//
// Same algorithm and stage order as the generic kernel; what changes is the mapping:
//   * compile-time candidate set (L,R,M,S / L,R / mono) -- no per-sample branches;
//   * every pass handles all candidates at once from one set of LDS loads;
//   * fixed and LPC predictors share one FIR evaluation pass (a fixed predictor of order k is the FIR with
//     binomial coefficients and shift 0), coefficients live in SGPRs;
//   * reductions and prefix sums use DPP row shifts / broadcasts instead of LDS shuffles;
//   * Rice partition sums, parameters and bit estimates live in registers, lane = partition;
//   * LDS accesses of the single wave are ordered by issue, so stages are separated by compiler fences only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fg_dev.h"
#include "fg_types.h"

#define FG_LN2 0.69314718055994530942
#define FG_PADF 32   // zero samples kept in front of each staged channel (>= max order)
// LDS arrays are addressed through address_space(3) pointers: ds_* instructions, and no FLAT aperture checks
// on the (possibly negative) sample offsets.
#define LDS __attribute__((address_space(3)))

using namespace fgdev;

namespace {

template <bool MS, int NCH, int MAXO, bool ACC64>
struct Fast {
    static constexpr int NC = MS ? 4 : NCH;
    typedef typename std::conditional<ACC64, u64, uint32_t>::type sum_t;
    // bits-per-sample <= 16 (the !ACC64 shapes): samples are staged as int16, halving the LDS footprint
    typedef typename std::conditional<ACC64, int32_t, int16_t>::type samp_t;

    // LDS
    LDS samp_t *sL, *sR;
    LDS double *dbuf;
    LDS double *autoc;   // [NC][nvec][MAXO+1]
    LDS int32_t *qres;   // [NC*nvec][MAXO]
    LDS uint32_t *lres;  // [NC*nvec]
    LDS int32_t *bestq;  // [NC][MAXO]
    LDS uint32_t *win;
    LDS uint16_t *crct;
    LDS uint32_t *misc;
    // uniform
    const FgEncParams *pp;   // kernel argument block (uniform, scalar loads)
    const float *window;
    int lane;
    uint32_t n;
    uint32_t err;
    uint32_t wst[NC], sbp[NC];
    // decisions (uniform) + per-lane Rice parameters (lane = partition)
    uint32_t d_type[NC], d_order[NC], d_prec[NC], d_porder[NC], d_method[NC], d_bits[NC];
    int d_shift[NC];
    uint32_t d_k[NC];
    // bit writer
    uint32_t bitpos, wbase;
    uint32_t *outw;
    uint32_t slot_words;

    // ------------------------------------------------------------------ candidate values
    template <int C>
    __device__ __forceinline__ int32_t cv(int32_t L, int32_t R) const
    {
        if (!MS) return (C == 0 ? L : R) >> wst[C];
        if (C == 0) return L >> wst[0];
        if (C == 1) return R >> wst[1];
        if (C == 2) return ((L + R) >> 1) >> wst[2];
        return (L - R) >> wst[3];
    }

    // ------------------------------------------------------------------ staging
    __device__ __forceinline__ void stage(const void *pcm, u64 pcm_off)
    {
        const int32_t lim = (int32_t)(pp->bps - 1);
        uint32_t bad = 0;
        if (lane < FG_PADF) { sL[lane] = 0; if (NCH == 2) sR[lane] = 0; }
        LDS samp_t *dl = sL + FG_PADF, *dr = sR + FG_PADF;
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * 64 + lane;
                if (i < n) {
                    int32_t a, b = 0;
                    if (NCH == 2) {
                        if (pp->pcm_i16) { const short2 v = ((const short2 *)pcm)[pcm_off + i]; a = v.x; b = v.y; }
                        else { const int2 v = ((const int2 *)pcm)[pcm_off + i]; a = v.x; b = v.y; }
                    }
                    else {
                        if (pp->pcm_i16) a = ((const int16_t *)pcm)[pcm_off + i];
                        else a = ((const int32_t *)pcm)[pcm_off + i];
                    }
                    if (pp->bps < 32) bad |= (uint32_t)(((a ^ (a >> 31)) >> lim) | ((b ^ (b >> 31)) >> lim));
                    dl[i] = (samp_t)a;
                    if (NCH == 2) dr[i] = (samp_t)b;
                }
            }
        }
        if (__any(bad != 0)) err |= FG_ERR_RANGE;
        wave_lds_fence();
    }

    // ------------------------------------------------------------------ wasted bits + fixed-predictor error sums
    // One pass for all candidates.  The sums assume wasted == 0 (the overwhelmingly common case); candidates
    // with wasted bits are redone by fixed_sums_one().
    __device__ __forceinline__ void sums_pass(u64 tot[NC][5])
    {
        sum_t acc[NC][5];
        uint32_t orv[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            orv[c] = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) acc[c][k] = 0;
        }
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        for (uint32_t i = lane; i < n; i += 64) {
            int32_t l[5], r[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { l[k] = pl[(int)i - k]; r[k] = (NCH == 2) ? pr[(int)i - k] : 0; }
            const bool on = i >= 4;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int32_t v[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (!MS) v[k] = (c == 0) ? l[k] : r[k];
                    else v[k] = (c == 0) ? l[k] : (c == 1) ? r[k] : (c == 2) ? ((l[k] + r[k]) >> 1) : (l[k] - r[k]);
                }
                orv[c] |= (uint32_t)v[0];
                const int32_t e1 = v[0] - v[1], d1 = v[1] - v[2], d2 = v[2] - v[3], d3 = v[3] - v[4];
                const int32_t e2 = e1 - d1, f2 = d1 - d2, g2 = d2 - d3;
                const int32_t e3 = e2 - f2, f3 = f2 - g2;
                const int32_t e4 = e3 - f3;
                if (on) {
                    acc[c][0] += (uint32_t)abs(v[0]); acc[c][1] += (uint32_t)abs(e1); acc[c][2] += (uint32_t)abs(e2);
                    acc[c][3] += (uint32_t)abs(e3); acc[c][4] += (uint32_t)abs(e4);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t o = wave_or32(orv[c]);
            uint32_t w = o ? (uint32_t)__builtin_ctz(o) : 0;
            const uint32_t nominal = pp->bps + ((MS && c == 3) ? 1u : 0u);
            if (w > nominal) w = nominal;
            wst[c] = w; sbp[c] = nominal - w;
#pragma unroll
            for (int k = 0; k < 5; k++) tot[c][k] = ACC64 ? wave_sum64((u64)acc[c][k]) : (u64)wave_sum((uint32_t)acc[c][k]);
        }
    }

    template <int C>
    __device__ __forceinline__ void fixed_sums_one(u64 tot[5])
    {
        u64 acc[5] = {0, 0, 0, 0, 0};
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        for (uint32_t i = 4 + lane; i < n; i += 64) {
            int32_t v[5];
#pragma unroll
            for (int k = 0; k < 5; k++) v[k] = cv<C>(pl[(int)i - k], (NCH == 2) ? pr[(int)i - k] : 0);
            const int32_t e1 = v[0] - v[1], d1 = v[1] - v[2], d2 = v[2] - v[3], d3 = v[3] - v[4];
            const int32_t e2 = e1 - d1, f2 = d1 - d2, g2 = d2 - d3;
            const int32_t e3 = e2 - f2, f3 = f2 - g2;
            const int32_t e4 = e3 - f3;
            acc[0] += (uint32_t)abs(v[0]); acc[1] += (uint32_t)abs(e1); acc[2] += (uint32_t)abs(e2);
            acc[3] += (uint32_t)abs(e3); acc[4] += (uint32_t)abs(e4);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) tot[k] = wave_sum64(acc[k]);
    }

    template <int C>
    __device__ __forceinline__ bool is_constant()
    {
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        const int32_t x0 = cv<C>(pl[0], (NCH == 2) ? pr[0] : 0);
        uint32_t ne = 0;
        for (uint32_t i = lane; i < n; i += 64) ne |= (cv<C>(pl[i], (NCH == 2) ? pr[i] : 0) != x0);
        return !__any(ne != 0);
    }

    // ------------------------------------------------------------------ autocorrelation (order-preserving fp64 chains)
    // lane = 16*candidate + lag.  part == 0: whole block under the window; else FLAC__lpc_window_data_partial.
    __device__ __forceinline__ void autocorr_vector(uint32_t v, uint32_t vec_len, uint32_t part, uint32_t sh, uint32_t mo)
    {
        const uint32_t DSTR = FG_DH + FG_DK;
        const uint32_t cl = lane >> 4, l = lane & 15;
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        double acc = 0.0;
        for (uint32_t j = lane; j < NC * FG_DH; j += 64) dbuf[(j / FG_DH) * DSTR + (j % FG_DH)] = 0.0;
        wave_lds_fence();
        const bool on = cl < (uint32_t)NC && l <= mo;
        const LDS double *cur = dbuf + (on ? cl : 0) * DSTR + FG_DH;
        const LDS double *hist = cur - (on ? l : 0);
        for (uint32_t k0 = 0; k0 < vec_len; k0 += FG_DK) {
            const uint32_t kn = (vec_len - k0) < FG_DK ? (vec_len - k0) : FG_DK;
            for (uint32_t j = lane; j < kn; j += 64) {
                const uint32_t i = k0 + j;
                float wv;
                uint32_t si;
                bool zero = false;
                if (part == 0) { wv = window[i]; si = i; }
                else if (i < part) { wv = window[i]; si = sh + i; }
                else if (i < 2 * part) { wv = window[n - 2 * part + i]; si = sh + i; }
                else { wv = 0.0f; si = 0; zero = true; }
                const int32_t L = pl[si], R = (NCH == 2) ? pr[si] : 0;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    int32_t x;
                    if (c == 0) x = cv<0>(L, R);
                    else if (c == 1) x = cv<1>(L, R);
                    else if (c == 2) x = cv<2>(L, R);
                    else x = cv<3>(L, R);
                    const float d = zero ? 0.0f : (float)x * wv;
                    dbuf[c * DSTR + FG_DH + j] = (double)d;
                }
            }
            wave_lds_fence();
            if (on) {
                uint32_t j = 0;
                for (; j + 8 <= kn; j += 8) {
                    double a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { a[u] = cur[j + u]; b[u] = hist[j + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = __builtin_fma(a[u], b[u], acc);
                }
                for (; j < kn; j++) acc = __builtin_fma(cur[j], hist[j], acc);
            }
            wave_lds_fence();
            if (k0 + kn < vec_len) {
                // keep the last FG_DH entries as history (kn == FG_DK here)
                double t[(NC * FG_DH + 63) / 64];
#pragma unroll
                for (int u = 0; u < (NC * FG_DH + 63) / 64; u++) {
                    const uint32_t j = u * 64 + lane;
                    t[u] = (j < NC * FG_DH) ? dbuf[(j / FG_DH) * DSTR + FG_DK + (j % FG_DH)] : 0.0;
                }
                wave_lds_fence();
#pragma unroll
                for (int u = 0; u < (NC * FG_DH + 63) / 64; u++) {
                    const uint32_t j = u * 64 + lane;
                    if (j < NC * FG_DH) dbuf[(j / FG_DH) * DSTR + (j % FG_DH)] = t[u];
                }
                wave_lds_fence();
            }
        }
        if (on) autoc[(cl * pp->nvec + v) * (MAXO + 1) + l] = acc;
        wave_lds_fence();
    }

    __device__ __forceinline__ double ebps(double e, double scale) const
    {
        if (e > 0.0) {
            const double b = 0.5 * log(scale * e) / FG_LN2;
            return b >= 0.0 ? b : 0.0;
        }
        else if (e < 0.0) return 1e32;
        return 0.0;
    }

    // ------------------------------------------------------------------ Levinson-Durbin, order guess, quantiser
    // lane = candidate * nvec + vector.  lres[idx] = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25
    __device__ __forceinline__ void lpc_decide(uint32_t nv, uint32_t mo, uint32_t vmask)
    {
        const uint32_t nidx = (uint32_t)NC * pp->nvec;
        const uint32_t LS = nidx;
        LDS double *lpcw = dbuf;
        LDS float *lpf = (LDS float *)(dbuf + (size_t)mo * LS);
        const uint32_t idx = lane;
        if (idx < nidx) {
            const uint32_t c = idx / pp->nvec, v = idx % pp->nvec;
            const LDS double *A = autoc + (c * pp->nvec + v) * (MAXO + 1);
            bool on = v < nv && ((vmask >> c) & 1);
            if (on && A[0] == 0.0) on = false;
            uint32_t sb = sbp[0];
#pragma unroll
            for (int cc = 1; cc < NC; cc++) if (c == (uint32_t)cc) sb = sbp[cc];
            const double a0 = on ? A[0] : 1.0;
            const uint32_t overhead = sb + pp->qlp_precision;
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
            double err2 = a0;
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
            for (uint32_t jj = 0; jj < ostar; jj++) lpf[jj * LS + idx] = (float)(-lpcw[jj * LS + idx]);
            uint32_t result = 0;
            for (uint32_t j = 0; j < (uint32_t)MAXO; j++) qres[idx * MAXO + j] = 0;
            if (on) {
                bool ok = !(ebps(err2, 0.5 / (double)(n - ostar)) >= (double)sb);
                uint32_t prec = pp->qlp_precision;
                if (sb <= 17) { const uint32_t lim = 32 - sb - ilog2_32(ostar); if (lim < prec) prec = lim; }
                int shift = 0;
                if (ok) {
                    const int p1 = (int)prec - 1;
                    const int32_t qmax = (1 << p1) - 1, qmin = -(1 << p1);
                    double cmax = 0.0;
                    for (uint32_t j = 0; j < ostar; j++) { const double d = fabs((double)lpf[j * LS + idx]); if (d > cmax) cmax = d; }
                    if (cmax <= 0.0) ok = false;
                    else {
                        const int e = (int)((__double_as_longlong(cmax) >> 52) & 0x7FF) - 1022;
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
                            qres[idx * MAXO + j] = qv;
                        }
                        if (neg) shift = 0;
                    }
                }
                result = ostar | (prec << 8) | (((uint32_t)shift & 0xFF) << 16) | ((ok ? 1u : 0u) << 24) | (1u << 25);
            }
            lres[idx] = result;
        }
        wave_lds_fence();
    }

    // ------------------------------------------------------------------ FIR residual of candidate C at sample i
    // xw[k] = candidate value at i-k, k = 0..MAXO.  q zero-padded to MAXO taps (SGPRs).
    __device__ __forceinline__ int32_t fir(const int32_t xw[MAXO + 1], const int32_t q[MAXO], int shift, uint32_t *ovf) const
    {
        if (MAXO == 0) return xw[0];
        if (!ACC64) {
            int32_t s = 0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) s += q[j] * xw[j + 1];
            return xw[0] - (s >> shift);
        }
        i64 s = 0;
#pragma unroll
        for (int j = 0; j < MAXO; j++) s += (i64)q[j] * (i64)xw[j + 1];
        const i64 r = (i64)xw[0] - (s >> shift);
        if (r <= (i64)INT32_MIN || r > (i64)INT32_MAX) *ovf = 1;
        return (int32_t)r;
    }

    // ------------------------------------------------------------------ evaluation pass: residual partition sums of all
    // enabled candidates under predictors (order[c], q[c][], shift[c]); psum[c] returns lane p = partition p.
    __device__ __forceinline__ void eval_pass(const uint32_t order[NC], const int32_t q[NC][MAXO > 0 ? MAXO : 1], const int shift[NC], uint32_t emask,
                              uint32_t pmax, sum_t psum[NC], uint32_t *ovfmask)
    {
        const uint32_t psz = n >> pmax, ipp = psz >> 6, parts = 1u << pmax;
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        uint32_t ovf[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) { psum[c] = 0; ovf[c] = 0; }
        uint32_t t = 0;
        for (uint32_t p = 0; p < parts; p++) {
            sum_t a[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) a[c] = 0;
            for (uint32_t k = 0; k < ipp; k++, t++) {
                const int i = (int)((t << 6) + lane);
                int32_t l[MAXO + 1], r[MAXO + 1];
#pragma unroll
                for (int j = 0; j <= MAXO; j++) { l[j] = pl[i - j]; r[j] = (NCH == 2) ? pr[i - j] : 0; }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (!((emask >> c) & 1)) continue;
                    int32_t xw[MAXO + 1];
#pragma unroll
                    for (int j = 0; j <= MAXO; j++) {
                        if (c == 0) xw[j] = cv<0>(l[j], r[j]);
                        else if (c == 1) xw[j] = cv<1>(l[j], r[j]);
                        else if (c == 2) xw[j] = cv<2>(l[j], r[j]);
                        else xw[j] = cv<3>(l[j], r[j]);
                    }
                    const int32_t res = fir(xw, q[c], shift[c], &ovf[c]);
                    if ((uint32_t)i >= order[c]) a[c] += (uint32_t)abs(res);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (!((emask >> c) & 1)) continue;
                sum_t tot;
                if (ACC64) tot = (sum_t)wave_sum64((u64)a[c]);
                else tot = (sum_t)wave_sum((uint32_t)a[c]);
                if ((uint32_t)lane == p) psum[c] = tot;
            }
        }
        uint32_t om = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) if (ACC64 && __any(ovf[c] != 0)) om |= 1u << c;
        *ovfmask = om;
    }

    // ------------------------------------------------------------------ Rice parameter / partition-order search for one
    // candidate.  In: psum (lane = partition at order pmax).  Out: bits, best order, kbest (lane = partition).
    __device__ __forceinline__ uint32_t rice_search(sum_t psum, uint32_t sb, uint32_t order, uint32_t pmax, uint32_t pmin, uint32_t *best_po,
                                    uint32_t *kbest) const
    {
        const uint32_t limit = pp->rice_limit;
        const bool wrap32 = (sb + 4) < (32 - ilog2_32(n >> pmax));
        u64 s = (u64)psum;
        if (wrap32) s &= 0xFFFFFFFFull;
        uint32_t best_bits = 0, bpo = 0, kb = 0;
        for (int po = (int)pmax; po >= (int)pmin; po--) {
            const uint32_t parts = 1u << po;
            const uint32_t pbase = n >> po;
            uint32_t np = pbase, div = 0x40000u / pbase;
            if (lane == 0) { np -= order; div = 0x40000u / np; }
            uint32_t k = 0;
            if (s >= 2) {
                const u64 qv = ((s - 1) * div) >> 18;
                if (qv != 0) k = ilog2_64(qv) + 1;
            }
            if (k >= limit) k = limit - 1;
            u64 pb = (u64)4 + (u64)(1 + k) * np + (k ? (s >> (k - 1)) : (s << 1)) - (np >> 1);
            if (pb > 0xFFFFFFFFull) pb = 0xFFFFFFFFull;
            if ((uint32_t)lane >= parts) pb = 0;
            const u64 total = wave_sum64(pb) + 6;
            const uint32_t bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
            if (best_bits == 0 || bits < best_bits) { best_bits = bits; bpo = (uint32_t)po; kb = k; }
            if (po > (int)pmin) {
                // merge pairs: lane p <- s[2p] + s[2p+1]
                const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);
                const int src = (lane * 2) & 63;
                const u64 s0 = ((u64)(uint32_t)__shfl((int)hi, src) << 32) | (uint32_t)__shfl((int)lo, src);
                const u64 s1 = ((u64)(uint32_t)__shfl((int)hi, src + 1) << 32) | (uint32_t)__shfl((int)lo, src + 1);
                s = ((uint32_t)lane < (parts >> 1)) ? s0 + s1 : 0;
            }
        }
        *best_po = bpo; *kbest = kb;
        return best_bits;
    }

    // ------------------------------------------------------------------ bit writer (LDS window -> HBM slot)
    __device__ __forceinline__ void bw_init(uint32_t *out, uint32_t words)
    {
        outw = out; slot_words = words; bitpos = 0; wbase = 0;
        for (uint32_t j = lane; j < FG_WINW + 2; j += 64) win[j] = 0;
        wave_lds_fence();
    }
    __device__ __forceinline__ void bw_flush(uint32_t newpos)
    {
        const uint32_t nfull = (newpos >> 5) - wbase;
        if (nfull == 0) return;
        if (wbase + nfull > slot_words) { err |= FG_ERR_SLOT; wbase += nfull; return; }
        if (nfull < 64) {
            const uint32_t v = win[lane];
            if ((uint32_t)lane < nfull) outw[wbase + lane] = __builtin_bswap32(v);
            const uint32_t carry = rl(v, (int)nfull);
            wave_lds_fence();
            if ((uint32_t)lane <= nfull) win[lane] = (lane == 0) ? carry : 0;
        }
        else {
            for (uint32_t j = lane; j < nfull; j += 64) outw[wbase + j] = __builtin_bswap32(win[j]);
            const uint32_t carry = win[nfull];
            wave_lds_fence();
            for (uint32_t j = lane; j < FG_WINW + 2; j += 64) win[j] = 0;
            wave_lds_fence();
            if (lane == 0) win[0] = carry;
        }
        wbase += nfull;
        wave_lds_fence();
    }
    __device__ __forceinline__ void bw_or(uint32_t pos, uint32_t val, uint32_t vbits)
    {
        const uint32_t rel = pos - (wbase << 5);
        const uint32_t word = rel >> 5, sh = rel & 31;
        const u64 x = (u64)val << (64 - sh - vbits);
        __hip_atomic_fetch_or(&win[word], (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t lo = (uint32_t)x;
        if (lo) __hip_atomic_fetch_or(&win[word + 1], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __device__ __forceinline__ void bw_zeros(uint32_t z)
    {
        while (((bitpos + z) >> 5) - wbase >= FG_WINW) {
            const uint32_t np = (wbase + FG_WINW) << 5;
            z -= np - bitpos;
            bitpos = np;
            bw_flush(np);
        }
        bitpos += z;
    }
    __device__ __forceinline__ void bw_round(uint32_t pv, uint32_t pb, uint32_t val, uint32_t vb, uint32_t nb)
    {
        const uint32_t mine = pb + nb;
        const uint32_t incl = wave_scan_add(mine);
        const uint32_t total = rl(incl, 63);
        if (total == 0) return;
        const bool anybig = __any(nb > (1u << 26));
        if (!anybig && (bitpos & 31) + total <= 32u * FG_WINW) {
            const uint32_t off = bitpos + incl - mine;
            if (pb) bw_or(off, pv, pb);
            if (vb) bw_or(off + pb + nb - vb, val, vb);
            wave_lds_fence();
            bitpos += total;
            bw_flush(bitpos);
        }
        else {
            for (int L = 0; L < 64; L++) {
                const uint32_t lpv = rl(pv, L), lpb = rl(pb, L), lval = rl(val, L), lvb = rl(vb, L), lnb = rl(nb, L);
                if (lpb) {
                    if (lane == 0) bw_or(bitpos, lpv, lpb);
                    wave_lds_fence();
                    bitpos += lpb;
                    bw_flush(bitpos);
                }
                if (lnb) {
                    bw_zeros(lnb - lvb);
                    bw_flush(bitpos);
                    if (lvb) {
                        if (lane == 0) bw_or(bitpos, lval, lvb);
                        wave_lds_fence();
                        bitpos += lvb;
                        bw_flush(bitpos);
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void bw_put(uint32_t val, uint32_t bits)
    {
        bw_round(0, 0, lane == 0 ? (bits < 32 ? (val & ((1u << bits) - 1)) : val) : 0, lane == 0 ? bits : 0, lane == 0 ? bits : 0);
    }
    __device__ __forceinline__ void bw_flush_all()
    {
        bw_flush(bitpos);
        if ((bitpos & 31) && lane == 0) {
            if (wbase < slot_words) outw[wbase] = __builtin_bswap32(win[0]);
            else err |= FG_ERR_SLOT;
        }
        err = wave_or32(err);
    }

    // ------------------------------------------------------------------ frame header
    __device__ __forceinline__ void write_header(uint32_t ca, uint32_t frame_number)
    {
        LDS uint8_t *hb = (LDS uint8_t *)misc;
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
            const uint32_t b2 = u << 4;
            const uint32_t sr = pp->sample_rate;
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
            switch (ca) { case 0: u = pp->channels - 1; break; case 1: u = 8; break; case 2: u = 9; break; default: u = 10; break; }
            const uint32_t b3 = u << 4;
            switch (pp->bps) { case 8: u = 1; break; case 12: u = 2; break; case 16: u = 4; break; case 20: u = 5; break;
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
        hl = rfl(hl);
        wave_lds_fence();
        const uint32_t v = (uint32_t)lane < hl ? hb[lane] : 0, b = (uint32_t)lane < hl ? 8 : 0;
        wave_lds_fence();
        bw_round(0, 0, v, b, b);
    }

    // ------------------------------------------------------------------ one subframe
    template <int C>
    __device__ __forceinline__ void write_subframe()
    {
        const uint32_t type = d_type[C], order = d_order[C], w = wst[C], sb = sbp[C];
        const uint32_t mask = sb < 32 ? ((1u << sb) - 1) : 0xFFFFFFFFu;
        const LDS samp_t *pl = sL + FG_PADF, *pr = sR + FG_PADF;
        uint32_t hdr;
        switch (type) {
        case 0: hdr = 0x00; break;
        case 1: hdr = 0x02; break;
        case 2: hdr = 0x10 | (order << 1); break;
        default: hdr = 0x40 | ((order - 1) << 1); break;
        }
        bw_put(hdr | (w ? 1 : 0), 8);
        if (w) bw_round(0, 0, lane == 0 ? 1 : 0, lane == 0 ? 1 : 0, lane == 0 ? w : 0);
        if (type == 0) { bw_put((uint32_t)cv<C>(pl[0], (NCH == 2) ? pr[0] : 0) & mask, sb); return; }
        if (type == 1) {
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                bw_round(0, 0, (uint32_t)cv<C>(pl[i], (NCH == 2) ? pr[i] : 0) & mask, sb, sb);
            }
            return;
        }
        {
            const bool on = (uint32_t)lane < order;
            const uint32_t x = (uint32_t)cv<C>(pl[lane], (NCH == 2) ? pr[lane] : 0) & mask;
            bw_round(0, 0, on ? x : 0, on ? sb : 0, on ? sb : 0);
        }
        int32_t q[MAXO > 0 ? MAXO : 1];
        int shift = 0;
        if (type == 3) {
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rfl((uint32_t)bestq[C * MAXO + j]);
            shift = d_shift[C];
            const uint32_t prec = d_prec[C];
            uint32_t pv = 0, pb = 0, val = 0, vb = 0;
            if (lane == 0) { pv = prec - 1; pb = 4; val = (uint32_t)shift & 31; vb = 5; }
            else if ((uint32_t)lane <= order) { val = (uint32_t)bestq[C * MAXO + lane - 1] & ((1u << prec) - 1); vb = prec; }
            bw_round(pv, pb, val, vb, vb);
        }
        else {
            // fixed predictor of order k == FIR with binomial coefficients, shift 0
            static const int32_t FX[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (j < 4) ? FX[order][j] : 0;
        }
        const uint32_t po = d_porder[C], method = d_method[C];
        bw_put((method << 4) | po, 6);
        const uint32_t plen = method ? 5 : 4;
        const uint32_t psz = n >> po, ipp = psz >> 6;
        uint32_t ovf = 0;
        uint32_t t = 0;
        for (uint32_t p = 0; p < (1u << po); p++) {
            const uint32_t k = rl(d_k[C], (int)p);
            for (uint32_t kk = 0; kk < ipp; kk++, t++) {
                const int i = (int)((t << 6) + lane);
                int32_t xw[MAXO + 1];
#pragma unroll
                for (int j = 0; j <= MAXO; j++) xw[j] = cv<C>(pl[i - j], (NCH == 2) ? pr[i - j] : 0);
                const int32_t r = fir(xw, q, shift, &ovf);
                uint32_t pv = 0, pb = 0, val = 0, vb = 0, nb = 0;
                if ((uint32_t)i >= order) {
                    const uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
                    val = (1u << k) | (u & ((1u << k) - 1));
                    vb = k + 1;
                    nb = (u >> k) + 1 + k;
                    if (kk == 0 && (uint32_t)i == (p == 0 ? order : p * psz)) { pv = k; pb = plen; }
                }
                if (t >= (order >> 6)) bw_round(pv, pb, val, vb, nb);
            }
        }
    }

    __device__ __forceinline__ void write_subframe_c(int c)
    {
        if (c == 0) write_subframe<0>();
        else if (NC > 1 && c == 1) write_subframe<(NC > 1 ? 1 : 0)>();
        else if (NC > 2 && c == 2) write_subframe<(NC > 2 ? 2 : 0)>();
        else if (NC > 3) write_subframe<(NC > 3 ? 3 : 0)>();
    }

    // ------------------------------------------------------------------ pad + CRC-16
    __device__ __forceinline__ uint32_t finish_frame()
    {
        if (bitpos & 7) bitpos += 8 - (bitpos & 7);
        bw_flush(bitpos);
        bw_flush_all();
        __threadfence_block();
        const uint32_t nbytes = bitpos >> 3;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
        uint32_t s = 0;
        const LDS uint16_t *t0 = crct, *thi = crct + 256, *tlo = crct + 512;
        for (uint32_t t = 0; t < T; t++) {
            const int qi = (int)(t * 64 + lane) - (int)pad;
            uint32_t wv = 0;
            if (qi >= 0) wv = __builtin_bswap32(__builtin_nontemporal_load(&outw[qi]));
            s = thi[s >> 8] ^ tlo[s & 0xFF];
            uint32_t cw = 0;
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 24)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 16)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 8)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ wv) & 0xFF];
            s ^= cw;
        }
        s = gf16_mul(s, misc[64 + (63 - lane)]);
        uint32_t crc = wave_xor32(s);
        if (tail) {
            const uint32_t wv = W < slot_words ? __builtin_bswap32(__builtin_nontemporal_load(&outw[W])) : 0;
            for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ (wv >> (24 - 8 * b))) & 0xFF];
        }
        bw_put(crc, 16);
        bw_flush_all();
        return bitpos >> 3;
    }
};

template <bool MS, int NCH, int MAXO, bool ACC64>
__global__ void __launch_bounds__(64)
fg_encode_fast_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, uint8_t *out,
                      FgBlockResult *results, FgDebugRec *dbg, const uint16_t *crctab)
{
    typedef Fast<MS, NCH, MAXO, ACC64> F;
    constexpr int NC = F::NC;
    constexpr int MQ = MAXO > 0 ? MAXO : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FgBlockDesc d = descs[blockIdx.x];
    F e;
    e.pp = &P;
    e.lane = threadIdx.x;
    e.n = d.n;
    e.err = 0;
    e.window = windows + d.win_off;
    size_t off = 0;
    LDS unsigned char *lbase = (LDS unsigned char *)smem;
    auto carve = [&](size_t bytes) __attribute__((always_inline)) { LDS unsigned char *p = lbase + off; off += (bytes + 15) & ~(size_t)15; return p; };
    e.sL = (LDS typename F::samp_t *)carve((size_t)(P.sig_stride + FG_PADF) * sizeof(typename F::samp_t));
    e.sR = (LDS typename F::samp_t *)carve(NCH == 2 ? (size_t)(P.sig_stride + FG_PADF) * sizeof(typename F::samp_t) : 16);
    e.dbuf = (LDS double *)carve(P.lds_dbuf_bytes);
    e.autoc = (LDS double *)carve((size_t)NC * P.nvec * (MAXO + 1) * 8);
    e.qres = (LDS int32_t *)carve((size_t)NC * P.nvec * MQ * 4);
    e.lres = (LDS uint32_t *)carve((size_t)NC * P.nvec * 4);
    e.bestq = (LDS int32_t *)carve((size_t)NC * MQ * 4);
    e.win = (LDS uint32_t *)carve((FG_WINW + 2) * 4);
    e.crct = (LDS uint16_t *)carve(768 * 2);
    e.misc = (LDS uint32_t *)carve(128 * 4);
    for (int j = e.lane; j < 768; j += 64) e.crct[j] = crctab[j];
    e.misc[64 + e.lane] = crctab[768 + e.lane];
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
    auto stamp = [&](int k) __attribute__((always_inline)) { if (mydbg && e.lane == 0) mydbg->t[k] = clock64(); };
#define FG_STOP(k) do { if (P.debug == 100 + (k)) return; } while (0)
    stamp(0);
    e.stage(pcm, d.pcm_off);
    stamp(1);
    FG_STOP(1);

    const uint32_t n = e.n;
    uint32_t pmax0 = 0;
    { uint32_t b = n; while (!(b & 1)) { pmax0++; b >>= 1; } if (pmax0 > 15) pmax0 = 15; }
    if (P.max_po < pmax0) pmax0 = P.max_po;
    const uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;

    // ---- fixed-predictor sums, wasted bits, constant detection
    u64 tot[NC][5];
    e.sums_pass(tot);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (e.wst[c]) {
            if (c == 0) e.template fixed_sums_one<0>(tot[c]);
            else if (c == 1) e.template fixed_sums_one<(NC > 1 ? 1 : 0)>(tot[c]);
            else if (c == 2) e.template fixed_sums_one<(NC > 2 ? 2 : 0)>(tot[c]);
            else e.template fixed_sums_one<(NC > 3 ? 3 : 0)>(tot[c]);
        }
    }
    uint32_t best[NC], guess[NC];
    uint32_t fixed_mask = 0, lpc_mask = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t w = e.wst[c], sb = e.sbp[c];
        const u64 vb = (u64)8 + w + (u64)n * sb;
        best[c] = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
        e.d_type[c] = 1; e.d_order[c] = 0; e.d_prec[c] = 0; e.d_shift[c] = 0; e.d_porder[c] = 0; e.d_method[c] = 0; e.d_k[c] = 0;
        const u64 m34 = tot[c][3] < tot[c][4] ? tot[c][3] : tot[c][4];
        const u64 m234 = tot[c][2] < m34 ? tot[c][2] : m34;
        const u64 m1234 = tot[c][1] < m234 ? tot[c][1] : m234;
        uint32_t g;
        if (tot[c][0] <= m1234) g = 0;
        else if (tot[c][1] <= m234) g = 1;
        else if (tot[c][2] <= m34) g = 2;
        else if (tot[c][3] <= tot[c][4]) g = 3;
        else g = 4;
        guess[c] = g;
        const double len = (double)(n - 4);
        const float rbg = (float)((tot[c][g] > 0) ? log(FG_LN2 * (double)tot[c][g] / len) / FG_LN2 : 0.0);
        bool constant = false;
        if (tot[c][1] == 0) {
            if (c == 0) constant = e.template is_constant<0>();
            else if (c == 1) constant = e.template is_constant<(NC > 1 ? 1 : 0)>();
            else if (c == 2) constant = e.template is_constant<(NC > 2 ? 2 : 0)>();
            else constant = e.template is_constant<(NC > 3 ? 3 : 0)>();
        }
        if (mydbg && e.lane == 0) {
            for (int k = 0; k < 5; k++) mydbg->cand[c].fixed_tot[k] = tot[c][k];
            mydbg->cand[c].fixed_guess = g;
        }
        if (constant) {
            const uint32_t cb = 8 + w + sb;
            if (cb < best[c]) { best[c] = cb; e.d_type[c] = 0; }
        }
        else {
            if (!(rbg >= (float)sb)) fixed_mask |= 1u << c;
            if (P.max_lpc_order > 0) lpc_mask |= 1u << c;
        }
    }
    stamp(2);
    FG_STOP(2);

    // ---- evaluation of one predictor set: eval_pass + rice search + strict-< update of the best
    auto evaluate = [&](const uint32_t order[NC], const int32_t q[NC][MQ], const int shift[NC], const uint32_t prec[NC], uint32_t emask,
                        int kind, uint32_t vecidx) __attribute__((always_inline)) {
        typename F::sum_t psum[NC];
        uint32_t ovfmask = 0;
        e.eval_pass(order, q, shift, emask, pmax0, psum, &ovfmask);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!((emask >> c) & 1)) continue;
            uint32_t est = 0;
            if (!((ovfmask >> c) & 1)) {
                uint32_t bpo, kb;
                const uint32_t rb = e.rice_search(psum[c], e.sbp[c], order[c], pmax0, pmin0, &bpo, &kb);
                est = kind == 0 ? (8 + e.wst[c] + order[c] * e.sbp[c]) : (8 + e.wst[c] + 4 + 5 + order[c] * (prec[c] + e.sbp[c]));
                if (rb < 0xFFFFFFFFu - est) est += rb; else est = 0xFFFFFFFFu;
                if (est > 0 && est < best[c]) {
                    best[c] = est;
                    e.d_type[c] = kind == 0 ? 2 : 3; e.d_order[c] = order[c]; e.d_prec[c] = prec[c]; e.d_shift[c] = shift[c];
                    e.d_porder[c] = bpo; e.d_k[c] = kb;
                    e.d_method[c] = __any(((uint32_t)e.lane < (1u << bpo)) && kb >= 15) ? 1 : 0;
                    if (kind == 1 && e.lane < MQ) e.bestq[c * MQ + e.lane] = e.qres[((uint32_t)c * P.nvec + vecidx) * MQ + e.lane];
                }
            }
            if (mydbg && e.lane == 0) {
                if (kind == 0) mydbg->cand[c].fixed_bits = est;
                else mydbg->cand[c].lpc_bits[vecidx] = est;
            }
        }
        wave_lds_fence();
    };

    // ---- fixed predictors (libFLAC evaluates them before LPC)
    if (fixed_mask) {
        uint32_t order[NC], prec[NC];
        int32_t q[NC][MQ];
        int shift[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            static const int32_t FX[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
            order[c] = guess[c]; shift[c] = 0; prec[c] = 0;
#pragma unroll
            for (int j = 0; j < MQ; j++) q[c][j] = (j < 4) ? FX[guess[c]][j] : 0;
        }
        evaluate(order, q, shift, prec, fixed_mask, 0, 0);
    }
    stamp(3);
    FG_STOP(3);

    // ---- LPC
    if (lpc_mask && MAXO > 0) {
        const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
        uint32_t nv = 0;
        e.autocorr_vector(0, n, 0, 0, mo);
        nv = 1;
        if (P.apod_parts >= 2) {
            for (uint32_t b = 2; b <= P.apod_parts; b++) {
                const uint32_t cmax = (b == 2) ? 2 : 2 * b - 1;
                for (uint32_t cc = 0; cc <= cmax; cc += (b == 2 ? 2 : 1)) {
                    if (n / b <= 32) continue;
                    if (!(cc & 1)) e.autocorr_vector(nv, n / b, n / b / 2, (cc / 2 * n) / b, mo);
                    else {
                        const uint32_t total = (uint32_t)NC * (mo + 1);
                        for (uint32_t j = e.lane; j < total; j += 64) {
                            const uint32_t c = j / (mo + 1), l = j % (mo + 1);
                            LDS double *base = e.autoc + c * P.nvec * (MAXO + 1);
                            const double prev = base[(nv - 1) * (MAXO + 1) + l];
                            base[nv * (MAXO + 1) + l] = (l < mo) ? base[l] - prev : prev;
                        }
                        wave_lds_fence();
                    }
                    nv++;
                }
            }
        }
        if (mydbg) {
            for (uint32_t j = e.lane; j < (uint32_t)NC * nv * (mo + 1); j += 64) {
                const uint32_t c = j / (nv * (mo + 1)), r = j % (nv * (mo + 1)), v = r / (mo + 1), l = r % (mo + 1);
                mydbg->cand[c].autoc[v][l] = e.autoc[(c * P.nvec + v) * (MAXO + 1) + l];
            }
            if (e.lane < NC) mydbg->cand[e.lane].nvec = nv;
        }
        stamp(4);
    FG_STOP(4);
        e.lpc_decide(nv, mo, lpc_mask);
        stamp(5);
    FG_STOP(5);
        for (uint32_t v = 0; v < nv; v++) {
            uint32_t order[NC], prec[NC], emask = 0;
            int32_t q[NC][MQ];
            int shift[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t idx = (uint32_t)c * P.nvec + v;
                const uint32_t r = rfl(e.lres[idx]);
                order[c] = r & 0xFF; prec[c] = (r >> 8) & 0xFF; shift[c] = (int)(int8_t)((r >> 16) & 0xFF);
                if (((lpc_mask >> c) & 1) && ((r >> 24) & 1)) emask |= 1u << c;
                if (order[c] == 0) order[c] = 1;
#pragma unroll
                for (int j = 0; j < MQ; j++) q[c][j] = (int32_t)rfl((uint32_t)e.qres[idx * MQ + j]);
                if (mydbg && e.lane == 0) mydbg->cand[c].lpc_guess[v] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            }
            if (emask) evaluate(order, q, shift, prec, emask, 1, v);
        }
    }
    stamp(6);
    FG_STOP(6);

    // ---- channel assignment
    uint32_t ca = 0, sub[2] = {0, 1};
    if (MS) {
        if (d.forced_ca != 0xFF) ca = d.forced_ca;
        else {
            const uint32_t bits[4] = {best[0] + best[NC > 1 ? 1 : 0], best[0] + best[NC > 3 ? 3 : 0], best[NC > 1 ? 1 : 0] + best[NC > 3 ? 3 : 0],
                                      best[NC > 2 ? 2 : 0] + best[NC > 3 ? 3 : 0]};
            uint32_t mn = bits[0];
#pragma unroll
            for (uint32_t k = 1; k <= 3; k++) if (bits[k] < mn) { mn = bits[k]; ca = k; }
        }
        switch (ca) { case 0: sub[0] = 0; sub[1] = 1; break; case 1: sub[0] = 0; sub[1] = 3; break;
                      case 2: sub[0] = 3; sub[1] = 1; break; default: sub[0] = 2; sub[1] = 3; break; }
    }
    if (mydbg && e.lane < NC) {
        FgDebugCand *dc = &mydbg->cand[e.lane];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (e.lane == c) {
                dc->wasted = e.wst[c]; dc->sbps = e.sbp[c]; dc->type = e.d_type[c]; dc->order = e.d_type[c] >= 2 ? e.d_order[c] : 0;
                dc->precision = e.d_type[c] == 3 ? e.d_prec[c] : 0; dc->shift = e.d_type[c] == 3 ? e.d_shift[c] : 0;
                dc->bits = best[c]; dc->porder = e.d_type[c] >= 2 ? e.d_porder[c] : 0; dc->rice_method = e.d_type[c] >= 2 ? e.d_method[c] : 0;
                for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (e.d_type[c] == 3 && j < e.d_order[c] && j < (uint32_t)MQ) ? e.bestq[c * MQ + j] : 0;
            }
        }
    }
    if (mydbg) {
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (e.d_type[c] >= 2 && (uint32_t)e.lane < (1u << e.d_porder[c])) mydbg->cand[c].rice_params[e.lane] = e.d_k[c];
    }
    stamp(7);
    FG_STOP(7);

    // ---- pack
    e.bw_init((uint32_t *)(out + (size_t)d.out_slot * P.slot_bytes), P.slot_bytes / 4);
    e.write_header(ca, d.frame_number);
    e.write_subframe_c((int)sub[0]);
    if (NCH == 2) e.write_subframe_c((int)sub[1]);
    stamp(8);
    FG_STOP(8);
    const uint32_t bytes = e.finish_frame();
    stamp(9);
    if (e.lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->bytes = bytes; r->ca = ca; r->err = e.err; r->reserved = 1;
#pragma unroll
        for (int c = 0; c < 4; c++) r->best_bits[c] = c < NC ? best[c < NC ? c : 0] : 0;
    }
}

}  // namespace
